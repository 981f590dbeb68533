// gsr_device.h -- device-side helpers shared by the rasterizer kernels (gfx950 only).
//
// Numerics contract of the FORWARD path (what makes it bit-reproducible against a CPU
// restatement): fp32, translation units built with -ffp-contract=off, expressions evaluated in
// the order written, IEEE division / sqrtf (hipcc's default correctly-rounded forms), and
// gsr_expf() below instead of a library exp.  The one fp64 spot is ndc2pix, which the upstream
// rasterizer writes with double literals.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gsr.h"

#define GSR_WAVE 64
#define GSR_TILE_PIX (GSR_BLOCK_X * GSR_BLOCK_Y)
#define GSR_SORT_LDS_KEYS 8192   // 64 KiB of uint64 keys per workgroup in the tile sort
#define GSR_SORT_XL_KEYS 16384  // 128 KiB of LDS, 16 keys per thread; beyond this the sort runs in global memory
#define GSR_SORT_SMALL_KEYS 2048 // tiles up to this many entries take the 256-thread / 16 KiB class
#define GSR_ACC_STRIDE 16        // floats per splat in the backward accumulator: nine sums in a 64-byte line of their own (an atomic burst never straddles two lines)
#define GSR_ACC64_STRIDE 10      // int64 per splat in the deterministic mode's fixed-point accumulator (9 sums + pad, 80 B)
#define GSR_FIXED_BITS 58        // fixed point: a partial p of splat s is added as llrint(p * 2^(GSR_FIXED_BITS - e_s - e_g)) with
                                 // 2^e_g > max |dL/dpixel| and 2^e_s >= the splat's own bound on the sum (two classes: conic sums, the rest) in units of
                                 // max |dL/dpixel| (k_preprocess: pixels it can touch x 8 x max(1, largest 2-D variance)).  The bound is on the
                                 // whole sum, so the only headroom needed is for what it leaves out (colours above 8/3): 5 bits below 2^63.
                                 // Resolution 2^-58 of the bound: a screen-filling splat's 10^5 partials, whose sum is ~10^-6 of its bound,
                                 // still come out at 10^-8 relative (at 46 bits the 3300 x 3300 test scene was off by 3e-3).
#define GSR_LDS_HIST_TILES 40960 // 160 KB of LDS / 4 B: the largest tile grid k_count / k_scatter privatise
#define GSR_RANK_MAX_BUCKETS 4096 // depth buckets of the rank path (16 KB of LDS beside the tile histogram)
#define GSR_RANK_HIST_TILES (GSR_LDS_HIST_TILES - GSR_RANK_MAX_BUCKETS)   // the largest grid of tile CORNERS, (gx+1)(gy+1), k_rcount / k_rscatter privatise
#define GSR_RANK_BANDS 24          // frames beyond GSR_RANK_MAX_SPLATS: horizontal bands of tile rows with a depth rank of their own (<= this many)
#define GSR_RANK_MAX_BANDS 32      // slots of BinHeader::band_total
#define GSR_RANK_BAND_CHUNK 256    // consecutive depth ranks one wave of k_band_count / k_band_rank walks
#define GSR_RANK_IDX_BITS 28      // a tile-list entry of the rank path is (rank, splat | quadrant mask << 28)
#define GSR_RANK_MAX_SPLATS 262144 // ranks one tile bitmap holds (8192 words of LDS: k_tile_rank); frames with more splats rank them per band of tile rows
#define GSR_RANK_TILE_THREADS 512 // threads of a k_tile_rank workgroup (a tile); swept 256 / 512 / 1024: 29 / 21 / 31 us at cfg3
#define GSR_RANK_WINDOW 4096     // k_tile_rank: entries of the sorted list per epilogue round (16 per thread)
#ifndef GSR_RANK_GROUP
#define GSR_RANK_GROUP 16         // lanes that expand one splat's tile rect together in k_rcount / k_rscatter (4 splats per wave at a time)
#endif
#define GSR_RANK_BIN_THREADS 1024 // threads per workgroup of k_rcount / k_rdscatter / k_rscatter (<= GSR_BIN_BLOCKS workgroups: 16 waves each keep the SIMDs busy)

#include "bind_math.h"

namespace gsr {

typedef float f32x8 __attribute__((ext_vector_type(8)));   // eight consecutive scalar registers (one s_load_dwordx8)

struct Settings {  // by-value kernel argument: scalars + the four device pointers of GsrSettings
    int H, W;
    float tanfovx, tanfovy;
    float scale_modifier;
    int sh_degree;
    int exact_scale_grad;
    int forward_only;
    int deterministic;
    int fast_blend;      // EFFECTIVE fast mode of this frame (gsr_api.hip: off with `deterministic` and on the per-tile sort path)
    int cont_chunks;     // fast blend: a quadrant's lone walk hands over to the continuation workgroups at entry cont_chunks * GSR_BWD_SEGMENT (0: never)
    int cont_mode;       // 1: they sit at the end of k_render's own grid and wait; 2: a kernel of their own behind it
    const float* __restrict__ bg;
    const float* __restrict__ viewmatrix;
    const float* __restrict__ projmatrix;
    const float* __restrict__ campos;
};

// SH basis constants (values of the reference's utils/sh_utils.py:26-43)
__device__ constexpr float kC0 = 0.28209479177387814f;
__device__ constexpr float kC1 = 0.4886025119029199f;
__device__ constexpr float kC2_0 = 1.0925484305920792f;
__device__ constexpr float kC2_1 = -1.0925484305920792f;
__device__ constexpr float kC2_2 = 0.31539156525252005f;
__device__ constexpr float kC2_3 = -1.0925484305920792f;
__device__ constexpr float kC2_4 = 0.5462742152960396f;
__device__ constexpr float kC3_0 = -0.5900435899266435f;
__device__ constexpr float kC3_1 = 2.890611442640554f;
__device__ constexpr float kC3_2 = -0.4570457994644658f;
__device__ constexpr float kC3_3 = 0.3731763325901154f;
__device__ constexpr float kC3_4 = -0.4570457994644658f;
__device__ constexpr float kC3_5 = 1.445305721320277f;
__device__ constexpr float kC3_6 = -0.5900435899266435f;

__device__ __forceinline__ float sel_min(float a, float b) { return a < b ? a : b; }
__device__ __forceinline__ float sel_max(float a, float b) { return a > b ? a : b; }

// exp(x) for the compositing weights: 2^(x*log2e) with n = rint(x*log2e), the fraction rebuilt by
// two fmaf against a hi/lo split of log2e, a degree-7 polynomial for 2^f and an exact scale by 2^n.
// Only fmaf / rint / ldexp, so a CPU evaluates the same bits.
__device__ __forceinline__ float gsr_expf(float x)
{
    if (!(x > -87.0f)) return 0.0f;
    if (x > 88.0f) x = 88.0f;
    const float l2e_hi = 1.44269502162933349609375f;
    const float l2e_lo = 1.92596299112661746e-8f;
    const float n = __builtin_rintf(x * l2e_hi);
    float f = __builtin_fmaf(x, l2e_hi, -n);
    f = __builtin_fmaf(x, l2e_lo, f);
    float p = 1.52527338040598402800e-5f;
    p = __builtin_fmaf(p, f, 1.54035303933816099544e-4f);
    p = __builtin_fmaf(p, f, 1.33335581464284434234e-3f);
    p = __builtin_fmaf(p, f, 9.61812910762847716197e-3f);
    p = __builtin_fmaf(p, f, 5.55041086648215799532e-2f);
    p = __builtin_fmaf(p, f, 2.40226506959100712334e-1f);
    p = __builtin_fmaf(p, f, 6.93147180559945309417e-1f);
    p = __builtin_fmaf(p, f, 1.0f);
    return __builtin_ldexpf(p, (int)n);
}

// Same polynomial without the range guards, for the blend loops: there the result is only USED when
// -87 < x <= 0 (x > 0 is skipped, and below -87 the un-guarded value is < 2^-125, which fails the
// alpha >= 1/255 test exactly like the guarded 0 does), so every used value is bit-identical to gsr_expf.
__device__ __forceinline__ float gsr_expf_blend(float x)
{
    const float l2e_hi = 1.44269502162933349609375f;
    const float l2e_lo = 1.92596299112661746e-8f;
    const float n = __builtin_rintf(x * l2e_hi);
    float f = __builtin_fmaf(x, l2e_hi, -n);
    f = __builtin_fmaf(x, l2e_lo, f);
    float p = 1.52527338040598402800e-5f;
    p = __builtin_fmaf(p, f, 1.54035303933816099544e-4f);
    p = __builtin_fmaf(p, f, 1.33335581464284434234e-3f);
    p = __builtin_fmaf(p, f, 9.61812910762847716197e-3f);
    p = __builtin_fmaf(p, f, 5.55041086648215799532e-2f);
    p = __builtin_fmaf(p, f, 2.40226506959100712334e-1f);
    p = __builtin_fmaf(p, f, 6.93147180559945309417e-1f);
    p = __builtin_fmaf(p, f, 1.0f);
    return __builtin_ldexpf(p, (int)n);
}

// float -> int32 with saturation and NaN -> 0 (what v_cvt_i32_f32 does, spelled out so that the
// C++ out-of-range UB never enters)
__device__ __forceinline__ int f2i_sat(float v)
{
    if (v != v) return 0;
    if (v >= 2147483648.0f) return 2147483647;
    if (v <= -2147483648.0f) return (-2147483647 - 1);
    return (int)v;
}

__device__ __forceinline__ float ndc2pix(float v, int S)
{
    return (float)((((double)v + 1.0) * (double)S - 1.0) * 0.5);
}

__device__ __forceinline__ int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// ---- wave64 reductions on the DPP network -------------------------------------------------
// xor-butterfly inside each 16-lane row (quad_perm, row_half_mirror, row_mirror), then
// row_bcast:15 into rows 1,3 and row_bcast:31 into rows 2,3: lanes 48..63 end up holding the
// full 64-lane sum.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum_hi(float v)
{
    v += dpp_f<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
    v += dpp_f<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
    v += dpp_f<0x141, 0xf>(v);  // row_half_mirror
    v += dpp_f<0x140, 0xf>(v);  // row_mirror
    v += dpp_f<0x142, 0xa>(v);  // row_bcast:15 -> rows 1,3
    v += dpp_f<0x143, 0xc>(v);  // row_bcast:31 -> rows 2,3
    return v;                   // valid in lanes 48..63
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v)
{
    // ds_swizzle-free max: butterfly through readlane-able DPP moves
    auto mx = [](uint32_t a, uint32_t b) { return a > b ? a : b; };
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false));
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, false));
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, false));
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xf, 0xf, false));
    uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)v, 0);
    uint32_t b = (uint32_t)__builtin_amdgcn_readlane((int)v, 16);
    uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)v, 32);
    uint32_t d = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
    return mx(mx(a, b), mx(c, d));
}

// inclusive prefix sum across the 64 lanes in six DPP adds (row_shr inside the rows of 16, then the row totals broadcast to the
// rows above): no LDS crossbar round trips (ds_bpermute), which is what __shfl_up costs per step
__device__ __forceinline__ uint32_t wave_scan_incl_u32(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);   // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);   // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);   // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);   // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2, 3
    return v;
}

// ---- tile-rect expansion shared by the counting and the scatter pass ------------------------
// Calls f(tile_id, k) for every tile of the rect.  Splats touching <= SMALL tiles are expanded by
// their own lane; larger ones are expanded cooperatively by the whole wave (one splat at a time,
// lanes striding the rect) so a screen-filling splat does not serialise one lane for thousands of
// iterations.  Must be called by all 64 lanes (n == 0 for idle ones).
// ---- exact reach test ------------------------------------------------------------------------------------
// Can the splat's {alpha >= 1/255} ellipse reach a rectangle of pixel centres?
// alpha >= 1/255  <=>  Q(d) = 1/2 (A dx^2 + 2 B dx dy + C dy^2) <= tau = ln(255 o).  A rectangle is reachable iff the
// minimum of the convex Q over it can be <= tau (0 if the centre is inside, else attained on one of the four edges);
// padded so fp32 rounding in the blend can never turn a dropped pair into a contributor.
struct Reach {
    float px, py, A, B, C, tau, nBiC, nBiA;
    int mode;   // 0: test, 1: keep everywhere (conic not positive definite), 2: never (opacity < 1/255)
};
__device__ __forceinline__ Reach reach_of(float px, float py, float A, float B, float C, float opacity)
{
    Reach r;
    r.px = px; r.py = py; r.A = A; r.B = B; r.C = C;
    r.tau = 0.f; r.nBiC = 0.f; r.nBiA = 0.f;
    if (!(opacity * 255.0f >= 1.0f)) { r.mode = 2; return r; }   // alpha = min(0.99, o * G) can reach 1/255 only if o >= 1/255
    r.tau = logf(255.0f * opacity) * 1.0001f + 1e-3f;
    if (!(A > 0.f && C > 0.f && (A * C - B * B) > 0.f)) { r.mode = 1; return r; }
    r.nBiC = -B / C;
    r.nBiA = -B / A;
    r.mode = 0;
    return r;
}
// rectangle of pixel centres [x0, x0 + w] x [y0, y0 + h]
// Q is convex with its minimum (0) at the splat centre, so over a rectangle that does not hold the centre the minimum sits on
// an edge that FACES the centre: at most one vertical and one horizontal edge need the 1-D minimisation (the two others can
// only give larger values).
__device__ __forceinline__ bool rect_reach(const Reach& r, float x0, float y0, float w, float h)
{
    if (r.mode) return r.mode == 1;
    const float a0 = r.px - (x0 + w), a1 = r.px - x0;   // dx range over the rectangle
    const float b0 = r.py - (y0 + h), b1 = r.py - y0;
    const bool in_x = a0 <= 0.f && a1 >= 0.f, in_y = b0 <= 0.f && b1 >= 0.f;
    if (in_x && in_y) return true;
    const float a = a0 > 0.f ? a0 : a1;                  // the facing vertical edge (dx = a) when the centre is left / right of the rect
    const float b = b0 > 0.f ? b0 : b1;                  // the facing horizontal edge (dy = b) when it is above / below
    const float dyv = fminf(fmaxf(r.nBiC * a, b0), b1);  // dx = a fixed, dy in [b0, b1]
    const float qv = 0.5f * (r.A * a * a + 2.f * r.B * a * dyv + r.C * dyv * dyv);
    const float dxh = fminf(fmaxf(r.nBiA * b, a0), a1);  // dy = b fixed, dx in [a0, a1]
    const float qh = 0.5f * (r.A * dxh * dxh + 2.f * r.B * dxh * b + r.C * b * b);
    const float big = 3.0e38f;
    const float mn = fminf(in_x ? big : qv, in_y ? big : qh);
    return mn * 0.999f - 1e-3f <= r.tau;
}

// The same question per 8-pixel BAND, which is what the quadrant streams ask (one answer per band serves every quadrant of it):
// the x-extent [xl, xr] of {Q <= T} over the rows of pixel centres y0 .. y0 + 7.  The extreme of dx over the band sits where the
// ellipse's own extreme point (dx_m, dyr), dyr = -(B/C) dx_m, is clamped into the band: on the line dy = b,
//     A dx^2 + 2 B b dx + C b^2 - 2 T = 0   =>   dx = (-B b +- sqrt(D)) / A,   D = 2 T A - det b^2,
// real iff the line meets the ellipse.  A quadrant of pixel centres x0 .. x0 + 7 is reached iff xl <= x0 + 7 and xr >= x0, i.e.
//     sqrt(Dl) >= (px - eps - x0 - 7) A - B bl     and     sqrt(Dr) >= (x0 - px - eps) A + B br
// -- compared through the squares, so the test costs no square root.  T = (tau + 1e-3) / 0.999 carries rect_reach's padding,
// eps = 1e-2 px another margin on the interval.
struct Span {
    float px, py, B, det, twoTA, A, dyr;
    int mode;   // as Reach: 0 test, 1 keep everywhere, 2 never
};
__device__ __forceinline__ Span span_of(const Reach& r)
{
    Span s;
    s.px = r.px; s.py = r.py; s.B = r.B; s.mode = r.mode;
    s.det = 0.f; s.twoTA = 0.f; s.A = 0.f; s.dyr = 0.f;
    if (r.mode) return s;
    const float T = (r.tau + 1e-3f) / 0.999f;
    s.det = r.A * r.C - r.B * r.B;
    s.twoTA = 2.f * T * r.A;
    s.A = r.A;
    s.dyr = r.nBiC * sqrtf(2.f * T * r.C / s.det);
    if (!(fabsf(s.dyr) < 1e15f && s.twoTA < 1e30f && s.A > 1e-30f && s.A < 1e30f && s.det > 0.f)) s.mode = 1;   // degenerate: keep everywhere
    return s;
}
struct Band {
    float dl, dr, Bbl, Bbr;   // discriminants and B b on the two lines that carry the band's left / right extreme
    int any;                  // 0: the ellipse does not reach the band, 1: test, 2: reaches everything (degenerate conic)
};
__device__ __forceinline__ Band band_of(const Span& s, float y0)
{
    Band e;
    e.dl = e.dr = e.Bbl = e.Bbr = 0.f;
    if (s.mode) { e.any = s.mode == 1 ? 2 : 0; return e; }
    const float b0 = y0 - s.py, b1 = b0 + 7.f;
    const float br = fminf(fmaxf(s.dyr, b0), b1), bl = fminf(fmaxf(-s.dyr, b0), b1);
    e.dr = s.twoTA - s.det * br * br;
    e.dl = s.twoTA - s.det * bl * bl;
    e.Bbr = s.B * br;
    e.Bbl = s.B * bl;
    e.any = (e.dr >= 0.f && e.dl >= 0.f) ? 1 : 0;
    return e;
}
// quadrant of pixel centres x0 .. x0 + 7 of the band
__device__ __forceinline__ bool band_hit(const Band& e, const Span& s, float x0)
{
    const float tl = (s.px - 1e-2f - (x0 + 7.f)) * s.A - e.Bbl;
    const float tr = (x0 - s.px - 1e-2f) * s.A + e.Bbr;
    const bool hit = (tl <= 0.f || e.dl >= tl * tl) && (tr <= 0.f || e.dr >= tr * tr);
    return e.any == 2 || (e.any == 1 && hit);
}

// Snug tile rect (GsrSettings.tile_culling): the axis-aligned bounding box of the {alpha >= 1/255} ellipse
//   |dx| <= sqrt(2 tau C / det),  |dy| <= sqrt(2 tau A / det),  det = A C - B^2
// intersected with the reference's 3-sigma rect.  A tile column tx holds pixel centres 16 tx .. 16 tx + 15, so it is
// reachable iff 16 tx <= px + xr and 16 tx + 15 >= px - xr.  Costs a few instructions per splat and nothing per tile.
__device__ __forceinline__ void snug_rect(const Reach& r, int& minx, int& miny, int& maxx, int& maxy, uint32_t& n)
{
    if (r.mode == 1 || n == 0) return;
    if (r.mode == 2) { n = 0; return; }
    const float det = r.A * r.C - r.B * r.B;
    const float k = 2.f * r.tau / det;
    const float xr = sqrtf(k * r.C) * 1.001f + 1e-2f, yr = sqrtf(k * r.A) * 1.001f + 1e-2f;
    if (!(xr < 1e9f && yr < 1e9f)) return;   // degenerate: keep the reference's rect
    const float fx = (float)GSR_BLOCK_X, fy = (float)GSR_BLOCK_Y;
    const int lox = (int)ceilf((r.px - xr - (fx - 1.f)) / fx), hix = (int)floorf((r.px + xr) / fx) + 1;
    const int loy = (int)ceilf((r.py - yr - (fy - 1.f)) / fy), hiy = (int)floorf((r.py + yr) / fy) + 1;
    minx = max(minx, lox); maxx = min(maxx, hix);
    miny = max(miny, loy); maxy = min(maxy, hiy);
    n = (maxx > minx && maxy > miny) ? (uint32_t)((maxx - minx) * (maxy - miny)) : 0u;
}

// Which of the tile's four 8x8 quadrants can the splat's {alpha >= 1/255} ellipse reach?  (bit q set = keep)
// Same test as the tile-level one (rect_reach, gsr_device.h), on the quadrant's rectangle of pixel centres.
__device__ __forceinline__ uint32_t quadrant_mask_of(const Span& s, float ox, float oy)
{
    // band_of / band_hit for the tile's two bands and two columns, written WITHOUT control flow: the same expressions in the same order
    // (the masks are the same bits), every comparison evaluated for every lane and combined with bit operations.  As short-circuit code
    // the compiler turned this into eight exec-mask branches per tile and ~200 instructions; the binning pass that calls it once per
    // (splat, tile) instance is bound by exactly that instruction stream (k_rscatter: 23 ps per instance at 2 M splats).
    uint32_t m = 0u;
#pragma unroll
    for (int by = 0; by < 2; ++by) {
        const float y0 = by ? oy + 8.f : oy;
        const float b0 = y0 - s.py, b1 = b0 + 7.f;
        const float br = fminf(fmaxf(s.dyr, b0), b1), bl = fminf(fmaxf(-s.dyr, b0), b1);
        const float dr = s.twoTA - s.det * br * br;
        const float dl = s.twoTA - s.det * bl * bl;
        const float Bbr = s.B * br;
        const float Bbl = s.B * bl;
        const uint32_t any = (uint32_t)(dr >= 0.f) & (uint32_t)(dl >= 0.f);
#pragma unroll
        for (int bx = 0; bx < 2; ++bx) {
            const float x0 = bx ? ox + 8.f : ox;
            const float tl = (s.px - 1e-2f - (x0 + 7.f)) * s.A - Bbl;
            const float tr = (x0 - s.px - 1e-2f) * s.A + Bbr;
            const uint32_t hit = ((uint32_t)(tl <= 0.f) | (uint32_t)(dl >= tl * tl)) & ((uint32_t)(tr <= 0.f) | (uint32_t)(dr >= tr * tr)) & any;
            m |= hit << (2 * by + bx);
        }
    }
    return s.mode == 1 ? 15u : (s.mode == 2 ? 0u : m);
}
__device__ __forceinline__ uint32_t quadrant_mask(float2 p, float4 co, float ox, float oy)
{
    return quadrant_mask_of(span_of(reach_of(p.x, p.y, co.x, co.y, co.z, co.w)), ox, oy);
}

template <typename F>
__device__ __forceinline__ void for_each_tile(int minx, int miny, int maxx, int maxy, uint32_t n, int gx, F f,
                                              uint32_t payload0, uint32_t payload1)
{
    constexpr uint32_t SMALL = 16;
    if (n > 0 && n <= SMALL) {
        for (int y = miny; y < maxy; ++y)
            for (int x = minx; x < maxx; ++x) f((uint32_t)(y * gx + x), payload0, payload1);
    }
    uint64_t big = __ballot(n > SMALL);
    const int lane = lane_id();
    while (big) {
        const int src = __builtin_ctzll(big);
        big &= big - 1;
        const int bminx = __builtin_amdgcn_readlane(minx, src);
        const int bminy = __builtin_amdgcn_readlane(miny, src);
        const int bmaxx = __builtin_amdgcn_readlane(maxx, src);
        const uint32_t bn = (uint32_t)__builtin_amdgcn_readlane((int)n, src);
        const uint32_t p0 = (uint32_t)__builtin_amdgcn_readlane((int)payload0, src);
        const uint32_t p1 = (uint32_t)__builtin_amdgcn_readlane((int)payload1, src);
        const uint32_t w = (uint32_t)(bmaxx - bminx);
        for (uint32_t k = (uint32_t)lane; k < bn; k += GSR_WAVE) {
            const uint32_t y = (uint32_t)bminy + k / w;
            const uint32_t x = (uint32_t)bminx + k % w;
            f(y * (uint32_t)gx + x, p0, p1);
        }
    }
}

// ---- SH rows staged through LDS -------------------------------------------------------------------
// A splat's SH block is 3*M contiguous floats (192 B at degree 3).  Read or written per thread that is a
// 192-byte stride between lanes; instead the workgroup moves its 256 rows as one contiguous, fully
// coalesced float4 stream to/from LDS and every thread works on its own LDS row.  Row stride (3M)|1 is odd,
// so the per-thread dword accesses are bank-conflict free.
#define GSR_SH_ROWS 256
// k_preprocess_bwd: 208 splats per workgroup -- four workgroups' staged SH rows (208 x 49 floats each) fill the CU's 160 KB of LDS exactly: 16 waves
// per CU instead of 12 (the kernel needs 90 VGPRs) and 212 992 splats resident at once instead of 196 608 (BASELINE configs[3], 200 k splats, was
// 782 workgroups for 768 places).  Measured on one box, 256 -> 208: 29.5 -> 28.8 us at 100 k, 59.1 -> 48.6 us at 200 k; 192 and 128 are slower.
#ifndef GSR_PREBWD_ROWS
#define GSR_PREBWD_ROWS 208
#endif
#define GSR_SH_MAX_STRIDE 49
__device__ __forceinline__ int sh_row_stride(int M) { return (3 * M) | 1; }

// copies `rows` source rows of width w into LDS columns [col0, col0+w) of rows laid out with `stride`
template <int NT = GSR_SH_ROWS>
__device__ __forceinline__ void sh_rows_load(float* __restrict__ lds, const float* __restrict__ src, int first, int rows, int w, int stride,
                                             int col0, int tid)
{
    const int total = rows * w;
    const float* base = src + (size_t)first * w;
    if ((total & 3) == 0 && ((((size_t)first * w) & 3) == 0)) {
        const float4* b4 = reinterpret_cast<const float4*>(base);
        for (int f = tid * 4; f < total; f += NT * 4) {
            const float4 v = b4[f >> 2];
            int r = f / w, c = f - r * w;
            const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                lds[r * stride + col0 + c] = e[k];
                if (++c == w) { c = 0; ++r; }
            }
        }
    } else {
        for (int f = tid; f < total; f += NT) {
            const int r = f / w, c = f - r * w;
            lds[r * stride + col0 + c] = base[f];
        }
    }
}

template <int NT = GSR_SH_ROWS>
__device__ __forceinline__ void sh_rows_store(const float* __restrict__ lds, float* __restrict__ dst, int first, int rows, int w, int stride,
                                              int col0, int tid)
{
    const int total = rows * w;
    float* base = dst + (size_t)first * w;
    if ((total & 3) == 0 && ((((size_t)first * w) & 3) == 0)) {
        float4* b4 = reinterpret_cast<float4*>(base);
        for (int f = tid * 4; f < total; f += NT * 4) {
            int r = f / w, c = f - r * w;
            float e[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                e[k] = lds[r * stride + col0 + c];
                if (++c == w) { c = 0; ++r; }
            }
            b4[f >> 2] = make_float4(e[0], e[1], e[2], e[3]);
        }
    } else {
        for (int f = tid; f < total; f += NT) {
            const int r = f / w, c = f - r * w;
            base[f] = lds[r * stride + col0 + c];
        }
    }
}

// ---- kernel argument blocks and entry points (gsr_forward.hip / gsr_backward.hip / gsr_binning.hip) ----------------
#define GSR_STAT_SLOTS 32         // k_preprocess spreads its per-block statistics over this many 64-byte slots (one hot line would
                                  // serialise ~1600 L2 atomics per frame); the first binning kernel folds them into the header
struct BinStatSlot {              // 64 bytes
    uint32_t nvis, dmin_inv, dmax_bits, pad0;
    unsigned long long binned_tiles, rect_total;
    uint32_t pad1[8];
};
struct BinHeader {              // first bytes of the binning buffer (include/gsr.h: GsrBinningLayout); 256 bytes + the slots
    unsigned long long total;       // instances of this frame (tile instances / quadrant-stream entries)
    unsigned long long rect_total;  // sum of tiles_touched: the reference's num_rendered
    uint32_t nvis;                  // production: splats that are binned (non-empty snug rect)
    uint32_t dmin_inv;              // production: ~(bits of the smallest binned depth)  (so that a zeroed slot is +inf)
    uint32_t dmax_bits;             // production: bits of the largest binned depth
    uint32_t dmin_bits;             // = ~dmin_inv
    unsigned long long binned_tiles;// production: tile instances after snug-rect culling (what the per-tile sort path would bin)
    uint32_t band_total[GSR_RANK_MAX_BANDS];   // rank path with bands: binned splats whose rect touches band b (= the band's rank space)
    uint32_t chunk_imbalance;       // rank path: 1 when the busiest 256 consecutive splats hold more than GSR_RANK_IMBALANCE x the mean's tile instances (k_rcount; posted with the count)
    uint32_t pad[21];
    BinStatSlot slot[GSR_STAT_SLOTS];
};
static_assert(sizeof(BinHeader) == 256 + 64 * GSR_STAT_SLOTS, "header layout is part of include/gsr.h");

#define GSR_WALK_MASKS 2           // saved hit masks per splat (rounds of 64 quadrants): rects beyond 128 quadrants are tested again by the scatter pass
struct QBinArgs {               // k_qcount / k_qscatter
    int Q, gx;                      // quadrants (4 * tiles), tile columns
    uint32_t chunks;
    BinHeader* __restrict__ hdr;
    const float4* __restrict__ brec;
    const uint32_t* __restrict__ order;
    uint32_t* __restrict__ qhist;           // [chunks][ceil(Q/4)] words of 4 byte counters
    const uint32_t* __restrict__ qprefix;   // [chunks][Q]
    const uint32_t* __restrict__ qstart;
    uint32_t* __restrict__ qpos;
    unsigned long long* __restrict__ qmask; // [P][GSR_WALK_MASKS] hit masks of the counting pass (by position in order[]), replayed by the scatter
    unsigned long long capacity;
};

// Bound entry (gsr_forward_bound / gsr_backward_bound, N1 of SURVEY.md 8(f)): the per-splat inputs are the model's mesh-LOCAL
// leaves and every splat is carried into world space right where it is read (bind_math.h) -- no world-space tensors, no bind launch.
// binding == nullptr: plain world-space inputs, or (leaves != 0) an UNBOUND model's leaves: the activations only (exp, normalize, sigmoid).
struct BoundDev {
    int leaves;                           // the inputs are the model's leaves (bound when binding != nullptr)
    const void* __restrict__ binding;     // (P) face of every splat, int32 or int64
    int is64;
    const float* __restrict__ fR;         // (F,3,3) face_orien_mat
    const float* __restrict__ fs;         // (F)     face_scaling
    const float* __restrict__ fc;         // (F,3)   face_center
    const float* __restrict__ fq;         // (F,4)   face_orien_quat (WXYZ)
    const int* __restrict__ slot;         // backward: (P) the splat's position in the per-face CSR of the binding
    float* __restrict__ rows;             // backward: (P, BINDM_ROW) its contributions to its face's gradients, parked at that position
};
__device__ __forceinline__ long long bound_face(const BoundDev& b, int i)
{
    return b.is64 ? reinterpret_cast<const long long*>(b.binding)[i] : (long long)reinterpret_cast<const int*>(b.binding)[i];
}

struct PreprocessArgs {
    int P, M;
    const float* __restrict__ means3D;
    const float* __restrict__ shs;
    const float* __restrict__ shs_rest;   // non-null: shs is (P,1,3) and this is (P,M-1,3) (the model's two leaf tensors)
    const float* __restrict__ colors_precomp;
    const float* __restrict__ opacities;
    const float* __restrict__ scales;
    const float* __restrict__ rotations;
    const float* __restrict__ cov3D_precomp;
    int32_t* __restrict__ radii;
    float* __restrict__ depths;
    float4* __restrict__ grec;
    float* __restrict__ cov3D;
    ushort4* __restrict__ rect;
    uint32_t* __restrict__ tiles_touched;
    uint8_t* __restrict__ clamped;
    uint8_t* __restrict__ visible;        // [P] radii > 0
    float4* __restrict__ acc;             // [P][3] backward accumulators, zeroed here for visible splats
    float4* __restrict__ acc64;           // [P][5] the deterministic mode's (80 B per splat), zeroed instead when settings.deterministic
    uint32_t* __restrict__ tile_count;    // [tiles] zeroed here (the binning histogram of this frame)
    uint32_t* __restrict__ units;         // the GSR_UNIT_LISTS counters of GsrImageLayout.units are zeroed here (k_render<true> appends, k_render_bwd_rp consumes)
    unsigned long long* __restrict__ rect_total;   // zeroed here; k_count sums tiles_touched into it
    int tiles;
    // production binning (gsr_binning.hip); brec == nullptr on the per-tile sort path
    float4* __restrict__ brec;            // [P][3] binning record
    BinHeader* __restrict__ hdr;          // zeroed by the API before this kernel; statistics accumulated here
    uint32_t* __restrict__ bcount;        // [nb] depth-bucket histogram, zeroed here
    int nb;
    // rank path (gsr_rank.hip); pstat == nullptr otherwise
    uint32_t* __restrict__ bcursor;       // [nb] bucket fill cursors, zeroed here
    uint4* __restrict__ pstat;            // [workgroups] (min, max) depth bits of this workgroup's visible splats ((~0, 0) when it has none), the tile instances its splats are binned into, 0;
                                          // then four planes [workgroups] of uint4: the instance sums of the workgroup's sixteen 16-splat groups (plane q: groups 4 q .. 4 q + 3), then k_rcount's chunk boundaries
    ushort4* __restrict__ srect;          // [P] tile rect the splat is binned into (snug when cull != 0); zero area = not binned
    float4* __restrict__ sspan;           // [P][2] the splat's Span (px, py, B, det | twoTA, A, dyr, mode): operands of the quadrant test
    int cull;                             // settings.tile_culling != 0
    BoundDev bound;                       // means3D / scales / rotations / opacities are _xyz / _scaling / _rotation / _opacity when bound.binding
};

struct PreBwdArgs {
    int P, M;
    const float* __restrict__ means3D;
    const float* __restrict__ shs;
    const float* __restrict__ shs_rest;
    const float* __restrict__ scales;
    const float* __restrict__ rotations;
    const float* __restrict__ cov3D;      // forward's (P,6)
    const int32_t* __restrict__ radii;
    const uint8_t* __restrict__ clamped;
    float* __restrict__ acc;              // [P][12]: dcolor(3) dmean2D(2) dconic(3) dopacity(1); re-zeroed after use
    long long* __restrict__ acc64;        // deterministic mode: [P][10] the same sums in fixed point
    const uint32_t* __restrict__ gmax;    // deterministic mode: bits of max |dL/dpixel| (the fixed-point scale)
    const float4* __restrict__ grec;      // deterministic mode: the per-splat records (their fixed-point exponents)
    int use_precomp_cov, use_precomp_color;
    float* __restrict__ dL_dmeans3D;
    float* __restrict__ dL_dmeans2D;      // (P,3)
    float* __restrict__ dL_dsh;           // (P,M,3) or null   ((P,1,3) when split)
    float* __restrict__ dL_dsh_rest;      // (P,M-1,3) when split
    float* __restrict__ dL_dcolors;       // (P,3)
    float* __restrict__ dL_dopacity;      // (P,1)
    float* __restrict__ dL_dscales;       // (P,3) or null
    float* __restrict__ dL_drotations;    // (P,4) or null
    float* __restrict__ dL_dcov3D;        // (P,6)
    BoundDev bound;                       // bound entry: the four inputs above are the local leaves, the gradients those of the leaves
    const float* __restrict__ opacities;  // bound entry: the opacity logits (sigmoid' for dL_dopacity)
};

__global__ void k_preprocess(Settings s, PreprocessArgs a);
__global__ void k_tile_scan(int tiles, const uint32_t* tile_count, uint32_t* tile_start, uint32_t* tile_cursor, uint2* ranges,
                            uint32_t* tile_order, unsigned long long* total_dev, unsigned long long* mailbox, unsigned long long seq, unsigned long long post_capacity);
template <bool CULL>
__global__ void k_count(int P, int gx, int tiles, const ushort4* rect, const uint32_t* tiles_touched, const float4* grec, uint32_t* tile_count,
                        unsigned long long* rect_total, uint32_t* block_hist);
template <bool CULL>
__global__ void k_scatter(int P, int gx, int tiles, const float* depths, const ushort4* rect, const uint32_t* tiles_touched, const float4* grec,
                          const uint32_t* tile_start, uint32_t* tile_cursor, unsigned long long* keys,
                          unsigned long long capacity, const unsigned long long* total_dev, const uint32_t* block_hist);
template <int KEYS, int THREADS>
__global__ void k_tile_sort(uint32_t n_lo, uint32_t n_hi, int gx, const uint32_t* tile_order, const uint32_t* tile_count, const uint32_t* tile_start, unsigned long long* keys,
                            uint32_t* point_list, uint32_t* qlist, uint32_t* qpos, uint32_t* qcount, uint32_t* qstart, const float4* grec,
                            unsigned long long capacity, const unsigned long long* total_dev);
template <bool FAST, int CONT, bool INFER>   // INFER: forward_only frames (no last-contributor bookkeeping); CONT: 0 tiles only (fast blend: parks deep quadrants for <true, 2> when s.cont_chunks > 0), 1 tiles + waiting continuation workgroups in one grid, 2 the continuation kernel
__global__ void k_render(Settings s, const uint32_t* tile_order, const uint32_t* qstart, const uint32_t* qcount, const float4* grec,
                         const uint32_t* qpos, const uint32_t* qlist, float* final_T,
                         uint32_t* n_contrib, uint32_t* n_contrib_q, float* c_final, float4* ck, float* out_color, unsigned long long capacity,
                         const unsigned long long* total_dev, uint32_t* units, int tiles);
// the continuation area behind the unit lists of GsrImageLayout.units (k_render): header words, then the list, then the parked states
#define GSR_CONT_HDR_WORDS 576    // word 0 parked, 32 pull cursor, 64 + 32 k (k < 16) reports of the tile waves: a 128-byte line each
#ifndef GSR_CONT_CHUNKS_DEFAULT
#define GSR_CONT_CHUNKS_DEFAULT 0   // OFF: measured on the MI355X, the continuation costs more than the tail it removes on both bench scenes; GSR_CONT_CHUNKS=3 or 4 turns it on
#endif
#ifndef GSR_CONT_WAVES
#define GSR_CONT_WAVES 8   // waves per workgroup of the continuation kernel (mode 2): chunks of a quadrant in flight
#endif
#ifndef GSR_CONT_MODE_DEFAULT
#define GSR_CONT_MODE_DEFAULT 1
#endif
#ifndef GSR_CONT_GRID_DEFAULT
#define GSR_CONT_GRID_DEFAULT 768   // workgroups of the continuation kernel (three per CU); they pull the parked quadrants in turn
#endif
#define GSR_CONT_STATE_FLOATS 320
__host__ __device__ inline size_t unit_list_cap(size_t tiles) { return (4 * tiles + GSR_UNIT_LISTS - 1) / GSR_UNIT_LISTS * (size_t)GSR_BWD_SEGMENTS; }
__host__ __device__ inline size_t cont_hdr_word(size_t tiles) { return (size_t)32 * GSR_UNIT_LISTS + (size_t)GSR_UNIT_LISTS * unit_list_cap(tiles); }
__global__ void k_mark_visible(int P, const float* means3D, const float* vm, uint8_t* present);
template <bool DET, bool FAST>
__global__ void k_render_bwd(Settings s, const uint32_t* tile_order, const uint32_t* qstart, const uint32_t* qcount, const float4* grec,
                             const uint32_t* qpos, const float* final_T,
                             const uint32_t* n_contrib_q, const float* dL_dpix, float* acc, const float* c_final, const float4* ck, int tiles,
                             long long* acc64, const uint32_t* gmax, unsigned long long capacity, const unsigned long long* total_dev);
template <int WPB>   // waves per workgroup (they share nothing; the LDS tables are per wave)
__global__ void k_render_bwd_rp(Settings s, const uint32_t* qstart, const uint32_t* qcount, const float4* grec,
                                const uint32_t* qpos, const float* final_T, const uint32_t* n_contrib_q, const float* dL_dpix, float* acc,
                                const float* c_final, const float4* ck, int tiles, unsigned long long capacity, const unsigned long long* total_dev,
                                const uint32_t* units);
__global__ void k_gmax(size_t n, const float* dL_dpix, uint32_t* gmax);
// exponent e_g with 2^e_g > the float whose bits are given (0 for no gradient at all)
__device__ __forceinline__ int gmax_exponent(uint32_t gmax_bits) { return gmax_bits ? (int)((gmax_bits >> 23) & 0xFFu) - 127 + 1 : 0; }
// the splat's fixed-point exponents (see GSR_FIXED_BITS) from what the forward knows about it: low byte for the colour / position /
// opacity sums (each partial stays below 8 |dL/dpixel| per pixel), next byte for the three conic sums, whose per-pixel factor
// G dx^2 reaches 0.74 x the largest diagonal entry of the 2-D covariance = 0.74 max(A, C) / det(conic)
__device__ __forceinline__ int splat_sum_exponents(uint32_t tiles, float conA, float conB, float conC, int W, int H)
{
    const float npix = fminf((float)tiles * (float)GSR_TILE_PIX, (float)W * (float)H);
    const float det = conA * conC - conB * conB;
    const float var = fmaxf(fabsf(conA), fabsf(conC)) / fmaxf(fabsf(det), 1e-30f);
    int ea, eb;
    (void)frexpf(npix * 8.0f, &ea);                       // bound < 2^e
    (void)frexpf(npix * 8.0f * fmaxf(1.0f, var), &eb);
    return min(max(ea, 0), 127) | (min(max(eb, 0), 127) << 8);
}
__global__ void k_preprocess_bwd(Settings s, PreBwdArgs a);
// rank path (gsr_rank.hip)
// true: the grid of tile corners does not fit the LDS histogram; instances are counted / placed with L2 atomics
// the splats of one workgroup of the rank path's three chunked passes (k_rcount reserves what k_rdscatter / k_rscatter fill: the same chunks in
// all three): an even share rounded up to 32 -- round 3 rounded to 256, which at 100 k splats left 60 of the 256 workgroups without a chunk
// Round 6: the chunks are INTERLEAVED -- groups of GSR_RANK_ILV consecutive splats dealt to the workgroups in turn (group g belongs to workgroup g mod nblk) -- so that
// a run of large splats (the coarse faces of a head: hundreds of tiles each, contiguous in Morton order) is spread over every workgroup instead of landing in a few
// (k_rscatter's time is its busiest workgroup's INSTANCE count: 89 us against 23 on the template-like avatar with contiguous chunks).  A group keeps the Morton
// neighbours together, which is what merges a workgroup's entries in L2.  rank_chunk: local indices per workgroup (a multiple of the group); rank_splat: local -> splat.
// `ilv` = 0: contiguous chunks (rounds 3 - 5); > 0: groups of `ilv` splats (a power of two) dealt round-robin.  Measured (r06_e, compile-time variants):
// groups of 8 / 16 / 64 take k_rsort_rscatter 88 -> 44 / 53 / 69 us on the template-like avatar, and cost k_rcount +6 / +5 / +3 us on BOTH scenes (every
// workgroup then touches every tile: one returning atomic per non-empty (workgroup, tile) bin) -- so it is chosen per frame size from what the previous frame's
// k_preprocess measured (gsr_api.hip: rank_ilv_for; BinHeader::chunk_imbalance).
#define GSR_RANK_ILV_AUTO 8      // the group size taken when the splats' instance counts are that uneven along the splat order
// (`ilv` is passed to the kernels as its base-2 logarithm, -1 for contiguous chunks: shifts, no integer division per splat)
__host__ __device__ inline int rank_chunk(int P, int nblk, int ilv_log2)
{
    if (ilv_log2 < 0) return ((P + nblk - 1) / nblk + 31) / 32 * 32;
    const int groups = (P + (1 << ilv_log2) - 1) >> ilv_log2;
    return ((groups + nblk - 1) / nblk) << ilv_log2;
}
__host__ __device__ inline int rank_splat(int j, int blk, int nblk, int chunk, int ilv_log2)
{
    return ilv_log2 < 0 ? blk * chunk + j : ((((j >> ilv_log2) * nblk + blk) << ilv_log2) | (j & ((1 << ilv_log2) - 1)));
}
// BALANCED chunks (round 6, `ilv` == -2: frames of one band): contiguous runs of splats of equal WEIGHT -- tile instances plus GSR_RANK_SPLAT_WEIGHT per splat --
// cut at groups of 16 splats.  k_preprocess leaves the instance sums of its sixteen 16-splat groups per workgroup behind its pstat rows, k_rcount (every
// workgroup for itself: <= 1024 rows, one per thread) finds its own two boundaries and leaves them in cb[] for the passes that follow.  The time of a rank
// pass is its busiest workgroup's instance count: equal splat counts left the busiest at 1.5 x the mean on the ellipsoid head and 10 x on the template-like one.
#ifndef GSR_RANK_SPLAT_WEIGHT
#define GSR_RANK_SPLAT_WEIGHT 4
#endif
struct RankMap { int chunk, start; };   // start < 0: rank_splat's arithmetic
__device__ __forceinline__ RankMap rank_map(int P, int nblk, int blk, int ilv, const uint32_t* __restrict__ cb)
{
    if (ilv == -2) { const int s0 = (int)cb[blk]; return RankMap{(int)cb[blk + 1] - s0, s0}; }
    return RankMap{rank_chunk(P, nblk, ilv), -1};
}
__device__ __forceinline__ int rank_splat_of(int j, int blk, int nblk, RankMap m, int ilv) { return m.start >= 0 ? m.start + j : rank_splat(j, blk, nblk, m.chunk, ilv); }
__host__ __device__ inline bool rank_direct(int gx, int tiles) { return (long long)(gx + 1) * (long long)(tiles / gx + 1) > (long long)GSR_RANK_HIST_TILES; }
#define GSR_RANK_IMBALANCE 3
__global__ void k_rcount(int ilv, int P, int gx, int tiles, int pblocks, uint32_t nb, const ushort4* srect, const uint32_t* tiles_touched,
                         const float* depths, const uint4* pstat, uint32_t* tile_count, unsigned long long* rect_total, uint32_t* block_hist,
                         uint32_t* bcount, uint32_t* bhist, BinHeader* hdr, uint32_t* cb);
struct TileScanArgs {            // the tile-counter scan that rides in k_rdscatter's last workgroup (gsr_rank.hip: tile_scan_256)
    int tiles;
    const uint32_t* __restrict__ tile_count;
    uint32_t* __restrict__ tile_start; uint32_t* __restrict__ tile_cursor;
    uint2* __restrict__ ranges; uint32_t* __restrict__ tile_order; uint4* __restrict__ tdesc;
    unsigned long long* __restrict__ total_dev; unsigned long long* mailbox;
    unsigned long long seq, post_capacity;
    const BinHeader* hdr;        // (chunk_imbalance rides in bit 39 of the posted count)
};
__global__ void k_rdscatter(int ilv, int P, uint32_t nb, const ushort4* srect, const float* depths, BinHeader* hdr, const uint32_t* bcount, uint32_t* bstart,
                            uint32_t* bcursor, unsigned long long* dkeys, const uint32_t* bhist, TileScanArgs ts, const uint32_t* cb);
// bands of the rank path: 1 (the whole frame) up to GSR_RANK_MAX_SPLATS splats, else bands of *band_rows tile rows
__host__ __device__ inline int rank_bands(long long P, int gy, bool force, int* band_rows)
{
    if ((P <= (long long)GSR_RANK_MAX_SPLATS && !force) || gy <= 1) { *band_rows = gy > 0 ? gy : 1; return 1; }
    const int bt = (gy + GSR_RANK_BANDS - 1) / GSR_RANK_BANDS;
    *band_rows = bt;
    return (gy + bt - 1) / bt;
}
// bytes of LDS the balanced expansion of k_rscatter / k_rsort_rscatter stages its 64 splats per wave in (gsr_rank.hip: RscatterStage)
#define GSR_RSCATTER_STAGE_BYTES(LEAN) ((size_t)(GSR_RANK_BIN_THREADS / 64) * 64 * (((LEAN) ? 3 : 4) * 16 + 4))
struct BandTables {              // rank path with bands (gsr_rank.hip)
    uint32_t nbands;
    float inv_band_rows;
    const uint32_t* __restrict__ over;  // [P][nbands] ranks of a splat in the fifth band onwards of its rect (very tall rects only)
};
__global__ void k_band_count(const BinHeader* hdr, const uint2* obs, uint32_t nbands, uint32_t nwc, uint32_t* bandcnt);
__global__ void k_band_scan(BinHeader* hdr, uint32_t nwc, uint32_t* bandcnt);
__global__ void k_band_rank(const BinHeader* hdr, const uint2* obs, uint32_t nbands, uint32_t nwc, const uint32_t* bandcnt, uint4* rank4, uint32_t* over);
__global__ void k_rdsort(const uint32_t* bcount, const uint32_t* bstart, unsigned long long* dkeys, unsigned long long* tmp,
                         uint32_t* rank, uint2* obs, const ushort4* srect, int band_rows);
__global__ void k_rsort_rscatter(int ilv, int scatter_blocks, int P, int gx, int tiles, const ushort4* srect, const float4* sspan, const uint32_t* tile_start,
                                 uint32_t* tile_cursor, uint32_t* entries, unsigned long long capacity, const unsigned long long* total_dev,
                                 const uint32_t* block_hist, const uint32_t* bcount, const uint32_t* bstart, unsigned long long* dkeys,
                                 unsigned long long* dtmp, uint32_t* rank, int stage_off, const uint32_t* cb);
template <int G>
__global__ void k_rscatter(int ilv, int P, int gx, int tiles, BandTables bt, const ushort4* srect, const uint32_t* rank, const float4* sspan, const uint32_t* tile_start,
                           uint32_t* tile_cursor, uint2* ranks, unsigned long long capacity, const unsigned long long* total_dev, const uint32_t* block_hist,
                           int stage_off, const uint32_t* cb);
__global__ void k_tile_rank(uint32_t words, int gx, int nbands, float inv_band_rows, const uint4* tdesc, const uint2* ranks, const uint32_t* rank_of,
                            const float* depths, const BinHeader* hdr, unsigned long long* keys, uint32_t* point_list,
                            uint32_t* qlist, uint32_t* qpos, uint32_t* qcount, uint32_t* qstart, unsigned long long capacity,
                            const unsigned long long* total_dev);
// production binning (gsr_binning.hip)
__global__ void k_dbucket(int P, const uint32_t* brec_rect, const float* depths, BinHeader* hdr, uint32_t nb, uint32_t* bcount, uint32_t* bhist);
__global__ void k_dscan(uint32_t nb, const uint32_t* bcount, uint32_t* bstart, uint32_t* bcursor, uint32_t* border, BinHeader* hdr);
__global__ void k_dscatter(int P, const uint32_t* brec_rect, const float* depths, const BinHeader* hdr, uint32_t nb, const uint32_t* bstart,
                           uint32_t* bcursor, unsigned long long* dkeys, const uint32_t* bhist);
template <int KEYS, int THREADS>
__global__ void k_dsort(uint32_t n_lo, uint32_t n_hi, const uint32_t* border, const uint32_t* bcount, const uint32_t* bstart,
                        unsigned long long* dkeys, unsigned long long* tmp, uint32_t* order);
__global__ void k_qcount(QBinArgs a);
__global__ void k_qscatter(QBinArgs a);
__global__ void k_qscan(int Q, uint32_t chunks, const uint8_t* qhist, uint32_t* qprefix, uint32_t* qcount);
__global__ void k_qscan_glob(int tiles, const uint32_t* qcount, uint32_t* qstart, uint32_t* tile_order, BinHeader* hdr,
                             unsigned long long* mailbox, unsigned long long seq, unsigned long long post_capacity);

}  // namespace gsr
