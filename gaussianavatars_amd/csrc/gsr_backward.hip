// gsr_backward.hip -- backward kernels of the MI355X-native splat rasterizer (gfx950, wave64).
//
//   k_render_bwd      per tile : 4 waves x 8x8 pixels walk the tile's record stream back to front
//                                (scalar-unit loads, as in the forward); the nine per-(pixel,splat)
//                                partials are summed across the wave on the DPP network and
//                                leave as ONE 9-lane atomic burst into a 48-byte-per-splat
//                                accumulator -- 64x fewer atomics than a per-pixel formulation
//   k_preprocess_bwd  per splat: conic -> cov2D -> (cov3D, mean), projective divide, SH, and
//                                cov3D -> (scale, raw quaternion); writes every output row
//                                (zeros for culled splats) so no gradient buffer needs a memset
//
// Behavioural spec: SURVEY.md Appendix A.4.  Summation order differs from any other
// implementation (float atomics), so this path is compared with a tolerance, never bit-wise.
#include "gsr_device.h"

#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

namespace gsr {

// 64-lane sums of RB(=4) x 9 values with a transposed reduction: v_permlane32_swap halves the lane
// count and doubles the records per register, v_permlane16_swap does it again, then a 4-step DPP
// butterfly finishes inside each 16-lane row.  Afterwards row r of t[c] holds, in every lane, the full
// sum of component c of record kRowRec[r].  ~105 VALU ops for 36 sums (a plain butterfly needs 216).
__device__ __forceinline__ float swap32_add(float a, float b)
{
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);   // lanes 0-31: a summed over halves, lanes 32-63: b
}
__device__ __forceinline__ float swap16_add(float a, float b)
{
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);   // rows 0,2: a's row pairs, rows 1,3: b's
}
// 16-lane sums of FOUR registers at once: bank b (lanes 4b..4b+3 of every row) of the result holds the row sum of register b.
__device__ __forceinline__ float bank_merge(float keep, float take, const int bank_mask_sel)
{
    // lanes of the banks in the mask take `take`, the others keep `keep` (v_mov_b32_dpp with an identity quad_perm)
    return bank_mask_sel == 0xA
               ? __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(keep), __float_as_int(take), 0xE4, 0xf, 0xA, false))
               : __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(keep), __float_as_int(take), 0xE4, 0xf, 0xC, false));
}
__device__ __forceinline__ float row_sum4(float a, float b, float c, float d)
{
    a += dpp_f<0x141, 0xf>(a);  // row_half_mirror: every bank now holds a half-row's partials (banks 0,1 the same total; 2,3 too)
    b += dpp_f<0x141, 0xf>(b);
    c += dpp_f<0x141, 0xf>(c);
    d += dpp_f<0x141, 0xf>(d);
    float ab = bank_merge(a, b, 0xA);   // banks 0,2: a    banks 1,3: b
    float cd = bank_merge(c, d, 0xA);
    ab += dpp_f<0x128, 0xf>(ab);        // row_ror:8: the other half-row's partials of the same register
    cd += dpp_f<0x128, 0xf>(cd);
    float r = bank_merge(ab, cd, 0xC);  // bank 0: a, 1: b, 2: c, 3: d
    r += dpp_f<0xB1, 0xf>(r);           // quad_perm [1,0,3,2]
    r += dpp_f<0x4E, 0xf>(r);           // quad_perm [2,3,0,1]
    return r;
}
__device__ __forceinline__ float row_sum(float v)
{
    v += dpp_f<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
    v += dpp_f<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
    v += dpp_f<0x141, 0xf>(v);  // row_half_mirror
    v += dpp_f<0x140, 0xf>(v);  // row_mirror
    return v;
}

#ifdef GSR_EXPERIMENT_TIMELINE   // `make timeline`: per-wave start/end stamps of k_render_bwd for tools/bwd_timeline.py
__device__ unsigned long long gsr_dbg[4 * 16384];
extern "C" int gsr_debug_read(unsigned long long* host, int n) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(gsr_dbg), (size_t)n * 8); }
#endif
// max |dL/dpixel| for the deterministic mode's fixed-point scale (a maximum does not depend on the order it is taken in)
__global__ __launch_bounds__(256) void k_gmax(size_t n, const float* __restrict__ dL_dpix, uint32_t* __restrict__ gmax)
{
    uint32_t m = 0u;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t b = __float_as_uint(dL_dpix[i]) & 0x7FFFFFFFu;
        m = (b < 0x7F800000u && b > m) ? b : m;       // finite magnitudes order like their bit patterns
    }
    m = wave_max_u32(m);
    if ((threadIdx.x & 63) == 0 && m) atomicMax(gmax, m);
}

template <bool DET, bool FAST>
__global__ __launch_bounds__(256) void k_render_bwd(Settings s, const uint32_t* __restrict__ tile_order, const uint32_t* __restrict__ qstart,
                                                     const uint32_t* __restrict__ qcount,
                                                     const float4* __restrict__ grec, const uint32_t* __restrict__ qpos,
                                                     const float* __restrict__ final_T,
                                                     const uint32_t* __restrict__ n_contrib_q, const float* __restrict__ dL_dpix,
                                                     float* __restrict__ acc /* [P][GSR_ACC_STRIDE] */,
                                                     const float* __restrict__ c_final, const float4* __restrict__ ck, int tiles,
                                                     long long* __restrict__ acc64, const uint32_t* __restrict__ gmax,
                                                     unsigned long long capacity, const unsigned long long* __restrict__ total_dev)
{
    if (*total_dev > capacity) return;   // the forward of this state overflowed its binning capacity and skipped the frame (deferred count)
    const int e_g = DET ? gmax_exponent(*gmax) : 0;
#ifdef GSR_EXPERIMENT_TIMELINE
    const unsigned long long t_start = wall_clock64();
#endif
    const int W = s.W, H = s.H;
    const int gx = (W + GSR_BLOCK_X - 1) / GSR_BLOCK_X;
    // Workgroup = (tile, segment): a quadrant's back-to-front walk is cut into GSR_BWD_SEGMENTS pieces of GSR_BWD_SEGMENT stream
    // entries (the last piece takes the rest) and every piece runs on its own wave, starting from the (T, C) checkpoint the
    // forward left at its upper end.  The walk is strictly serial per pixel, and the kernel used to end when the deepest
    // quadrant (~600 records) ended while the average wave was done after a quarter of that.  Segment 0 of every tile is
    // dispatched first, the (ever sparser) deeper segments behind them.
    const int seg = (int)blockIdx.x / tiles;
    const int tile = (int)tile_order[(int)blockIdx.x - seg * tiles];
    const int tile_x = tile % gx, tile_y = tile / gx;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const int pxi = tile_x * GSR_BLOCK_X + (wave & 1) * 8 + (lane & 7);
    const int pyi = tile_y * GSR_BLOCK_Y + (wave >> 1) * 8 + (lane >> 3);
    const bool inside = pxi < W && pyi < H;
    const float pixx = (float)pxi, pixy = (float)pyi;
    const int nq = (int)qcount[4 * tile + wave];
    const int seg_lo = seg * GSR_BWD_SEGMENT;
    if (nq <= seg_lo) return;                    // the stream ends below this segment
    const float4* __restrict__ rec = grec;                            // the per-splat records (48 bytes each)
    const uint32_t* __restrict__ qp = qpos + qstart[4 * tile + wave];   // this quadrant's stream of splat indices

    const int pix_id = W * pyi + pxi;
    const size_t HW = (size_t)H * W;
    const float T_final = inside ? final_T[pix_id] : 0.f;
    const int last_pix = inside ? (int)n_contrib_q[pix_id] : 0;        // one past the pixel's last contributing stream entry
    const int seg_hi = seg == GSR_BWD_SEGMENTS - 1 ? 0x7fffffff : seg_lo + GSR_BWD_SEGMENT;
    const int last = last_pix > seg_lo ? min(last_pix, seg_hi) : 0;    // this wave handles entries seg_lo .. last-1 of the pixel
    const int jmax = min((int)wave_max_u32((uint32_t)last), nq);
    if (jmax == 0) return;                       // nothing of this segment in this quadrant
    float g0 = 0.f, g1 = 0.f, g2 = 0.f;
    if (inside) {
        g0 = dL_dpix[0 * HW + pix_id];
        g1 = dL_dpix[1 * HW + pix_id];
        g2 = dL_dpix[2 * HW + pix_id];
    }
    const float bg_dot = s.bg[0] * g0 + s.bg[1] * g1 + s.bg[2] * g2;
    const float neg_Tf_bg = -T_final * bg_dot;   // the background's share of dL/dalpha, times (1 - alpha) of the record

    float T = T_final;
    float Sb = 0.f;                              // (colour accumulated behind the current splat) . dL/dpixel
    float last_cg = 0.f;                         // (colour of the previously visited splat) . dL/dpixel
    float last_alpha = 0.f, last_one_m = 1.f;
    if (last_pix > seg_hi) {
        // the pixel's walk continues above this segment: start from the forward's checkpoint at entry seg_hi.  T is the
        // transmittance there; the colour behind it is what the pixel ended with minus what it had there, over T (the
        // division by T cancels against the factor T of every use, so the cancellation error stays at fp32 round-off).
        const float4 c = ck[(size_t)seg * HW + pix_id];
        T = c.x;
        const float b0 = c_final[0 * HW + pix_id] - c.y, b1 = c_final[1 * HW + pix_id] - c.z, b2 = c_final[2 * HW + pix_id] - c.w;
        Sb = (b0 * g0 + b1 * g1 + b2 * g2) / T;
    }

    constexpr int RB = 4;
    // The kernel ends when its longest walk ends (0.6 M wave-records in total, but the deepest quadrant walks ~500 of them
    // strictly in order): let the deep walks win issue arbitration over the shallow ones sharing their SIMD.
    const int walk = jmax - seg_lo;              // only the open-ended last segment can be long
    if (walk > 320) __builtin_amdgcn_s_setprio(3);
    else if (walk > 192) __builtin_amdgcn_s_setprio(2);
    else if (walk > 96) __builtin_amdgcn_s_setprio(1);
    struct Rec2 { f32x8 a[2]; float cbl[2]; int ex[2]; uint32_t id[2]; float op[2]; };   // a = (x, y, conic a, conic b, conic c, opacity, red, green)
    struct Pos2 { uint32_t p[2]; };
    // both streams through the CONSTANT address space with 32-bit byte offsets (as k_render): scalar loads, register-offset form
    typedef const __attribute__((address_space(4))) char* cbytes;
    const cbytes recb = (cbytes)(uintptr_t)rec;
    const cbytes qpb = (cbytes)(uintptr_t)qp;
    auto loadp = [&](int jp, Pos2& P) {
#pragma unroll
        for (int u = 0; u < 2; ++u) P.p[u] = *(const __attribute__((address_space(4))) uint32_t*)(qpb + min((uint32_t)(jp + u), (uint32_t)(nq - 1)) * 4u);   // one unsigned clamp: a negative index (walk over) wraps high
    };
    auto load2 = [&](const Pos2& P, Rec2& R) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const uint32_t off = P.p[u] * 48u;
            R.a[u] = *(const __attribute__((address_space(4))) f32x8*)(recb + off);
            if (DET) {   // blue and the splat's fixed-point exponent in one 8-byte load
                typedef float f32x2 __attribute__((ext_vector_type(2)));
                const f32x2 be = *(const __attribute__((address_space(4))) f32x2*)(recb + off + 32);
                R.cbl[u] = be[0];
                R.ex[u] = __float_as_int(be[1]);
            } else {
                R.cbl[u] = *(const __attribute__((address_space(4))) float*)(recb + off + 32);
                R.ex[u] = 0;
            }
            R.op[u] = R.a[u][5];
            if (FAST) R.op[u] = *(const __attribute__((address_space(4))) float*)(recb + off + 40);   // fast-blend record: slot 5 holds log2(opacity), slot 10 the opacity
            R.id[u] = P.p[u];   // the stream entry IS the splat index
        }
    };
    float v[RB][9];
    uint32_t id[RB];
    int ex[RB];
    // which component of which register a lane delivers to the accumulator after the reduction below
    const int l16 = lane & 15;
    const bool sel_v2 = (l16 & 3) == 1, sel_v3 = l16 == 2;
    const int c_sel = sel_v3 ? 8 : ((l16 & 3) == 0 ? (l16 >> 2) : (sel_v2 ? 4 + (l16 >> 2) : -1));
    // records jp, jp+1 (visited jp+1 first: back to front) -> v[uo], v[uo+1]
    auto grad2 = [&](int jp, const Rec2& R, int uo) -> bool {
        float G[2], alpha[2], dxs[2], dys[2];
        bool hit[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            dxs[u] = R.a[u][0] - pixx;
            dys[u] = R.a[u][1] - pixy;
            // hardware 2^x (v_exp_f32, ~1 ulp) instead of the forward's 13-instruction bit-reproducible polynomial: the backward
            // is compared with a tolerance, and a record whose alpha sits within an ulp of 1/255 flipping in or out is noise
            float power;
            if constexpr (FAST) {   // fast blend: the record's conic is pre-scaled by -log2(e)/2 (exactly the forward's expression)
                power = __builtin_fmaf(__builtin_fmaf(R.a[u][3], dys[u], R.a[u][2] * dxs[u]), dxs[u], (R.a[u][4] * dys[u]) * dys[u]);
                G[u] = __builtin_amdgcn_exp2f(power);
            } else {
                power = -0.5f * (R.a[u][2] * dxs[u] * dxs[u] + R.a[u][4] * dys[u] * dys[u]) - R.a[u][3] * dxs[u] * dys[u];
                G[u] = __builtin_amdgcn_exp2f(power * 1.4426950408889634f);
            }
            alpha[u] = sel_min(0.99f, R.op[u] * G[u]);
            hit[u] = (jp + u) < last && power <= 0.0f && alpha[u] >= 1.0f / 255.0f;
            id[uo + u] = R.id[u];
            ex[uo + u] = R.ex[u];
        }
        if (__ballot(hit[0] || hit[1]) == 0ull) {      // no pixel of this wave touched these two splats
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int c = 0; c < 9; ++c) v[uo + u][c] = 0.f;
            return false;
        }
        // Sequential part, written for instruction count (this wave retires one instruction per ~6.6 cycles whatever its kind).
        // A record that does not touch this pixel gets G = alpha = 0: then 1/(1-alpha) = 1, T is multiplied by exactly 1, every
        // partial below is a product with G or alpha, and the "colour behind" recurrence passes it by (its weight is the
        // record's own alpha).  That recurrence is tracked as ONE scalar, the colour behind dotted with this pixel's
        // dL/dpixel (the only way it is ever used).  The partials leave here WITHOUT their constant factors (-W/2, -H/2 on the
        // position pair, -1/2 on the conic triple): k_preprocess_bwd applies them once per splat instead of once per pixel.
#pragma unroll
        for (int u = 1; u >= 0; --u) {
            const float Gh = hit[u] ? G[u] : 0.f;
            const float al = hit[u] ? alpha[u] : 0.f;
            const float one_m = 1.f - al;
            const float rinv = __builtin_amdgcn_rcpf(one_m);
            const float Tn = T * rinv;
            const float w = al * Tn;
            const float cg = R.a[u][6] * g0 + R.a[u][7] * g1 + R.cbl[u] * g2;
            Sb = last_alpha * last_cg + last_one_m * Sb;
            const float dL_dalpha = (cg - Sb) * Tn + neg_Tf_bg * rinv;
            const float q = Gh * (R.op[u] * dL_dalpha);       // G * dL/dG
            const float ax = q * dxs[u], ay = q * dys[u];
            float* vv = v[uo + u];
            vv[0] = w * g0;
            vv[1] = w * g1;
            vv[2] = w * g2;
            if constexpr (FAST) {   // the two position sums leave without the conic: k_preprocess_bwd, which has it, forms A sx + B sy, C sy + B sx per splat
                vv[3] = ax;
                vv[4] = ay;
            } else {
                vv[3] = ax * R.a[u][2] + ay * R.a[u][3];          // x (-W/2) = dL/dmean2D.x
                vv[4] = ay * R.a[u][4] + ax * R.a[u][3];          // x (-H/2) = dL/dmean2D.y
            }
            vv[5] = ax * dxs[u];                              // x (-1/2) = dL/dconic (a, b, c)
            vv[6] = ax * dys[u];
            vv[7] = ay * dys[u];
            vv[8] = FAST ? q : Gh * dL_dalpha;                // fast blend: times the opacity (k_preprocess_bwd divides once per splat)
            T = Tn;
            last_alpha = al;
            last_one_m = one_m;
            last_cg = cg;
        }
        return true;
    };
    // Batches of RB records, back to front; inside a batch the high pair, then the low pair.  The scalar
    // loads of the NEXT pair are issued before the current pair is processed (same lgkmcnt(0) pinning as
    // the forward kernel).
    int jb = ((jmax + RB - 1) / RB) * RB - RB;
    if (jb >= seg_lo) {   // seg_lo is a multiple of RB: batches never straddle a segment boundary
        Rec2 A, B;
        Pos2 PA, PB;
        // two-level scalar fetch (as the forward): positions two pairs ahead, records one pair ahead
        loadp(jb + 2, PA);
        loadp(jb, PB);
        asm volatile("" ::"s"(PA.p[0]), "s"(PB.p[0]) : "memory");
        load2(PA, A);
        for (; jb >= seg_lo; jb -= RB) {
            asm volatile("" ::"s"(A.a[0]), "s"(PB.p[0]) : "memory");
            load2(PB, B);
            loadp(jb - RB + 2, PA);          // clamped inside loadp when the walk is about to end
            const bool h1 = grad2(jb + 2, A, 2);
            asm volatile("" ::"s"(B.a[0]), "s"(PA.p[0]) : "memory");
            if (jb - RB >= seg_lo) load2(PA, A);
            loadp(jb - RB, PB);
            const bool h0 = grad2(jb, B, 0);
            if (!(h0 || h1)) continue;
            // Transposed reduction.  Lane halves, then row pairs (permlane swaps): afterwards row r of t[c] holds 16 partial
            // sums of component c of record kRowRec[r] = {0, 2, 1, 3}[r].  Inside the rows the transposition goes on at bank
            // (4-lane) granularity: two registers share one after the half-row step, four after the row step, so the 16-lane
            // sums of four components cost 11 DPP operations instead of 16, and bank b of the result holds component b.
            float t[9];
#pragma unroll
            for (int c = 0; c < 9; ++c) {
                const float s01 = swap32_add(v[0][c], v[1][c]);
                const float s23 = swap32_add(v[2][c], v[3][c]);
                t[c] = swap16_add(s01, s23);
            }
            const float V1 = row_sum4(t[0], t[1], t[2], t[3]);
            const float V2 = row_sum4(t[4], t[5], t[6], t[7]);
            const float V3 = row_sum(t[8]);
            float mine = V1;                       // lanes 0,4,8,12 of a row: components 0..3
            mine = sel_v2 ? V2 : mine;             // lanes 1,5,9,13: components 4..7
            mine = sel_v3 ? V3 : mine;             // lane 2: component 8
            const int row = lane >> 4;
            uint32_t myid = id[0];
            myid = (row == 1) ? id[2] : myid;
            myid = (row == 2) ? id[1] : myid;
            myid = (row == 3) ? id[3] : myid;
            int myex = ex[0];
            if (DET) {
                myex = (row == 1) ? ex[2] : myex;
                myex = (row == 2) ? ex[1] : myex;
                myex = (row == 3) ? ex[3] : myex;
            }
            const int rec_u = (row == 1) ? 2 : ((row == 2) ? 1 : row);
            if (c_sel >= 0 && (jb + rec_u) < nq && mine != 0.f) {
                if (DET)   // fixed point: integer adds commute, the sum is the same whatever order the waves arrive in
                    atomicAdd((unsigned long long*)(acc64 + (size_t)GSR_ACC64_STRIDE * myid + c_sel),
                              (unsigned long long)__builtin_llrintf(__builtin_ldexpf(mine, GSR_FIXED_BITS - e_g - ((c_sel >= 5 && c_sel <= 7) ? (myex >> 8) : (myex & 0xFF)))));
                else
                    unsafeAtomicAdd(acc + (size_t)GSR_ACC_STRIDE * myid + c_sel, mine);
            }
        }
    }
#ifdef GSR_EXPERIMENT_TIMELINE
    if (lane == 0 && (size_t)blockIdx.x * 4 + wave < 16384) {
        unsigned long long* d = gsr_dbg + 4 * ((size_t)blockIdx.x * 4 + wave);
        d[0] = t_start;
        d[1] = wall_clock64();
        d[2] = (unsigned long long)jmax;
        d[3] = 0ull;
    }
#endif
}

template __global__ void k_render_bwd<false, true>(Settings, const uint32_t*, const uint32_t*, const uint32_t*, const float4*, const uint32_t*, const float*,
                                                   const uint32_t*, const float*, float*, const float*, const float4*, int, long long*, const uint32_t*,
                                                   unsigned long long, const unsigned long long*);
template __global__ void k_render_bwd<false, false>(Settings, const uint32_t*, const uint32_t*, const uint32_t*, const float4*, const uint32_t*, const float*,
                                             const uint32_t*, const float*, float*, const float*, const float4*, int, long long*, const uint32_t*,
                                             unsigned long long, const unsigned long long*);
template __global__ void k_render_bwd<true, false>(Settings, const uint32_t*, const uint32_t*, const uint32_t*, const float4*, const uint32_t*, const float*,
                                            const uint32_t*, const float*, float*, const float*, const float4*, int, long long*, const uint32_t*,
                                            unsigned long long, const unsigned long long*);

// ------------------------------------------------------------------------------------------
// k_render_bwd_rp -- the fast blend's backward (GsrSettings.fast_blend), lanes = RECORDS.
//
// The walk above gives every lane a pixel and every record a turn: 64 pixels x (73 VALU + 19 scalar instructions) per record, a
// third of them the 64-lane reduction of the nine per-splat sums.  Here the roles are swapped: a wave still owns (quadrant, 60-entry
// segment of its stream), but its LANES hold the segment's 60 records (one 48-byte gather per lane and segment, highest entry in lane
// 0: back to front in lane order), and the quadrant's pixels take turns, four at a time (half a pixel row).  What was a serial
// recurrence over the records becomes two wave scans per pixel,
//     T_k   = T_end * prod_{i<=k} 1/(1 - alpha_i)               (transmittance in front of record k: prefix product over lanes)
//     S_k   = S_end + sum_{i<k} alpha_i T_i (c_i . dL/dC)       (what lies behind record k, dotted with the pixel's gradient)
// started from the forward's checkpoint at the segment's upper end exactly as the pixel-parallel kernel starts its walk; what was a
// 64-lane reduction per record disappears -- every lane keeps the nine sums of ITS record in registers and adds them to the splat's
// accumulator once per segment.  Per-pixel operands are wave-uniform (one LDS table per wave, broadcast reads); two pixels share every
// arithmetic instruction through packed fp32 (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32), the scans of four pixels are issued
// interleaved so that the DPP read-after-write distance is covered by useful instructions.
// Same results as the pixel-parallel fast kernel up to summation order (tests/test_fast_blend_gpu.py holds both to the oracle).
// ------------------------------------------------------------------------------------------
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

// inclusive prefix scans over the 64 lanes of FOUR registers at once (in place).  v_OP_dpp dst, dst, dst: dst = shifted(dst) OP dst where
// the shift has a source lane, unchanged elsewhere (bound_ctrl off disables the write) -- one instruction per step and register, which
// the compiler will not form from the builtin for a multiply (its "old" operand has to be the identity, re-materialised per step).
// A DPP operand needs two wait states behind the VALU write of the same register: the four chains provide three.
#define GSR_SCAN4_STEP(OP, CTRL)                                      \
    OP " %0, %0, %0 " CTRL "\n\t" OP " %1, %1, %1 " CTRL "\n\t" OP " %2, %2, %2 " CTRL "\n\t" OP " %3, %3, %3 " CTRL "\n\t"
#define GSR_SCAN4(OP, a, b, c, d)                                                                      \
    asm volatile("s_nop 1\n\t" GSR_SCAN4_STEP(OP, "row_shr:1 row_mask:0xf bank_mask:0xf")              \
                 GSR_SCAN4_STEP(OP, "row_shr:2 row_mask:0xf bank_mask:0xf")                            \
                 GSR_SCAN4_STEP(OP, "row_shr:4 row_mask:0xf bank_mask:0xf")                            \
                 GSR_SCAN4_STEP(OP, "row_shr:8 row_mask:0xf bank_mask:0xf")                            \
                 GSR_SCAN4_STEP(OP, "row_bcast:15 row_mask:0xa bank_mask:0xf")                         \
                 GSR_SCAN4_STEP(OP, "row_bcast:31 row_mask:0xc bank_mask:0xf")                         \
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d))

#ifdef GSR_EXP_RP_WAVES
__attribute__((amdgpu_waves_per_eu(GSR_EXP_RP_WAVES, GSR_EXP_RP_WAVES)))
#endif
template <int WPB>
__global__ __launch_bounds__(64 * WPB) void k_render_bwd_rp(Settings s, const uint32_t* __restrict__ qstart,
                                                        const uint32_t* __restrict__ qcount, const float4* __restrict__ grec,
                                                        const uint32_t* __restrict__ qpos, const float* __restrict__ final_T,
                                                        const uint32_t* __restrict__ n_contrib_q, const float* __restrict__ dL_dpix,
                                                        float* __restrict__ acc /* [P][GSR_ACC_STRIDE] */, const float* __restrict__ c_final,
                                                        const float4* __restrict__ ck, int tiles, unsigned long long capacity,
                                                        const unsigned long long* __restrict__ total_dev, const uint32_t* __restrict__ units)
{
    if (*total_dev > capacity) return;
    __shared__ __attribute__((aligned(16))) float tab_all[WPB][32 * 12];   // per wave: the pixel table (below)
    __shared__ float xpose_all[WPB][64 * 9];                                // per wave: the nine sums of every lane on their way out
    constexpr int CH = GSR_BWD_SEGMENT;          // records per chunk = lanes in use (60 of 64)
    static_assert(CH <= 64 && CH > 32, "a chunk of the stream has to fit the wave");
    const int W = s.W, H = s.H;
    const int gx = (W + GSR_BLOCK_X - 1) / GSR_BLOCK_X;
    const int wave_in_wg = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    // ---- the wave's UNITS (gsr.h: GsrImageLayout.units): entries p, p + stride, ... of list (wave id mod GSR_UNIT_LISTS); every unit is a
    // (tile, quadrant, segment) the forward found a contributor in, so a wave that gets one has work (round 3 launched a workgroup per (tile,
    // segment): four of five found nothing, and the waves of the others were tied to the four quadrants whether those reached the segment or not)
    // (the waves of a workgroup share nothing: the launch decides how many there are -- gsr_api.hip)
    constexpr uint32_t wpb = (uint32_t)WPB;
    const uint32_t wid = (uint32_t)blockIdx.x * wpb + (uint32_t)wave_in_wg, list = wid % (uint32_t)GSR_UNIT_LISTS;
    const uint32_t ustride = (uint32_t)gridDim.x * wpb / (uint32_t)GSR_UNIT_LISTS;
    const uint32_t ucap = (4u * (uint32_t)tiles + GSR_UNIT_LISTS - 1u) / GSR_UNIT_LISTS * (uint32_t)GSR_BWD_SEGMENTS;
    const uint32_t ucount = min(units[32u * list], ucap);
    const uint32_t* __restrict__ ulist = units + 32u * GSR_UNIT_LISTS + list * ucap;
    float* const tab = tab_all[wave_in_wg];
    float* const xs = xpose_all[wave_in_wg];
    for (uint32_t up = wid / (uint32_t)GSR_UNIT_LISTS; up < ucount; up += ustride) {
    // (from the END of the list: the forward appends a quadrant's units when its walk is over, so the deep quadrants -- whose open-ended last segment
    //  is many chunks for one wave -- come last in the list; taken first they are not the kernel's tail: template-like frame 109.8 -> see DESIGN.md 7.6)
#ifndef GSR_EXP_RP_FORWARD_ORDER
    const uint32_t unit = (uint32_t)__builtin_amdgcn_readfirstlane((int)ulist[ucount - 1u - up]);
#else
    const uint32_t unit = (uint32_t)__builtin_amdgcn_readfirstlane((int)ulist[up]);
#endif
    const int seg = (int)(unit & 15u), wave = (int)((unit >> 4) & 3u), tile = (int)(unit >> 6);   // `wave`: the quadrant
    const int tile_x = tile % gx, tile_y = tile / gx;
    const int nq = (int)qcount[4 * tile + wave];
    const int seg_lo = seg * CH;
    if (nq <= seg_lo) continue;
    const uint32_t* __restrict__ qp = qpos + qstart[4 * tile + wave];

    // ---- this lane's PIXEL (lane = 8 * row + column of the quadrant): what the pixel contributes to the table below
    const int qx0 = tile_x * GSR_BLOCK_X + (wave & 1) * 8, qy0 = tile_y * GSR_BLOCK_Y + (wave >> 1) * 8;
    const int pxi = qx0 + (lane & 7), pyi = qy0 + (lane >> 3);
    const bool inside = pxi < W && pyi < H;
    const int pix_id = W * pyi + pxi;
    const size_t HW = (size_t)H * W;
    const int last_pix = inside ? (int)n_contrib_q[pix_id] : 0;        // one past the pixel's last contributing stream entry
    const int jtop = (int)wave_max_u32((uint32_t)last_pix);
    if (jtop <= seg_lo) continue;                                      // (cannot happen: the forward listed the unit because something reaches it)
    const bool open_ended = seg == GSR_BWD_SEGMENTS - 1;               // the last segment takes whatever is left, a chunk at a time
    int hi = open_ended ? seg_lo + ((jtop - seg_lo + CH - 1) / CH) * CH : seg_lo + CH;
    {
        float g0 = 0.f, g1 = 0.f, g2 = 0.f, Tend = 0.f, Sx = 0.f;
        if (inside) {
            g0 = dL_dpix[0 * HW + pix_id];
            g1 = dL_dpix[1 * HW + pix_id];
            g2 = dL_dpix[2 * HW + pix_id];
            Tend = final_T[pix_id];
            Sx = Tend * (s.bg[0] * g0 + s.bg[1] * g1 + s.bg[2] * g2);   // the background's share of what lies behind every record
            if (!open_ended && last_pix > hi) {
                // the pixel's stream goes on above this segment: the forward's checkpoint at entry hi gives the transmittance there and, as
                // (final colour - colour so far) . dL/dC, everything that lies behind it
                const float4 c = ck[(size_t)seg * HW + pix_id];
                Tend = c.x;
                Sx += (c_final[0 * HW + pix_id] - c.y) * g0 + (c_final[1 * HW + pix_id] - c.z) * g1 + (c_final[2 * HW + pix_id] - c.w) * g2;
            }
        }
        // ---- the wave's pixel table (LDS, 12 floats per PAIR of horizontally adjacent pixels, three broadcast ds_read_b128 per pair):
        //      (T_end, T_end', S, S' | g0, g0', g1, g1' | g2, g2', last, last')
        float* tw = tab + ((lane >> 3) * 4 + ((lane & 7) >> 1)) * 12 + (lane & 1);
        tw[0] = Tend; tw[2] = Sx; tw[4] = g0; tw[6] = g1; tw[8] = g2; tw[10] = __int_as_float(last_pix);
    }
    const float fqx0 = (float)qx0, fqy0 = (float)qy0;

    for (;;) {
        const int lo = hi - CH;
        // ---- this lane's RECORD: entry hi - 1 - lane of the stream (lanes CH.. and entries past the end: opacity 0, never hit)
        const int j = hi - 1 - lane;
        const bool valid = lane < CH && j < nq;
        const uint32_t idx = qp[valid ? j : nq - 1];
        const float4 r0 = grec[3 * (size_t)idx + 0], r1 = grec[3 * (size_t)idx + 1], r2 = grec[3 * (size_t)idx + 2];
        const float xq = r0.x - fqx0, yq = r0.y - fqy0;          // relative to the quadrant's first pixel
        const float cA = r0.z, cB = r0.w, cC = r1.x;             // the conic, pre-scaled into the 2^x domain (k_preprocess)
        const float op = valid ? r2.z : 0.f;                      // (fast-blend record: slot 5 holds log2(opacity) for the forward, slot 10 the opacity)
        const float cr = r1.z, cgn = r1.w, cb = r2.x;
        const int jrec = valid ? j : 0x7fffffff;                 // "last > jrec" is the pixel's range test
        const unsigned long long act = __ballot(last_pix > lo);  // pixels with something in this chunk
        const bool more = lo > seg_lo;                           // (open-ended segment) another chunk follows below this one
        f2 a0 = {0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0, a6 = a0, a7 = a0, a8 = a0;

        for (int r = 0; r < 8; ++r) {
            const uint32_t rm = (uint32_t)(act >> (8 * r)) & 0xFFu;
            if (!rm) continue;
            const float dy = yq - (float)r;
            const float Bdy = cB * dy, Cdy2 = (cC * dy) * dy;
#ifndef GSR_EXP_NO_ROW_SUMS
            // the six position moments of tq = G dL/dG: dy is the same for the row's eight pixels, so the row adds up (sum tq, sum tq dx, sum tq dx^2)
            // and the three dy-weighted sums are formed once per row (4 packed operations per pixel pair instead of 8)
            f2 rs0 = {0.f, 0.f}, rs1 = rs0, rs2 = rs0;
#endif
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (!((rm >> (4 * h)) & 0xFu)) continue;
                const f4* tp = reinterpret_cast<const f4*>(tab + (r * 4 + 2 * h) * 12);
                f4 t[2][3];
#pragma unroll
                for (int e = 0; e < 2; ++e)
#pragma unroll
                    for (int k = 0; k < 3; ++k) t[e][k] = tp[3 * e + k];
                f2 dx[2], a0m[2], al[2], rv[2], R[2], Tj[2], cg[2], aT[2], Pw[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const float c0 = (float)(4 * h + 2 * e);
                    dx[e] = f2{xq - c0, xq - (c0 + 1.0f)};
                    const f2 pw = (cA * dx[e] + Bdy) * dx[e] + Cdy2;
                    const f2 G = {__builtin_amdgcn_exp2f(pw.x), __builtin_amdgcn_exp2f(pw.y)};
                    const f2 araw = op * G;
                    const int l0 = __float_as_int(t[e][2].z), l1 = __float_as_int(t[e][2].w);
#ifdef GSR_EXP_HI_TEST
                    a0m[e].x = (l0 > jrec && pw.x <= 0.0f && araw.x >= 1.0f / 255.0f) ? araw.x : 0.0f;
                    a0m[e].y = (l1 > jrec && pw.y <= 0.0f && araw.y >= 1.0f / 255.0f) ? araw.y : 0.0f;
#else
                    a0m[e].x = (l0 > jrec && araw.x >= 1.0f / 255.0f) ? araw.x : 0.0f;
                    a0m[e].y = (l1 > jrec && araw.y >= 1.0f / 255.0f) ? araw.y : 0.0f;
#endif
                    al[e] = f2{__builtin_fminf(0.99f, a0m[e].x), __builtin_fminf(0.99f, a0m[e].y)};
                    const f2 om = 1.0f - al[e];
                    rv[e] = f2{__builtin_amdgcn_rcpf(om.x), __builtin_amdgcn_rcpf(om.y)};
                    R[e] = rv[e];
                }
#ifndef GSR_EXP_RP_NOSCAN
                GSR_SCAN4("v_mul_f32_dpp", R[0].x, R[0].y, R[1].x, R[1].y);
#endif
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const f2 Te = {t[e][0].x, t[e][0].y};
                    const f2 g0 = {t[e][1].x, t[e][1].y}, g1 = {t[e][1].z, t[e][1].w}, g2 = {t[e][2].x, t[e][2].y};
                    Tj[e] = Te * R[e];
                    cg[e] = cr * g0 + (cgn * g1 + cb * g2);
                    aT[e] = al[e] * Tj[e];
                    Pw[e] = aT[e] * cg[e];      // w = alpha T (c . dL/dC); scanned in place
                }
#ifndef GSR_EXP_RP_NOSCAN
                GSR_SCAN4("v_add_f32_dpp", Pw[0].x, Pw[0].y, Pw[1].x, Pw[1].y);
#endif
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const f2 Sx = {t[e][0].z, t[e][0].w};
                    const f2 g0 = {t[e][1].x, t[e][1].y}, g1 = {t[e][1].z, t[e][1].w}, g2 = {t[e][2].x, t[e][2].y};
                    // dL/dalpha = T c.g - S / (1 - alpha), S = what lies behind the record = (inclusive sum - w) + S_end; with w = alpha T c.g and
                    // 1 + alpha / (1 - alpha) = 1 / (1 - alpha) that is (T c.g - inclusive sum - S_end) / (1 - alpha)
                    const f2 dLda = ((Tj[e] * cg[e] - Pw[e]) - Sx) * rv[e];
                    const f2 tq = a0m[e] * dLda;                        // opacity * G * dL/dalpha  (= G * dL/dG; the 0.99 clamp has no gradient)
                    a0 += aT[e] * g0;
                    a1 += aT[e] * g1;
                    a2 += aT[e] * g2;
#ifndef GSR_EXP_NO_ROW_SUMS
                    const f2 ax = tq * dx[e];
                    rs0 += tq;
                    rs1 += ax;
                    rs2 += ax * dx[e];
#else
                    const f2 ax = tq * dx[e], ay = tq * dy;
                    a3 += ax;                                           // k_preprocess_bwd forms A sx + B sy, C sy + B sx and the constant factors
                    a4 += ay;
                    a5 += ax * dx[e];
                    a6 += ax * dy;
                    a7 += ay * dy;
                    a8 += tq;                                           // / opacity, once per splat
#endif
                    if (more) {
                        // state at the chunk's lower end for the chunk below: lane CH-1 holds the lowest entry -- the transmittance in front of it
                        // and, inclusive of it, the sum behind
                        const float T0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(Tj[e].x), CH - 1));
                        const float T1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(Tj[e].y), CH - 1));
                        const float S0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(Pw[e].x), CH - 1)) + Sx.x;
                        const float S1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(Pw[e].y), CH - 1)) + Sx.y;
                        if (lane == 0) {
                            float* up = tab + (r * 4 + 2 * h + e) * 12;
                            up[0] = T0; up[1] = T1; up[2] = S0; up[3] = S1;
                        }
                    }
                }
            }
#ifndef GSR_EXP_NO_ROW_SUMS
            a3 += rs1;                                                  // k_preprocess_bwd forms A sx + B sy, C sy + B sx and the constant factors
            a4 += rs0 * dy;
            a5 += rs2;
            a6 += rs1 * dy;
            a7 += rs0 * (dy * dy);
            a8 += rs0;                                                  // / opacity, once per splat
#endif
        }
        // ---- nine sums per record.  A lane adding its own nine would send 60 different cache lines per instruction through the L2 atomic
        // units (measured: the kernel three times slower than the walk it replaces); instead the sums cross the wave through LDS and leave
        // record-major -- lane e of pass i carries element 64 i + e of the (record, component) matrix, so each record's nine floats are
        // consecutive lanes of one instruction and reach its 48-byte accumulator as one burst.
#ifndef GSR_EXP_RP_NOATOMIC
        {
            const float v[9] = {a0.x + a0.y, a1.x + a1.y, a2.x + a2.y, a3.x + a3.y, a4.x + a4.y, a5.x + a5.y, a6.x + a6.y, a7.x + a7.y, a8.x + a8.y};
#pragma unroll
            for (int c = 0; c < 9; ++c) xs[9 * lane + c] = v[c];
            // seven records (63 lanes) per pass: a record's nine floats never straddle two instructions
#pragma unroll
            for (int i = 0; i < (CH + 6) / 7; ++i) {
                const int rl7 = (lane * 7282) >> 16;                             // lane / 9 (0..7)
                const int c = lane - 9 * rl7;
                const int rl = 7 * i + rl7;                                      // the lane whose record this is
                const float val = xs[63 * i + lane];
                const uint32_t rid = (uint32_t)__builtin_amdgcn_ds_bpermute(rl << 2, (int)idx);   // its splat
                const bool live = lane < 63 && rl < CH && (hi - 1 - rl) < nq;
#ifdef GSR_EXP_RP_PLAINSTORE    // timing experiment: the sums stay live, nothing is added (a store no value ever triggers)
                if (live && val == 12345.678f) acc[(size_t)GSR_ACC_STRIDE * rid + c] = val;
#else
                if (live && val != 0.f) unsafeAtomicAdd(acc + (size_t)GSR_ACC_STRIDE * rid + c, val);
#endif
            }
        }
#endif
        if (!more) break;
        hi = lo;
    }
    }   // units
}

template __global__ void k_render_bwd_rp<1>(Settings, const uint32_t*, const uint32_t*, const float4*, const uint32_t*, const float*, const uint32_t*, const float*, float*, const float*,
                                             const float4*, int, unsigned long long, const unsigned long long*, const uint32_t*);
template __global__ void k_render_bwd_rp<2>(Settings, const uint32_t*, const uint32_t*, const float4*, const uint32_t*, const float*, const uint32_t*, const float*, float*, const float*,
                                             const float4*, int, unsigned long long, const unsigned long long*, const uint32_t*);
template __global__ void k_render_bwd_rp<4>(Settings, const uint32_t*, const uint32_t*, const float4*, const uint32_t*, const float*, const uint32_t*, const float*, float*, const float*,
                                             const float4*, int, unsigned long long, const unsigned long long*, const uint32_t*);

// ------------------------------------------------------------------------------------------

__global__ __launch_bounds__(GSR_PREBWD_ROWS) void k_preprocess_bwd(Settings s, PreBwdArgs a)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int M = a.M;
    // SH coefficients in, SH gradients out: both as one coalesced stream per workgroup through LDS
    // (gsr_device.h); each thread reads its coefficient row and then overwrites it with the gradient row.
    __shared__ float sh_lds[GSR_PREBWD_ROWS * GSR_SH_MAX_STRIDE];
    const bool sh_staged = !a.use_precomp_color && a.dL_dsh != nullptr && M <= 16;
    const int first = blockIdx.x * GSR_PREBWD_ROWS, rows = min(GSR_PREBWD_ROWS, a.P - first);
    const int sh_stride = sh_row_stride(M);
    if (sh_staged) {
        if (a.shs_rest) {   // the model's two leaf tensors: DC (P,1,3) and rest (P,M-1,3)
            sh_rows_load<GSR_PREBWD_ROWS>(sh_lds, a.shs, first, rows, 3, sh_stride, 0, (int)threadIdx.x);
            sh_rows_load<GSR_PREBWD_ROWS>(sh_lds, a.shs_rest, first, rows, 3 * (M - 1), sh_stride, 3, (int)threadIdx.x);
        } else {
            sh_rows_load<GSR_PREBWD_ROWS>(sh_lds, a.shs, first, rows, 3 * M, sh_stride, 0, (int)threadIdx.x);
        }
        __syncthreads();
    }
    if (i < a.P) {
    float dmean[3] = {0.f, 0.f, 0.f};
    float gcov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float dscale[3] = {0.f, 0.f, 0.f};
    float drot[4] = {0.f, 0.f, 0.f, 0.f};
    float gcol[3] = {0.f, 0.f, 0.f};
    float g2x = 0.f, g2y = 0.f, gop = 0.f;
    const bool vis = a.radii[i] > 0;
    float* gsh = a.dL_dsh ? (sh_staged ? sh_lds + (int)threadIdx.x * sh_row_stride(M) : a.dL_dsh + (size_t)3 * M * i) : nullptr;

    if (vis) {
        float4 q0, q1, q2;
        if (s.deterministic) {   // fixed-point sums -> fp32 (exact scaling by a power of two after the int -> float rounding)
            long long* ai = a.acc64 + (size_t)GSR_ACC64_STRIDE * i;
            const int ex = __float_as_int(a.grec[3 * (size_t)i + 2].y), back = gmax_exponent(*a.gmax) - GSR_FIXED_BITS;
            float f[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) { f[k] = __builtin_ldexpf((float)ai[k], back + ((k >= 5 && k <= 7) ? (ex >> 8) : (ex & 0xFF))); ai[k] = 0ll; }
            q0 = make_float4(f[0], f[1], f[2], f[3]); q1 = make_float4(f[4], f[5], f[6], f[7]); q2 = make_float4(f[8], 0.f, 0.f, 0.f);
        } else {
            float4* ac4 = (float4*)(a.acc + (size_t)GSR_ACC_STRIDE * i);
            q0 = ac4[0]; q1 = ac4[1]; q2 = ac4[2];
            // leave the accumulators zeroed: the state is ready for another backward
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            ac4[0] = z; ac4[1] = z; ac4[2] = z;
        }
        gcol[0] = q0.x; gcol[1] = q0.y; gcol[2] = q0.z;
        // k_render_bwd leaves the constant factors of these five to us (once per splat instead of once per pixel)
        g2x = -0.5f * (float)s.W * q0.w; g2y = -0.5f * (float)s.H * q1.x;
        const float gA = -0.5f * q1.y, gB = -0.5f * q1.z, gC = -0.5f * q1.w;
        gop = q2.x;
        const float* vm = s.viewmatrix;
        const float* proj = s.projmatrix;
        const int W = s.W, H = s.H;
        const float fx = (float)W / (2.0f * s.tanfovx), fy = (float)H / (2.0f * s.tanfovy);
        float mx = a.means3D[3 * i], my = a.means3D[3 * i + 1], mz = a.means3D[3 * i + 2];
        if (a.bound.binding) {   // bound entry: recompute the world-space position from the local leaf (bind_math.h)
            const long long bface = bound_face(a.bound, i);
            float w[3];
            bindm::world_xyz(a.bound.fR + 9 * bface, a.bound.fs[bface], a.bound.fc + 3 * bface, mx, my, mz, w);
            mx = w[0]; my = w[1]; mz = w[2];
        }
        const float* c6 = a.cov3D + 6 * i;

        // ---- SH: coefficients and view direction (FIRST: its 48 coefficient registers are dead before the geometric chain below builds
        // up its own -- 90 VGPRs instead of 130 with the two blocks the other way round) --------------------------------------------
        if (!a.use_precomp_color) {
            // the coefficient row is copied to registers first: in staged mode the gradient row overwrites it in LDS
            float shr[48];
            {
                const float* row = sh_staged ? sh_lds + (int)threadIdx.x * sh_row_stride(M) : a.shs + (size_t)3 * M * i;
#pragma unroll
                for (int k = 0; k < 48; ++k) shr[k] = k < 3 * M ? row[k] : 0.f;
            }
            const float* sh = shr;
            const int deg = s.sh_degree;
            const float d0 = mx - s.campos[0], d1 = my - s.campos[1], d2 = mz - s.campos[2];
            const float sum2 = d0 * d0 + d1 * d1 + d2 * d2;
            const float inv_len = 1.0f / sqrtf(sum2);
            const float x = d0 * inv_len, y = d1 * inv_len, z = d2 * inv_len;
            const uint32_t cl = a.clamped[i];
            float ddir[3] = {0.f, 0.f, 0.f};
            const float xx = x * x, yy = y * y, zz = z * z, xy_ = x * y, yz = y * z, xz = x * z;
            const int ncoef = (deg + 1) * (deg + 1);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float g = ((cl >> c) & 1u) ? 0.f : gcol[c];
                float dxs = 0.f, dys = 0.f, dzs = 0.f;
                gsh[0 + c] = kC0 * g;
                if (deg > 0) {
                    gsh[3 + c] = -kC1 * y * g;
                    gsh[6 + c] = kC1 * z * g;
                    gsh[9 + c] = -kC1 * x * g;
                    dxs = -kC1 * sh[9 + c];
                    dys = -kC1 * sh[3 + c];
                    dzs = kC1 * sh[6 + c];
                    if (deg > 1) {
                        gsh[12 + c] = kC2_0 * xy_ * g;
                        gsh[15 + c] = kC2_1 * yz * g;
                        gsh[18 + c] = kC2_2 * (2.f * zz - xx - yy) * g;
                        gsh[21 + c] = kC2_3 * xz * g;
                        gsh[24 + c] = kC2_4 * (xx - yy) * g;
                        dxs += kC2_0 * y * sh[12 + c] + kC2_2 * 2.f * -x * sh[18 + c] + kC2_3 * z * sh[21 + c] + kC2_4 * 2.f * x * sh[24 + c];
                        dys += kC2_0 * x * sh[12 + c] + kC2_1 * z * sh[15 + c] + kC2_2 * 2.f * -y * sh[18 + c] + kC2_4 * 2.f * -y * sh[24 + c];
                        dzs += kC2_1 * y * sh[15 + c] + kC2_2 * 2.f * 2.f * z * sh[18 + c] + kC2_3 * x * sh[21 + c];
                        if (deg > 2) {
                            gsh[27 + c] = kC3_0 * y * (3.f * xx - yy) * g;
                            gsh[30 + c] = kC3_1 * xy_ * z * g;
                            gsh[33 + c] = kC3_2 * y * (4.f * zz - xx - yy) * g;
                            gsh[36 + c] = kC3_3 * z * (2.f * zz - 3.f * xx - 3.f * yy) * g;
                            gsh[39 + c] = kC3_4 * x * (4.f * zz - xx - yy) * g;
                            gsh[42 + c] = kC3_5 * z * (xx - yy) * g;
                            gsh[45 + c] = kC3_6 * x * (xx - 3.f * yy) * g;
                            dxs += kC3_0 * sh[27 + c] * 3.f * 2.f * xy_ + kC3_1 * sh[30 + c] * yz + kC3_2 * sh[33 + c] * -2.f * xy_ +
                                   kC3_3 * sh[36 + c] * -3.f * 2.f * xz + kC3_4 * sh[39 + c] * (-3.f * xx + 4.f * zz - yy) +
                                   kC3_5 * sh[42 + c] * 2.f * xz + kC3_6 * sh[45 + c] * 3.f * (xx - yy);
                            dys += kC3_0 * sh[27 + c] * 3.f * (xx - yy) + kC3_1 * sh[30 + c] * xz +
                                   kC3_2 * sh[33 + c] * (-3.f * yy + 4.f * zz - xx) + kC3_3 * sh[36 + c] * -3.f * 2.f * yz +
                                   kC3_4 * sh[39 + c] * -2.f * xy_ + kC3_5 * sh[42 + c] * -2.f * yz + kC3_6 * sh[45 + c] * -3.f * 2.f * xy_;
                            dzs += kC3_1 * sh[30 + c] * xy_ + kC3_2 * sh[33 + c] * 4.f * 2.f * yz +
                                   kC3_3 * sh[36 + c] * 3.f * (2.f * zz - xx - yy) + kC3_4 * sh[39 + c] * 4.f * 2.f * xz +
                                   kC3_5 * sh[42 + c] * (xx - yy);
                        }
                    }
                }
                ddir[0] += dxs * g;
                ddir[1] += dys * g;
                ddir[2] += dzs * g;
            }
            for (int k = 3 * ncoef; k < 3 * M; ++k) gsh[k] = 0.f;   // coefficients above the active degree
            const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
            dmean[0] += ((sum2 - d0 * d0) * ddir[0] - d1 * d0 * ddir[1] - d2 * d0 * ddir[2]) * invsum32;
            dmean[1] += (-d0 * d1 * ddir[0] + (sum2 - d1 * d1) * ddir[1] - d2 * d1 * ddir[2]) * invsum32;
            dmean[2] += (-d0 * d2 * ddir[0] - d1 * d2 * ddir[1] + (sum2 - d2 * d2) * ddir[2]) * invsum32;
        }

        // ---- conic -> cov2D -> (Sigma, mean) -------------------------------------------------
        float t0 = vm[0] * mx + vm[4] * my + vm[8] * mz + vm[12];
        float t1 = vm[1] * mx + vm[5] * my + vm[9] * mz + vm[13];
        const float t2 = vm[2] * mx + vm[6] * my + vm[10] * mz + vm[14];
        const float limx = 1.3f * s.tanfovx, limy = 1.3f * s.tanfovy;
        const float txtz = t0 / t2, tytz = t1 / t2;
        t0 = sel_min(limx, sel_max(-limx, txtz)) * t2;
        t1 = sel_min(limy, sel_max(-limy, tytz)) * t2;
        const float x_grad_mul = (txtz < -limx || txtz > limx) ? 0.f : 1.f;
        const float y_grad_mul = (tytz < -limy || tytz > limy) ? 0.f : 1.f;
        const float J00 = fx / t2, J02 = -(fx * t0) / (t2 * t2), J11 = fy / t2, J12 = -(fy * t1) / (t2 * t2);
        float A[2][3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            A[0][j] = J00 * vm[4 * j + 0] + J02 * vm[4 * j + 2];
            A[1][j] = J11 * vm[4 * j + 1] + J12 * vm[4 * j + 2];
        }
        const float V[3][3] = {{c6[0], c6[1], c6[2]}, {c6[1], c6[3], c6[4]}, {c6[2], c6[4], c6[5]}};
        float AV[2][3];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int j = 0; j < 3; ++j) AV[r][j] = A[r][0] * V[0][j] + A[r][1] * V[1][j] + A[r][2] * V[2][j];
        const float ca = AV[0][0] * A[0][0] + AV[0][1] * A[0][1] + AV[0][2] * A[0][2] + 0.3f;
        const float cb = AV[0][0] * A[1][0] + AV[0][1] * A[1][1] + AV[0][2] * A[1][2];
        const float cc = AV[1][0] * A[1][0] + AV[1][1] * A[1][1] + AV[1][2] * A[1][2] + 0.3f;
        const float denom = ca * cc - cb * cb;
        if (s.fast_blend) {   // the blend handed over sum(q dx), sum(q dy): the conic (the forward's: x * det_inv) turns them into the position gradient
            const float det_inv = 1.f / denom;
            const float kA = cc * det_inv, kB = -cb * det_inv, kC = ca * det_inv;
            g2x = -0.5f * (float)s.W * (kA * q0.w + kB * q1.x);
            g2y = -0.5f * (float)s.H * (kC * q1.x + kB * q0.w);
            const float opac = a.grec[3 * (size_t)i + 2].z;   // the blend summed opacity * G * dL/dalpha (fast-blend record: the opacity sits in slot 10)
            gop = opac > 0.f ? q2.x / opac : 0.f;
        }
        const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
        float dL_da = 0.f, dL_db = 0.f, dL_dc = 0.f;
        if (denom2inv != 0.f) {
            dL_da = denom2inv * (-cc * cc * gA + 2.f * cb * cc * gB + (denom - ca * cc) * gC);
            dL_dc = denom2inv * (-ca * ca * gC + 2.f * ca * cb * gB + (denom - ca * cc) * gA);
            dL_db = denom2inv * 2.f * (cb * cc * gA - (denom + 2.f * cb * cb) * gB + ca * cb * gC);
            gcov[0] = A[0][0] * A[0][0] * dL_da + A[0][0] * A[1][0] * dL_db + A[1][0] * A[1][0] * dL_dc;
            gcov[3] = A[0][1] * A[0][1] * dL_da + A[0][1] * A[1][1] * dL_db + A[1][1] * A[1][1] * dL_dc;
            gcov[5] = A[0][2] * A[0][2] * dL_da + A[0][2] * A[1][2] * dL_db + A[1][2] * A[1][2] * dL_dc;
            gcov[1] = 2.f * A[0][0] * A[0][1] * dL_da + (A[0][0] * A[1][1] + A[0][1] * A[1][0]) * dL_db + 2.f * A[1][0] * A[1][1] * dL_dc;
            gcov[2] = 2.f * A[0][0] * A[0][2] * dL_da + (A[0][0] * A[1][2] + A[0][2] * A[1][0]) * dL_db + 2.f * A[1][0] * A[1][2] * dL_dc;
            gcov[4] = 2.f * A[0][2] * A[0][1] * dL_da + (A[0][1] * A[1][2] + A[0][2] * A[1][1]) * dL_db + 2.f * A[1][1] * A[1][2] * dL_dc;
        }
        float dA[2][3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float a0v = A[0][0] * V[j][0] + A[0][1] * V[j][1] + A[0][2] * V[j][2];
            const float a1v = A[1][0] * V[j][0] + A[1][1] * V[j][1] + A[1][2] * V[j][2];
            dA[0][j] = 2.f * a0v * dL_da + a1v * dL_db;
            dA[1][j] = 2.f * a1v * dL_dc + a0v * dL_db;
        }
        const float dJ00 = vm[0] * dA[0][0] + vm[4] * dA[0][1] + vm[8] * dA[0][2];
        const float dJ02 = vm[2] * dA[0][0] + vm[6] * dA[0][1] + vm[10] * dA[0][2];
        const float dJ11 = vm[1] * dA[1][0] + vm[5] * dA[1][1] + vm[9] * dA[1][2];
        const float dJ12 = vm[2] * dA[1][0] + vm[6] * dA[1][1] + vm[10] * dA[1][2];
        const float tz = 1.f / t2, tz2 = tz * tz, tz3 = tz2 * tz;
        const float dtx = x_grad_mul * -fx * tz2 * dJ02;
        const float dty = y_grad_mul * -fy * tz2 * dJ12;
        const float dtz = -fx * tz2 * dJ00 - fy * tz2 * dJ11 + (2.f * fx * t0) * tz3 * dJ02 + (2.f * fy * t1) * tz3 * dJ12;
        dmean[0] += vm[0] * dtx + vm[1] * dty + vm[2] * dtz;   // (+=: the SH block above has put the view direction's share there already)
        dmean[1] += vm[4] * dtx + vm[5] * dty + vm[6] * dtz;
        dmean[2] += vm[8] * dtx + vm[9] * dty + vm[10] * dtz;

        // ---- projective divide: d(ndc.xy)/d(mean) --------------------------------------------
        const float hw = proj[3] * mx + proj[7] * my + proj[11] * mz + proj[15];
        const float m_w = 1.0f / (hw + 0.0000001f);
        const float mul1 = (proj[0] * mx + proj[4] * my + proj[8] * mz + proj[12]) * m_w * m_w;
        const float mul2 = (proj[1] * mx + proj[5] * my + proj[9] * mz + proj[13]) * m_w * m_w;
        dmean[0] += (proj[0] * m_w - proj[3] * mul1) * g2x + (proj[1] * m_w - proj[3] * mul2) * g2y;
        dmean[1] += (proj[4] * m_w - proj[7] * mul1) * g2x + (proj[5] * m_w - proj[7] * mul2) * g2y;
        dmean[2] += (proj[8] * m_w - proj[11] * mul1) * g2x + (proj[9] * m_w - proj[11] * mul2) * g2y;

        // ---- Sigma -> scale, raw quaternion --------------------------------------------------
        if (!a.use_precomp_cov) {
            float4 q = reinterpret_cast<const float4*>(a.rotations)[i];
            float ws[3] = {a.scales[3 * i], a.scales[3 * i + 1], a.scales[3 * i + 2]};
            if (a.bound.binding) {
                const long long bface = bound_face(a.bound, i);
                q = bindm::world_rotation(reinterpret_cast<const float4*>(a.bound.fq)[bface], q);
                const float bscale = a.bound.fs[bface];
#pragma unroll
                for (int k = 0; k < 3; ++k) ws[k] = bindm::world_scaling(ws[k], bscale);
            } else if (a.bound.leaves) {
                q = bindm::unit_rotation(q);
#pragma unroll
                for (int k = 0; k < 3; ++k) ws[k] = bindm::world_scaling(ws[k], 1.f);
            }
            const float r = q.x, x = q.y, y = q.z, z = q.w;
            float R[3][3];
            R[0][0] = 1.f - 2.f * (y * y + z * z);
            R[0][1] = 2.f * (x * y - r * z);
            R[0][2] = 2.f * (x * z + r * y);
            R[1][0] = 2.f * (x * y + r * z);
            R[1][1] = 1.f - 2.f * (x * x + z * z);
            R[1][2] = 2.f * (y * z - r * x);
            R[2][0] = 2.f * (x * z - r * y);
            R[2][1] = 2.f * (y * z + r * x);
            R[2][2] = 1.f - 2.f * (x * x + y * y);
            const float mod = s.scale_modifier;
            const float sc[3] = {mod * ws[0], mod * ws[1], mod * ws[2]};
            const float gS[3][3] = {{gcov[0], 0.5f * gcov[1], 0.5f * gcov[2]},
                                    {0.5f * gcov[1], gcov[3], 0.5f * gcov[4]},
                                    {0.5f * gcov[2], 0.5f * gcov[4], gcov[5]}};
            float dR[3][3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                float dMk[3];
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    dMk[j] = 2.0f * sc[k] * (R[0][k] * gS[0][j] + R[1][k] * gS[1][j] + R[2][k] * gS[2][j]);
                // upstream returns the gradient w.r.t. (mod * scale); the modifier's own factor is the option (gsr.h)
                dscale[k] = (s.exact_scale_grad ? mod : 1.0f) * (R[0][k] * dMk[0] + R[1][k] * dMk[1] + R[2][k] * dMk[2]);
#pragma unroll
                for (int j = 0; j < 3; ++j) dR[j][k] = sc[k] * dMk[j];
            }
            drot[0] = 2.f * z * (dR[1][0] - dR[0][1]) + 2.f * y * (dR[0][2] - dR[2][0]) + 2.f * x * (dR[2][1] - dR[1][2]);
            drot[1] = 2.f * y * (dR[0][1] + dR[1][0]) + 2.f * z * (dR[0][2] + dR[2][0]) + 2.f * r * (dR[2][1] - dR[1][2]) - 4.f * x * (dR[2][2] + dR[1][1]);
            drot[2] = 2.f * x * (dR[0][1] + dR[1][0]) + 2.f * r * (dR[0][2] - dR[2][0]) + 2.f * z * (dR[1][2] + dR[2][1]) - 4.f * y * (dR[2][2] + dR[0][0]);
            drot[3] = 2.f * r * (dR[1][0] - dR[0][1]) + 2.f * x * (dR[0][2] + dR[2][0]) + 2.f * y * (dR[1][2] + dR[2][1]) - 4.f * z * (dR[1][1] + dR[0][0]);
        }
    } else if (gsh) {
        for (int k = 0; k < 3 * M; ++k) gsh[k] = 0.f;
    }

    if (a.bound.binding) {
        // bound entry: (dL/d world xyz, scaling, rotation, opacity) -> the gradients of the splat's own leaves, and its 17
        // contributions to its face's gradients parked at its CSR position for the per-face reduction (gab_bind_backward_faces)
        const long long bface = bound_face(a.bound, i);
        float Rf[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) Rf[k] = a.bound.fR[9 * bface + k];
        const float xl[3] = {a.means3D[3 * i], a.means3D[3 * i + 1], a.means3D[3 * i + 2]};
        const float ls[3] = {a.scales ? a.scales[3 * i] : 0.f, a.scales ? a.scales[3 * i + 1] : 0.f, a.scales ? a.scales[3 * i + 2] : 0.f};
        const float4 ql = a.rotations ? reinterpret_cast<const float4*>(a.rotations)[i] : make_float4(1.f, 0.f, 0.f, 0.f);
        float dx[3], dls[3], acc[BINDM_ROW];
        float4 dq;
        bindm::bind_backward(Rf, a.bound.fs[bface], reinterpret_cast<const float4*>(a.bound.fq)[bface], xl, ls, ql, dmean, dscale,
                             make_float4(drot[0], drot[1], drot[2], drot[3]), dx, dls, &dq, acc);
#pragma unroll
        for (int k = 0; k < 3; ++k) { dmean[k] = dx[k]; dscale[k] = dls[k]; }
        drot[0] = dq.x; drot[1] = dq.y; drot[2] = dq.z; drot[3] = dq.w;
        const float o = bindm::sigmoid(a.opacities[i]);
        gop = gop * o * (1.f - o);
        float4* row = reinterpret_cast<float4*>(a.bound.rows + (size_t)BINDM_ROW * a.bound.slot[i]);
#pragma unroll
        for (int k = 0; k < BINDM_ROW / 4; ++k) row[k] = make_float4(acc[4 * k], acc[4 * k + 1], acc[4 * k + 2], acc[4 * k + 3]);
    }
    else if (a.bound.leaves) {
        // an unbound model's leaves: the activations' chain rule (exp, normalize, sigmoid); the position is the leaf itself
        if (a.scales) {
#pragma unroll
            for (int k = 0; k < 3; ++k) dscale[k] *= bindm::world_scaling(a.scales[3 * i + k], 1.f);
        }
        if (a.rotations) {
            const float4 dq = bindm::unit_rotation_backward(reinterpret_cast<const float4*>(a.rotations)[i], make_float4(drot[0], drot[1], drot[2], drot[3]));
            drot[0] = dq.x; drot[1] = dq.y; drot[2] = dq.z; drot[3] = dq.w;
        }
        const float o = bindm::sigmoid(a.opacities[i]);
        gop = gop * o * (1.f - o);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) a.dL_dmeans3D[3 * i + k] = dmean[k];
    a.dL_dmeans2D[3 * i + 0] = g2x;
    a.dL_dmeans2D[3 * i + 1] = g2y;
    a.dL_dmeans2D[3 * i + 2] = 0.f;
    if (a.dL_dcolors) {   // (the leaves entries have no consumer for the colour / covariance gradients: 36 bytes per splat not written)
#pragma unroll
        for (int k = 0; k < 3; ++k) a.dL_dcolors[3 * i + k] = gcol[k];
    }
    a.dL_dopacity[i] = gop;
    if (a.dL_dcov3D) {
#pragma unroll
        for (int k = 0; k < 6; ++k) a.dL_dcov3D[6 * i + k] = gcov[k];
    }
    if (a.dL_dscales) {
#pragma unroll
        for (int k = 0; k < 3; ++k) a.dL_dscales[3 * i + k] = dscale[k];
    }
    if (a.dL_drotations) reinterpret_cast<float4*>(a.dL_drotations)[i] = make_float4(drot[0], drot[1], drot[2], drot[3]);
    }   // i < P
    if (sh_staged) {
        __syncthreads();
        if (a.shs_rest) {
            sh_rows_store<GSR_PREBWD_ROWS>(sh_lds, a.dL_dsh, first, rows, 3, sh_stride, 0, (int)threadIdx.x);
            sh_rows_store<GSR_PREBWD_ROWS>(sh_lds, a.dL_dsh_rest, first, rows, 3 * (M - 1), sh_stride, 3, (int)threadIdx.x);
        } else {
            sh_rows_store<GSR_PREBWD_ROWS>(sh_lds, a.dL_dsh, first, rows, 3 * M, sh_stride, 0, (int)threadIdx.x);
        }
    }
}

}  // namespace gsr
