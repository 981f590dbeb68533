// gsr_forward.hip -- forward kernels of the MI355X-native splat rasterizer (gfx950, wave64).
//
// Pipeline (one frame):
//   k_preprocess   per splat : project, EWA cov2D, conic, radius, tile rect, SH->RGB
//   k_count        <=256 WGs : tile histogram of a chunk of splats in LDS, non-empty bins -> L2 atomics
//   k_tile_scan    1 block   : exclusive scan of the per-tile counts -> tile_start / ranges / I, posts I to the
//                              host mailbox, and lists the tiles heaviest-first for the per-tile kernels
//   k_scatter      same WGs  : drops (depth bits << 32 | splat) into its tiles' segments (LDS slot allocation)
//   k_tile_sort    per tile  : register/cross-lane bitonic sort of the segment by (depth, splat) == the order a
//                              stable radix sort of (tile << 32 | depth) produces; emits, per 8x8 quadrant, a stream
//                              of splat indices culled with the exact {alpha >= 1/255} ellipse (and the
//                              reference-format sorted keys / point list in the parity modes)
//   k_render       per tile  : 4 waves x 8x8 pixels; every wave walks its quadrant's stream through the scalar unit
//                              (wave-uniform s_loads of index, then of the splat's 48-byte record; software-pipelined,
//                              no LDS, no barriers) and composites front to back
//
// Behavioural spec: SURVEY.md Appendix A.1-A.3 (the reference's rasterizer is an un-vendored
// submodule; call site gaussian_renderer/__init__.py:37-52,86-94).  This TU is built with
// -ffp-contract=off: the arithmetic below is evaluated exactly as written.
#include "gsr_device.h"
#include "gsr_sort.h"
#include <type_traits>

namespace gsr {

// ------------------------------------------------------------------------------------------
// k_preprocess
// ------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void k_preprocess(Settings s, PreprocessArgs a)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int W = s.W, H = s.H;
    const int gx = (W + GSR_BLOCK_X - 1) / GSR_BLOCK_X, gy = (H + GSR_BLOCK_Y - 1) / GSR_BLOCK_Y;
    if (a.brec) {
        // production binning: the depth-bucket histogram starts at zero (k_dbucket runs after this kernel); the header was
        // zeroed by the API before this launch and collects this kernel's statistics
        for (int t = i; t < a.nb; t += (int)(gridDim.x * blockDim.x)) a.bcount[t] = 0u;
        // the backward's work lists start empty (k_render<true> appends): the image state is a fresh, uninitialised allocation per forward
        if (i < GSR_UNIT_LISTS) a.units[32 * i] = 0u;
        if (i < GSR_CONT_HDR_WORDS / 32) a.units[cont_hdr_word((size_t)a.tiles) + 32 * i] = 0u;   // (and the continuation area's counters)
        for (int t = i; t < 4 * a.tiles; t += (int)(gridDim.x * blockDim.x)) a.units[cont_hdr_word((size_t)a.tiles) + GSR_CONT_HDR_WORDS + t] = 0xFFFFFFFFu;
    } else {
        // this frame's tile histogram starts at zero (k_count runs after this kernel)
        for (int t = i; t < a.tiles; t += (int)(gridDim.x * blockDim.x)) a.tile_count[t] = 0u;
        if (i < GSR_UNIT_LISTS) a.units[32 * i] = 0u;   // the backward's work lists start empty (k_render<true> appends)
        if (i < GSR_CONT_HDR_WORDS / 32) a.units[cont_hdr_word((size_t)a.tiles) + 32 * i] = 0u;   // ... no quadrant is parked for the continuation workgroups, none pulled, no wave has reported
        for (int t = i; t < 4 * a.tiles; t += (int)(gridDim.x * blockDim.x)) a.units[cont_hdr_word((size_t)a.tiles) + GSR_CONT_HDR_WORDS + t] = 0xFFFFFFFFu;   // (an unwritten list entry)
        if (i == 0) *a.rect_total = 0ull;
        if (a.pstat)   // rank path: the depth-bucket histogram and its fill cursors start at zero as well
            for (int t = i; t < a.nb; t += (int)(gridDim.x * blockDim.x)) { a.bcount[t] = 0u; a.bcursor[t] = 0u; }
    }

    bool visible = false;
    float depth = 0.f, px = 0.f, py = 0.f;
    float con0 = 0.f, con1 = 0.f, con2 = 0.f, opac = 0.f;
    float c6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float col[3] = {0.f, 0.f, 0.f};
    uint32_t clampbits = 0;
    int radius = 0;
    int rminx = 0, rminy = 0, rmaxx = 0, rmaxy = 0;
    int sum_exp = 0;

    if (i < a.P) {
        const float* vm = s.viewmatrix;
        const float* pm = s.projmatrix;
        float mx = a.means3D[3 * i + 0], my = a.means3D[3 * i + 1], mz = a.means3D[3 * i + 2];
        long long bface = 0;
        float bscale = 1.f;
        if (a.bound.binding) {   // bound entry: the mesh-local position goes to world space here
            bface = bound_face(a.bound, i);
            bscale = a.bound.fs[bface];
            float w[3];
            bindm::world_xyz(a.bound.fR + 9 * bface, bscale, a.bound.fc + 3 * bface, mx, my, mz, w);
            mx = w[0]; my = w[1]; mz = w[2];
        }
        // view space (3 rows of W2C) and clip space (4 rows of P*W2C); storage is transposed
        const float vx = vm[0] * mx + vm[4] * my + vm[8] * mz + vm[12];
        const float vy = vm[1] * mx + vm[5] * my + vm[9] * mz + vm[13];
        const float vz = vm[2] * mx + vm[6] * my + vm[10] * mz + vm[14];
        if (vz > 0.2f) {
            const float hx = pm[0] * mx + pm[4] * my + pm[8] * mz + pm[12];
            const float hy = pm[1] * mx + pm[5] * my + pm[9] * mz + pm[13];
            const float hw = pm[3] * mx + pm[7] * my + pm[11] * mz + pm[15];
            const float p_w = 1.0f / (hw + 0.0000001f);
            const float ndc_x = hx * p_w, ndc_y = hy * p_w;

            // ---- 3D covariance: Sigma = M^T M, M[k][j] = s_k * R[j][k], q used un-normalised
            if (a.cov3D_precomp) {
                for (int k = 0; k < 6; ++k) c6[k] = a.cov3D_precomp[6 * i + k];
            } else {
                float4 q = reinterpret_cast<const float4*>(a.rotations)[i];
                float ws[3] = {a.scales[3 * i + 0], a.scales[3 * i + 1], a.scales[3 * i + 2]};
                if (a.bound.leaves) {   // the model's leaves: activations here (bscale == 1 without a binding)
                    q = a.bound.binding ? bindm::world_rotation(reinterpret_cast<const float4*>(a.bound.fq)[bface], q) : bindm::unit_rotation(q);
#pragma unroll
                    for (int k = 0; k < 3; ++k) ws[k] = bindm::world_scaling(ws[k], bscale);
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
                const float sc[3] = {s.scale_modifier * ws[0], s.scale_modifier * ws[1], s.scale_modifier * ws[2]};
                float Mm[3][3];
#pragma unroll
                for (int k = 0; k < 3; ++k)
#pragma unroll
                    for (int j = 0; j < 3; ++j) Mm[k][j] = sc[k] * R[j][k];
                auto sig = [&](int p, int q2) { return Mm[0][p] * Mm[0][q2] + Mm[1][p] * Mm[1][q2] + Mm[2][p] * Mm[2][q2]; };
                c6[0] = sig(0, 0);
                c6[1] = sig(0, 1);
                c6[2] = sig(0, 2);
                c6[3] = sig(1, 1);
                c6[4] = sig(1, 2);
                c6[5] = sig(2, 2);
            }

            // ---- EWA: cov2D = A Sigma A^T with A = J * Rwc, tan-limit 1.3x, +0.3 low-pass
            const float fx = (float)W / (2.0f * s.tanfovx);
            const float fy = (float)H / (2.0f * s.tanfovy);
            const float limx = 1.3f * s.tanfovx, limy = 1.3f * s.tanfovy;
            const float txtz = vx / vz, tytz = vy / vz;
            const float tx = sel_min(limx, sel_max(-limx, txtz)) * vz;
            const float ty = sel_min(limy, sel_max(-limy, tytz)) * vz;
            const float J00 = fx / vz;
            const float J02 = -(fx * tx) / (vz * vz);
            const float J11 = fy / vz;
            const float J12 = -(fy * ty) / (vz * vz);
            float A[2][3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const float r0 = vm[4 * j + 0], r1 = vm[4 * j + 1], r2 = vm[4 * j + 2];
                A[0][j] = J00 * r0 + 0.0f * r1 + J02 * r2;
                A[1][j] = 0.0f * r0 + J11 * r1 + J12 * r2;
            }
            const float V[3][3] = {{c6[0], c6[1], c6[2]}, {c6[1], c6[3], c6[4]}, {c6[2], c6[4], c6[5]}};
            float AV[2][3];
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int j = 0; j < 3; ++j) AV[r][j] = A[r][0] * V[0][j] + A[r][1] * V[1][j] + A[r][2] * V[2][j];
            const float cxx = (AV[0][0] * A[0][0] + AV[0][1] * A[0][1] + AV[0][2] * A[0][2]) + 0.3f;
            const float cxy = AV[0][0] * A[1][0] + AV[0][1] * A[1][1] + AV[0][2] * A[1][2];
            const float cyy = (AV[1][0] * A[1][0] + AV[1][1] * A[1][1] + AV[1][2] * A[1][2]) + 0.3f;

            const float det = cxx * cyy - cxy * cxy;
            if (det != 0.0f) {
                const float det_inv = 1.f / det;
                const float mid = 0.5f * (cxx + cyy);
                const float lambda1 = mid + sqrtf(sel_max(0.1f, mid * mid - det));
                const float lambda2 = mid - sqrtf(sel_max(0.1f, mid * mid - det));
                const float my_radius = ceilf(3.f * sqrtf(sel_max(lambda1, lambda2)));
                const float ppx = ndc2pix(ndc_x, W), ppy = ndc2pix(ndc_y, H);
                const int rad = f2i_sat(my_radius);
                const float rf = (float)rad;
                auto clampi = [](int v, int hi) { return min(hi, max(0, v)); };
                const int x0 = clampi(f2i_sat((ppx - rf) / (float)GSR_BLOCK_X), gx);
                const int y0 = clampi(f2i_sat((ppy - rf) / (float)GSR_BLOCK_Y), gy);
                const int x1 = clampi(f2i_sat((ppx + rf + (float)GSR_BLOCK_X - 1.0f) / (float)GSR_BLOCK_X), gx);
                const int y1 = clampi(f2i_sat((ppy + rf + (float)GSR_BLOCK_Y - 1.0f) / (float)GSR_BLOCK_Y), gy);
                if ((x1 - x0) * (y1 - y0) != 0) {
                    visible = true;
                    depth = vz;
                    px = ppx;
                    py = ppy;
                    radius = rad;
                    con0 = cyy * det_inv;
                    con1 = -cxy * det_inv;
                    con2 = cxx * det_inv;
                    opac = a.bound.leaves ? bindm::sigmoid(a.opacities[i]) : a.opacities[i];
                    rminx = x0; rminy = y0; rmaxx = x1; rmaxy = y1;

                    // ---- colour
                    if (a.colors_precomp) {
                        col[0] = a.colors_precomp[3 * i + 0];
                        col[1] = a.colors_precomp[3 * i + 1];
                        col[2] = a.colors_precomp[3 * i + 2];
                    } else {
                        const int deg = s.sh_degree;
                        const float dx = mx - s.campos[0], dy = my - s.campos[1], dz = mz - s.campos[2];
                        const float len = sqrtf(dx * dx + dy * dy + dz * dz);
                        const float x = dx / len, y = dy / len, z = dz / len;
                        // 3*M contiguous floats per splat, consumed entirely by this thread: the 192-byte rows are
                        // fetched line by line through L1 (measured faster than staging them through LDS here)
                        // split layout: sh0 = the DC triple, sh = the rest block shifted so that sh[3k+c] is coefficient k
                        const float* sh0 = a.shs_rest ? a.shs + (size_t)3 * i : a.shs + (size_t)3 * a.M * i;
                        const float* sh = a.shs_rest ? a.shs_rest + (size_t)3 * (a.M - 1) * i - 3 : sh0;
                        float res[3];
#pragma unroll
                        for (int c = 0; c < 3; ++c) res[c] = kC0 * sh0[c];
                        if (deg > 0) {
#pragma unroll
                            for (int c = 0; c < 3; ++c)
                                res[c] = res[c] - kC1 * y * sh[3 + c] + kC1 * z * sh[6 + c] - kC1 * x * sh[9 + c];
                            if (deg > 1) {
                                const float xx = x * x, yy = y * y, zz = z * z;
                                const float xy_ = x * y, yz = y * z, xz = x * z;
#pragma unroll
                                for (int c = 0; c < 3; ++c)
                                    res[c] = res[c] + kC2_0 * xy_ * sh[12 + c] + kC2_1 * yz * sh[15 + c] +
                                             kC2_2 * (2.0f * zz - xx - yy) * sh[18 + c] + kC2_3 * xz * sh[21 + c] +
                                             kC2_4 * (xx - yy) * sh[24 + c];
                                if (deg > 2) {
#pragma unroll
                                    for (int c = 0; c < 3; ++c)
                                        res[c] = res[c] + kC3_0 * y * (3.0f * xx - yy) * sh[27 + c] +
                                                 kC3_1 * xy_ * z * sh[30 + c] + kC3_2 * y * (4.0f * zz - xx - yy) * sh[33 + c] +
                                                 kC3_3 * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * sh[36 + c] +
                                                 kC3_4 * x * (4.0f * zz - xx - yy) * sh[39 + c] +
                                                 kC3_5 * z * (xx - yy) * sh[42 + c] + kC3_6 * x * (xx - 3.0f * yy) * sh[45 + c];
                                }
                            }
                        }
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            res[c] += 0.5f;
                            if (res[c] < 0.0f) clampbits |= (1u << c);
                            col[c] = sel_max(res[c], 0.0f);
                        }
                    }
                }
            }
        }
        if (!visible) {
#pragma unroll
            for (int k = 0; k < 6; ++k) c6[k] = 0.f;
        }
        const uint32_t n = visible ? (uint32_t)((rmaxx - rminx) * (rmaxy - rminy)) : 0u;
        a.radii[i] = radius;
        a.depths[i] = depth;
        // .y of the third: the splat's two fixed-point exponents for the deterministic backward (gsr_device.h: GSR_FIXED_BITS), as an integer in float bits
        sum_exp = visible ? splat_sum_exponents(n, con0, con1, con2, W, H) : 0;
        // forward_only (no backward can follow): what only the backward reads -- cov3D, the clamp flags, and on the rank path (which
        // bins by srect) the rect -- is not written: 33 of ~134 bytes per splat
        if (!s.forward_only) a.clamped[i] = (uint8_t)clampbits;
        if (!s.forward_only || !a.pstat)
            a.rect[i] = make_ushort4((unsigned short)rminx, (unsigned short)rminy, (unsigned short)rmaxx, (unsigned short)rmaxy);
        a.tiles_touched[i] = n;
        a.visible[i] = radius > 0 ? (uint8_t)1 : (uint8_t)0;
    }
    // ---- the per-splat ROWS (48-byte record, 24-byte covariance, 32-byte span): a thread storing its own row writes 16 bytes into
    // each of 64 different lines per instruction, and the lines leave L2 in pieces (PMC: 355 MB written for 202 MB at 2 M splats).
    // The workgroup's rows are contiguous in memory, so they go through LDS and out as whole lines, a kilobyte per wave-instruction.
    __shared__ float4 stage[256 * 3];
    const int first = (int)(blockIdx.x * blockDim.x), rows = min(256, a.P - first), tix = (int)threadIdx.x;
    auto put_rows = [&](auto* dst, const auto* mine, auto per_row) {   // dst: global array of the row elements, mine[per_row]: this thread's row
        constexpr int K = decltype(per_row)::value;
        using E = std::remove_cv_t<std::remove_reference_t<decltype(mine[0])>>;
        E* lds = reinterpret_cast<E*>(stage);
#pragma unroll
        for (int k = 0; k < K; ++k) lds[K * tix + k] = mine[k];
        __syncthreads();
        E* out = dst + (size_t)K * first;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int e = tix + 256 * k;
            if (e < K * rows) out[e] = lds[e];
        }
        __syncthreads();
    };
    {
        // fast blend (include/gsr.h: GsrSettings.fast_blend): the blend kernels evaluate opacity * 2^(A' dx^2 + B' dx dy + C' dy^2)
        const float k2 = s.fast_blend ? -0.72134752044448170368f : 1.0f;   // -log2(e) / 2
        float4 rec[3] = {make_float4(px, py, con0 * k2, s.fast_blend ? con1 * (2.0f * k2) : con1), make_float4(con2 * k2, opac, col[0], col[1]),
                         make_float4(col[2], __int_as_float(sum_exp), 0.f, 0.f)};
        if (s.fast_blend) {
            // fast blend (round 4): the opacity enters the exponent -- alpha = 2^(A' dx^2 + B' dx dy + C' dy^2 + log2 opacity), one
            // instruction less per record and pixel than opacity * 2^(...) -- so slot 5 holds L = log2(opacity) (hardware v_log_f32),
            // slot 9 the upper bound of the blend's acceptance test 1/255 <= alpha <= 2^L (1 + 2^-18): that bound IS the reference's
            // `power <= 0` (alpha <= opacity <=> exponent <= 0), with a margin far below the stated tolerance, so that a pixel on the
            // splat's centre can never fall out by an ulp of the hardware exponential; slot 10 keeps the opacity itself for the backward.
            // An opacity that cannot reach 1/255 (or is not a number) gets L = NaN: its alpha is NaN and fails every comparison.
            const bool can = opac >= 1.0f / 255.0f;
            const float L = can ? __builtin_amdgcn_logf(opac) : __int_as_float(0x7fc00000);
            rec[1].y = L;
            rec[2].y = can ? __builtin_amdgcn_exp2f(L) * 1.000003814697265625f : 0.0f;
            rec[2].z = opac;
        }
        put_rows(a.grec, rec, std::integral_constant<int, 3>{});
    }
    if (!s.forward_only) {
        const float2 cv[3] = {make_float2(c6[0], c6[1]), make_float2(c6[2], c6[3]), make_float2(c6[4], c6[5])};
        put_rows(reinterpret_cast<float2*>(a.cov3D), cv, std::integral_constant<int, 3>{});
        // the backward's accumulators start at zero (skipped when no backward can follow): every row of the workgroup, whole lines
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        if (s.deterministic) {
            float4* out = a.acc64 + (size_t)5 * first;   // 80 bytes of fixed-point sums per splat
#pragma unroll
            for (int k = 0; k < 5; ++k) { const int e = tix + 256 * k; if (e < 5 * rows) out[e] = z; }
        } else {
            constexpr int Q = GSR_ACC_STRIDE / 4;
            float4* out = a.acc + (size_t)Q * first;
#pragma unroll
            for (int k = 0; k < Q; ++k) { const int e = tix + 256 * k; if (e < Q * rows) out[e] = z; }
        }
    }
    if (a.pstat) {
        // the tile rect this splat is binned into (snug in the culling modes) and the operands of the per-quadrant reach test
        // (gsr_device.h: band_of): computed here, once per splat, for the two binning passes that expand the rect
        float4 span2[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};   // (rows of splats that are not binned are never read)
        uint32_t ps_nt = 0;
        if (i < a.P) {
            uint32_t nt = 0;
            int minx = rminx, miny = rminy, maxx = rmaxx, maxy = rmaxy;
            if (visible) {
                nt = (uint32_t)((rmaxx - rminx) * (rmaxy - rminy));
                const Reach r = reach_of(px, py, con0, con1, con2, opac);
                if (a.cull) snug_rect(r, minx, miny, maxx, maxy, nt);
                if (nt) {
                    const Span sp = span_of(r);
                    span2[0] = make_float4(sp.px, sp.py, sp.B, sp.det);
                    span2[1] = make_float4(sp.twoTA, sp.A, sp.dyr, __int_as_float(sp.mode));
                }
            }
            a.srect[i] = nt ? make_ushort4((unsigned short)minx, (unsigned short)miny, (unsigned short)maxx, (unsigned short)maxy) : make_ushort4(0, 0, 0, 0);
            ps_nt = nt;
        }
        // ---- rank path: this workgroup's depth range and the tile instances its 256 splats are binned into (plain stores, one row per workgroup; k_rcount folds
        // them: the range maps depths to buckets, how evenly the instances are spread along the splat order decides the chunking of the NEXT frame's rank passes)
        __shared__ uint32_t ps_mn[4], ps_mx[4], ps_sum[4];
        const uint32_t dbits = __float_as_uint(depth);
        const uint32_t mn_inv = wave_max_u32(visible ? ~dbits : 0u), mx = wave_max_u32(visible ? dbits : 0u);
        const uint32_t wsum = (uint32_t)__builtin_amdgcn_readlane((int)wave_scan_incl_u32(ps_nt), 63);
        if ((threadIdx.x & 63) == 0) { ps_mn[threadIdx.x >> 6] = mn_inv; ps_mx[threadIdx.x >> 6] = mx; ps_sum[threadIdx.x >> 6] = wsum; }
        {   // the instance sums of the workgroup's sixteen 16-splat groups (for k_rcount's balanced chunks): a 16-lane row sum, four rows per wave
            __shared__ uint32_t ps_grp[16];
            uint32_t r = ps_nt;
            r += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)r, 0x111, 0xf, 0xf, false);   // row_shr:1
            r += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)r, 0x112, 0xf, 0xf, false);   // row_shr:2
            r += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)r, 0x114, 0xf, 0xf, false);   // row_shr:4
            r += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)r, 0x118, 0xf, 0xf, false);   // row_shr:8
            if ((threadIdx.x & 15) == 15) ps_grp[threadIdx.x >> 4] = r;
            __syncthreads();
            if (threadIdx.x < 4)
                a.pstat[(size_t)gridDim.x * (1 + threadIdx.x) + blockIdx.x] =
                    make_uint4(ps_grp[4 * threadIdx.x], ps_grp[4 * threadIdx.x + 1], ps_grp[4 * threadIdx.x + 2], ps_grp[4 * threadIdx.x + 3]);
        }
        if (threadIdx.x == 0)
            a.pstat[blockIdx.x] = make_uint4(~max(max(ps_mn[0], ps_mn[1]), max(ps_mn[2], ps_mn[3])), max(max(ps_mx[0], ps_mx[1]), max(ps_mx[2], ps_mx[3])),
                                             (ps_sum[0] + ps_sum[1]) + (ps_sum[2] + ps_sum[3]), 0u);
        put_rows(a.sspan, span2, std::integral_constant<int, 2>{});
    }
    if (a.brec) {
        // ---- binning record: operands of the exact reach test and the quadrant rect to run it on (= 2 x the snug TILE rect, so that
        // the streams hold exactly what the parity path's per-tile epilogue keeps), plus the frame statistics of the binned splats
        uint32_t nt = 0, r0 = 0, r1 = 0;
        if (visible) {
            Reach r = reach_of(px, py, con0, con1, con2, opac);
            nt = (uint32_t)((rmaxx - rminx) * (rmaxy - rminy));
            int minx = rminx, miny = rminy, maxx = rmaxx, maxy = rmaxy;
            snug_rect(r, minx, miny, maxx, maxy, nt);
            if (nt) {
                const Span sp = span_of(r);   // what the per-quadrant test of the binning walk needs (gsr_device.h: band_of)
                r0 = (uint32_t)(2 * minx) | ((uint32_t)(2 * miny) << 16);
                r1 = (uint32_t)(2 * maxx) | ((uint32_t)(2 * maxy) << 16);
                a.brec[3 * i + 0] = make_float4(sp.px, sp.py, sp.B, sp.det);
                a.brec[3 * i + 1] = make_float4(sp.twoTA, sp.A, sp.dyr, __int_as_float(sp.mode));
            }
        }
        if (i < a.P) a.brec[3 * i + 2] = make_float4(__uint_as_float(r0), __uint_as_float(r1), 0.f, 0.f);
        // statistics of the binned splats: wave partials through LDS, ONE set of atomics per workgroup, spread over GSR_STAT_SLOTS lines
        __shared__ uint32_t st_n[4], st_mx[4], st_mn[4];
        __shared__ unsigned long long st_t[4], st_r[4];
        const bool binned = nt != 0;
        const unsigned long long bal = __ballot(binned);
        const uint32_t dbits = binned ? __float_as_uint(depth) : 0u;
        const uint32_t mx = wave_max_u32(dbits);
        const uint32_t mn_inv = wave_max_u32(binned ? ~dbits : 0u);
        unsigned long long tsum = nt, rsum = visible ? (unsigned long long)((rmaxx - rminx) * (rmaxy - rminy)) : 0ull;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { tsum += __shfl_xor(tsum, d, 64); rsum += __shfl_xor(rsum, d, 64); }
        const int wv = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) { st_n[wv] = (uint32_t)__builtin_popcountll(bal); st_mx[wv] = mx; st_mn[wv] = mn_inv; st_t[wv] = tsum; st_r[wv] = rsum; }
        __syncthreads();
        if (threadIdx.x == 0) {
            BinStatSlot* sl = a.hdr->slot + (blockIdx.x & (GSR_STAT_SLOTS - 1));
            const uint32_t n4 = st_n[0] + st_n[1] + st_n[2] + st_n[3];
            const unsigned long long r4 = st_r[0] + st_r[1] + st_r[2] + st_r[3];
            if (n4) {
                atomicAdd(&sl->nvis, n4);
                atomicMax(&sl->dmax_bits, max(max(st_mx[0], st_mx[1]), max(st_mx[2], st_mx[3])));
                atomicMax(&sl->dmin_inv, max(max(st_mn[0], st_mn[1]), max(st_mn[2], st_mn[3])));
                atomicAdd(&sl->binned_tiles, st_t[0] + st_t[1] + st_t[2] + st_t[3]);
            }
            if (r4) atomicAdd(&sl->rect_total, r4);
        }
    }
}

// ------------------------------------------------------------------------------------------
// k_count: instances per tile.  Each workgroup owns a contiguous chunk of splats and histograms
// their tile rects in LDS (ds_add, no return); only the non-empty bins go to the global counters,
// so a hot tile sees one L2 atomic per workgroup instead of one per instance.
// ------------------------------------------------------------------------------------------
template <bool CULL>
__global__ __launch_bounds__(256) void k_count(int P, int gx, int tiles, const ushort4* __restrict__ rect,
                                                const uint32_t* __restrict__ tiles_touched, const float4* __restrict__ grec,
                                                uint32_t* __restrict__ tile_count, unsigned long long* __restrict__ rect_total,
                                                uint32_t* __restrict__ block_hist)
{
    extern __shared__ uint32_t hist[];
    __shared__ unsigned long long rect_sum[4];
    const int tid = threadIdx.x;
    const bool direct = tiles > GSR_LDS_HIST_TILES;   // tile grid too large for an LDS histogram: count in L2 (slow path)
    if (!direct) {
        for (int t = tid; t < tiles; t += 256) hist[t] = 0u;
        __syncthreads();
    }
    const int chunk = ((P + (int)gridDim.x - 1) / (int)gridDim.x + 255) / 256 * 256;
    const int begin = blockIdx.x * chunk;
    const int end = min(P, begin + chunk);
    unsigned long long touched = 0;
    for (int base = begin; base < end; base += 256) {
        const int i = base + tid;
        uint32_t n = 0;
        int minx = 0, miny = 0, maxx = 0, maxy = 0;
        if (i < end) {
            n = tiles_touched[i];
            touched += n;
            const ushort4 r = rect[i];
            minx = r.x; miny = r.y; maxx = r.z; maxy = r.w;
            if (CULL && n) {
                const float4 g0 = grec[3 * (size_t)i], g1 = grec[3 * (size_t)i + 1];
                snug_rect(reach_of(g0.x, g0.y, g0.z, g0.w, g1.x, g1.y), minx, miny, maxx, maxy, n);
            }
        }
        if (direct)
            for_each_tile(minx, miny, maxx, maxy, n, gx, [=](uint32_t tile, uint32_t, uint32_t) { atomicAdd(&tile_count[tile], 1u); }, 0u, 0u);
        else
            for_each_tile(minx, miny, maxx, maxy, n, gx, [](uint32_t tile, uint32_t, uint32_t) { atomicAdd(&hist[tile], 1u); }, 0u, 0u);
    }
    // the rect-based instance count (the reference's num_rendered) is kept beside the culled one: sum of tiles_touched
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) touched += __shfl_xor(touched, d, 64);
    if ((tid & 63) == 0) rect_sum[tid >> 6] = touched;
    __syncthreads();
    if (tid == 0) atomicAdd(rect_total, rect_sum[0] + rect_sum[1] + rect_sum[2] + rect_sum[3]);
    if (direct) return;
    uint32_t* __restrict__ mine = block_hist + (size_t)blockIdx.x * tiles;   // kept for k_scatter (same chunk, same histogram)
    for (int t = tid; t < tiles; t += 256) {
        const uint32_t v = hist[t];
        mine[t] = v;
        if (v) atomicAdd(&tile_count[t], v);
    }
}
template __global__ void k_count<false>(int, int, int, const ushort4*, const uint32_t*, const float4*, uint32_t*, unsigned long long*, uint32_t*);
template __global__ void k_count<true>(int, int, int, const ushort4*, const uint32_t*, const float4*, uint32_t*, unsigned long long*, uint32_t*);

// ------------------------------------------------------------------------------------------
// k_tile_scan: exclusive scan over the tile counters (single workgroup, 1024 threads)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_tile_scan(int tiles, const uint32_t* __restrict__ tile_count,
                                                     uint32_t* __restrict__ tile_start, uint32_t* __restrict__ tile_cursor,
                                                     uint2* __restrict__ ranges, uint32_t* __restrict__ tile_order,
                                                     unsigned long long* __restrict__ total_dev, unsigned long long* mailbox,
                                                     unsigned long long seq, unsigned long long post_capacity)
{
    __shared__ uint32_t wave_tot[16];
    __shared__ unsigned long long carry_s;
    __shared__ uint32_t bucket[34];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    unsigned long long grand = 0;
    for (int base = 0; base < tiles; base += 1024) {
        const int t = base + tid;
        const uint32_t v = (t < tiles) ? tile_count[t] : 0u;
        // wave inclusive scan
        uint32_t incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = __shfl_up(incl, d, 64);
            if (lane >= d) incl += up;
        }
        if (lane == 63) wave_tot[wid] = incl;
        __syncthreads();
        uint32_t wave_off = 0;
        for (int w = 0; w < wid; ++w) wave_off += wave_tot[w];
        const unsigned long long carry = carry_s;
        const uint32_t excl = (uint32_t)carry + wave_off + incl - v;   // offsets are 32-bit like upstream's
        if (t < tiles) {
            tile_start[t] = excl;
            tile_cursor[t] = 0u;
            ranges[t] = v ? make_uint2(excl, excl + v) : make_uint2(0u, 0u);
        }
        __syncthreads();
        if (tid == 1023) carry_s = carry + wave_off + incl;   // 64-bit running total: a frame past 2^32 instances is reported, not wrapped
        __syncthreads();
    }
    if (tid == 0) {
        grand = carry_s;
        *total_dev = grand;
        // post (seq, I) to the host: one 8-byte system-scope store into mapped pinned memory
        __hip_atomic_store(mailbox, (seq << 40) | (grand & 0xFFFFFFFFFFull), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        // deferred count (post_capacity = the binning capacity, else ~0): a frame that does not fit leaves a STICKY mark in the slot's second
        // word -- later frames overwrite the count above, nothing but the host clears this one (gsr_count_slot_overflow)
        if (grand > post_capacity) __hip_atomic_store(mailbox + GSR_COUNT_SLOTS, grand, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // ---- heavy-first launch order for the per-tile kernels --------------------------------------------
    // Per-tile work is heavy-tailed (a few tiles hold thousands of instances).  The dispatcher hands out
    // workgroups in index order, so listing tiles by descending instance count (bucketed by log2, one
    // counting-sort pass in LDS) lets the light tiles fill in behind the heavy ones instead of a heavy
    // tile starting late on an already busy CU.  Pure scheduling: results do not depend on the order.
    if (tid < 34) bucket[tid] = 0u;
    __syncthreads();
    auto bucket_of = [](uint32_t c) { return c ? 32u - (uint32_t)(31 - __builtin_clz(c)) - 1u : 32u; };   // big counts first, empty last
    for (int t = tid; t < tiles; t += 1024) atomicAdd(&bucket[bucket_of(tile_count[t])], 1u);
    __syncthreads();
    if (tid == 0) {
        uint32_t run = 0;
        for (int b = 0; b < 33; ++b) { const uint32_t c = bucket[b]; bucket[b] = run; run += c; }
    }
    __syncthreads();
    for (int t = tid; t < tiles; t += 1024) tile_order[atomicAdd(&bucket[bucket_of(tile_count[t])], 1u)] = (uint32_t)t;
}

// ------------------------------------------------------------------------------------------
// k_scatter: one (depth bits << 32 | splat) entry into every touched tile's segment.  Same chunking
// as k_count, whose per-workgroup histogram it re-uses: the workgroup reserves one contiguous sub-range per
// touched tile with a single returning L2 atomic, then hands out slots with returning LDS atomics.
// Order inside a tile segment is arbitrary here; k_tile_sort fixes it.
// ------------------------------------------------------------------------------------------
template <bool CULL>
__global__ __launch_bounds__(256) void k_scatter(int P, int gx, int tiles, const float* __restrict__ depths,
                                                  const ushort4* __restrict__ rect, const uint32_t* __restrict__ tiles_touched,
                                                  const float4* __restrict__ grec, const uint32_t* __restrict__ tile_start, uint32_t* __restrict__ tile_cursor,
                                                  unsigned long long* __restrict__ keys, unsigned long long capacity,
                                                  const unsigned long long* __restrict__ total_dev, const uint32_t* __restrict__ block_hist)
{
    extern __shared__ uint32_t hist[];
    if (*total_dev > capacity) return;  // the host will grow the buffer and replay the frame
    const int tid = threadIdx.x;
    const bool direct = tiles > GSR_LDS_HIST_TILES;
    const int chunk = ((P + (int)gridDim.x - 1) / (int)gridDim.x + 255) / 256 * 256;
    const int begin = blockIdx.x * chunk;
    const int end = min(P, begin + chunk);
    if (direct) {
        // slow path for tile grids beyond the LDS histogram: one returning L2 atomic per instance
        for (int base = begin; base < end; base += 256) {
            const int i = base + tid;
            uint32_t n = 0, dbits = 0;
            int minx = 0, miny = 0, maxx = 0, maxy = 0;
                if (i < end) {
                n = tiles_touched[i];
                const ushort4 r = rect[i];
                minx = r.x; miny = r.y; maxx = r.z; maxy = r.w;
                dbits = __float_as_uint(depths[i]);
                if (CULL && n) {
                    const float4 g0 = grec[3 * (size_t)i], g1 = grec[3 * (size_t)i + 1];
                    snug_rect(reach_of(g0.x, g0.y, g0.z, g0.w, g1.x, g1.y), minx, miny, maxx, maxy, n);
                }
            }
            for_each_tile(minx, miny, maxx, maxy, n, gx,
                          [=](uint32_t tile, uint32_t db, uint32_t idx) {
                              const uint32_t slot = tile_start[tile] + atomicAdd(&tile_cursor[tile], 1u);
                              keys[slot] = ((unsigned long long)db << 32) | (unsigned long long)idx;
                          },
                          dbits, (uint32_t)i);
        }
        return;
    }
    // this workgroup's tile histogram was already built by k_count (same chunk): reserve one contiguous sub-range per touched
    // tile with a single returning L2 atomic
    const uint32_t* __restrict__ mine = block_hist + (size_t)blockIdx.x * tiles;
    for (int t = tid; t < tiles; t += 256) {
        const uint32_t v = mine[t];
        hist[t] = v ? tile_start[t] + atomicAdd(&tile_cursor[t], v) : 0u;   // first slot of this workgroup in tile t
    }
    __syncthreads();
    for (int base = begin; base < end; base += 256) {
        const int i = base + tid;
        uint32_t n = 0, dbits = 0;
        int minx = 0, miny = 0, maxx = 0, maxy = 0;
        if (i < end) {
            n = tiles_touched[i];
            const ushort4 r = rect[i];
            minx = r.x; miny = r.y; maxx = r.z; maxy = r.w;
            if (CULL && n) {
                const float4 g0 = grec[3 * (size_t)i], g1 = grec[3 * (size_t)i + 1];
                snug_rect(reach_of(g0.x, g0.y, g0.z, g0.w, g1.x, g1.y), minx, miny, maxx, maxy, n);
            }
            dbits = __float_as_uint(depths[i]);
        }
        for_each_tile(minx, miny, maxx, maxy, n, gx,
                      [=](uint32_t tile, uint32_t db, uint32_t idx) {
                          const uint32_t slot = atomicAdd(&hist[tile], 1u);
                          keys[slot] = ((unsigned long long)db << 32) | (unsigned long long)idx;
                      },
                      dbits, (uint32_t)i);
    }
}
template __global__ void k_scatter<false>(int, int, int, const float*, const ushort4*, const uint32_t*, const float4*, const uint32_t*, uint32_t*,
                                          unsigned long long*, unsigned long long, const unsigned long long*, const uint32_t*);
template __global__ void k_scatter<true>(int, int, int, const float*, const ushort4*, const uint32_t*, const float4*, const uint32_t*, uint32_t*,
                                         unsigned long long*, unsigned long long, const unsigned long long*, const uint32_t*);

// Epilogue for the register-sorted classes.  The sorted keys are first laid out in LDS in natural order so
// that thread t handles entries c*THREADS + t (c = 0..7): consecutive lanes own consecutive entries and
// their compacted records land next to each other (coalesced stores).  Membership is one 4-bit mask per
// entry; the stable compaction into the four quadrant streams uses ballots for the in-wave rank and ONE
// exclusive scan over the (chunk, wave) counters -- three barriers in total, gathers issued four chunks at a time.
template <int THREADS, int EPT>
__device__ __forceinline__ void epilogue_striped(const u64 (&key)[EPT], uint32_t n, uint32_t tile, uint32_t start, float ox, float oy,
                                                 u64* __restrict__ seg, uint32_t* __restrict__ point_list, uint32_t* __restrict__ qlbase,
                                                 uint32_t* __restrict__ qpbase, uint32_t* __restrict__ qcount, const float4* __restrict__ grec,
                                                 u64* __restrict__ sk, uint32_t (*__restrict__ cntw)[EPT * (THREADS / 64) + 1], int tid)
{
    constexpr int NW = THREADS / 64, NE = EPT * NW;   // waves, (chunk, wave) counters per quadrant
    constexpr int PER = (NE + 63) / 64;               // counters per lane in the scan
    const int lane = tid & 63, wid = tid >> 6;
#pragma unroll
    for (int k = 0; k < EPT; ++k) sk[tid * EPT + k] = key[k];
    __syncthreads();
    uint32_t msk[EPT], rank[EPT];   // rank: 4 x 8-bit in-wave exclusive ranks (0..63)
    const u64 tile_hi = (u64)tile << 32;
    // pass 1: gather the geometry half of each instance's per-splat record (centre, conic, opacity: 32 of its 48 bytes),
    // evaluate its quadrant mask / in-wave ranks, and write the reference-format lists in the parity modes
#pragma unroll
    for (int h = 0; h < EPT / 4; ++h) {
        float4 g0[4], g1[4];
        u64 kk[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t i = (uint32_t)(4 * h + k) * THREADS + (uint32_t)tid;
            kk[k] = i < n ? sk[i] : 0ull;
            const uint32_t idx = (uint32_t)kk[k];
            g0[k] = grec[3 * (size_t)idx + 0];
            g1[k] = grec[3 * (size_t)idx + 1];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = 4 * h + k;
            const uint32_t i = (uint32_t)c * THREADS + (uint32_t)tid;
            if (i < n && point_list) {   // the reference-format lists are a parity/debug artefact: nothing downstream reads them
                seg[i] = tile_hi | (kk[k] >> 32);   // reference-format key: tile id | depth bits
                point_list[start + i] = (uint32_t)kk[k];
            }
            const uint32_t m = i < n ? quadrant_mask(make_float2(g0[k].x, g0[k].y), make_float4(g0[k].z, g0[k].w, g1[k].x, g1[k].y), ox, oy) : 0u;
            msk[c] = m;
            uint32_t r = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const unsigned long long bal = __ballot((m >> q) & 1u);
                r |= (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u)) << (8 * q);
                if (lane == 0) cntw[q][c * NW + wid] = (uint32_t)__builtin_popcountll(bal);
            }
            rank[c] = r;
        }
    }
    __syncthreads();
    // exclusive scan of the NE counters of each quadrant (wave q scans quadrant q; NE <= 128 = 2 per lane)
    if (wid < 4) {
        uint32_t a[PER], sum = 0;
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int e = PER * lane + j;
            a[j] = e < NE ? cntw[wid][e] : 0u;
            sum += a[j];
        }
        uint32_t incl = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = (uint32_t)__builtin_amdgcn_ds_bpermute((lane >= d ? lane - d : lane) << 2, (int)incl);
            if (lane >= d) incl += up;
        }
        uint32_t run = incl - sum;
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int e = PER * lane + j;
            if (e < NE) cntw[wid][e] = run;
            run += a[j];
        }
        if (lane == 63) cntw[wid][NE] = incl;   // total
    }
    __syncthreads();
    // pass 2: the quadrant streams get the SPLAT INDICES (4 bytes per (instance, quadrant) pair), stable order; in the parity
    // modes a twin stream keeps each entry's position in the tile list (what the reference's n_contrib counts)
#pragma unroll
    for (int c = 0; c < EPT; ++c) {
        const uint32_t i = (uint32_t)c * THREADS + (uint32_t)tid;
        const uint32_t m = msk[c];
        const uint32_t idx = m ? (uint32_t)sk[i] : 0u;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if ((m >> q) & 1u) {
                const uint32_t pos = cntw[q][c * NW + wid] + ((rank[c] >> (8 * q)) & 0xFFu);
                qpbase[(size_t)q * n + pos] = idx;
                if (qlbase) qlbase[(size_t)q * n + pos] = i;
            }
        }
    }
    if (tid < 4) qcount[4 * tile + tid] = cntw[tid][NE];
}

// Three size classes share this body: tiles with n_lo < n <= n_hi are handled, the rest exit at once.
//   small: <= 2048 entries, 256 threads x 8 keys, 16 KiB LDS (many workgroups per CU)
//   large: <= 8192 entries, 1024 threads x 8 keys, 64 KiB LDS
//   xl:    <= 16384 entries, 1024 threads x 16 keys, 128 KiB LDS; beyond that chunked sorts + merges in global memory
template <int KEYS, int THREADS>
__global__ __launch_bounds__(THREADS) void k_tile_sort(uint32_t n_lo, uint32_t n_hi, int gx, const uint32_t* __restrict__ tile_order,
                                                        const uint32_t* __restrict__ tile_count,
                                                    const uint32_t* __restrict__ tile_start, unsigned long long* __restrict__ keys,
                                                    uint32_t* __restrict__ point_list, uint32_t* __restrict__ qlist,
                                                    uint32_t* __restrict__ qpos, uint32_t* __restrict__ qcount, uint32_t* __restrict__ qstart,
                                                    const float4* __restrict__ grec,
                                                    unsigned long long capacity, const unsigned long long* __restrict__ total_dev)
{
    __shared__ unsigned long long skeys[KEYS];
    __shared__ uint32_t wave_cnt[4][THREADS / 64];   // [quadrant][wave]  (chunked fallback epilogue)
    constexpr int EPT = KEYS / THREADS;
    __shared__ uint32_t cntw[4][EPT * (THREADS / 64) + 1];   // [quadrant][(chunk, wave)] + total  (striped epilogue)
    if (*total_dev > capacity) return;
    const uint32_t tile = tile_order[blockIdx.x];
    const uint32_t n = tile_count[tile];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (n <= n_lo || n > n_hi) {
        if (n == 0 && n_lo == 0 && tid < 4) { qcount[4 * tile + tid] = 0u; qstart[4 * tile + tid] = 0u; }
        return;
    }
    const uint32_t start = tile_start[tile];
    if (tid < 4) qstart[4 * tile + tid] = 4u * start + (uint32_t)tid * n;   // the tile's four n-slot streams
    unsigned long long* seg = keys + start;
    const bool in_lds = n <= (uint32_t)KEYS;
    const float ox = (float)((tile % (uint32_t)gx) * GSR_BLOCK_X), oy = (float)((tile / (uint32_t)gx) * GSR_BLOCK_Y);
    uint32_t* const qpbase = qpos + (size_t)4 * start;      // the tile's four quadrant streams of splat indices, n slots each
    uint32_t* const qlbase = qlist ? qlist + (size_t)4 * start : nullptr;   // parity modes: the entries' positions in the tile list
    if (in_lds) {
        static_assert(EPT == 8 || EPT == 16, "register sort holds 8 or 16 keys per thread");
        u64 key[EPT];
        block_sort_regs<THREADS, EPT>(key, skeys, seg, n, tid);
        epilogue_striped<THREADS, EPT>(key, n, tile, start, ox, oy, seg, point_list, qlbase, qpbase, qcount, grec, skeys, cntw, tid);
        return;
    } else {
        // more entries than this class holds in LDS: chunked register sorts + global merge passes (scratch: the tile's
        // still-unwritten quadrant streams, 16 n bytes for n 8-byte keys)
        oversize_sort<THREADS, EPT>(seg, reinterpret_cast<u64*>(qpbase), skeys, n, tid);
        __syncthreads();
    }
    // ---- epilogue: reference-format keys / point list, and the four 8x8-quadrant record streams --------
    // A record goes to quadrant q only if the exact ellipse {alpha >= 1/255} of the splat can reach a pixel
    // centre of q (conservatively padded), i.e. only (record, quadrant) pairs the blend would skip for every
    // pixel are dropped: the composited result is unchanged, the blend's lists get ~3x shorter.  Order inside
    // a quadrant stream is the tile order (stable compaction), r2.z keeps the position in the tile list.
    const unsigned long long tile_hi = (unsigned long long)tile << 32;
    uint32_t running[4] = {0u, 0u, 0u, 0u};
    for (uint32_t base = 0; base < n; base += THREADS) {
        const uint32_t i = base + tid;
        const bool valid = i < n;
        bool f[4] = {false, false, false, false};
        uint32_t idx = 0u;
        if (valid) {
            const unsigned long long k = in_lds ? skeys[i] : seg[i];
            idx = (uint32_t)k;
            if (point_list) {
                seg[i] = tile_hi | (k >> 32);
                point_list[start + i] = idx;
            }
            const float4 r0 = grec[3 * (size_t)idx + 0];
            const float4 r1 = grec[3 * (size_t)idx + 1];
            const uint32_t m = quadrant_mask(make_float2(r0.x, r0.y), make_float4(r0.z, r0.w, r1.x, r1.y), ox, oy);
#pragma unroll
            for (int q = 0; q < 4; ++q) f[q] = (m >> q) & 1u;
        }
        uint32_t prefix[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned long long bal = __ballot(f[q]);
            prefix[q] = (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
            if (lane == 0) wave_cnt[q][wid] = (uint32_t)__builtin_popcountll(bal);
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint32_t off = running[q], tot = 0;
#pragma unroll
            for (int w = 0; w < THREADS / 64; ++w) {
                const uint32_t cw = wave_cnt[q][w];
                if (w < wid) off += cw;
                tot += cw;
            }
            if (f[q]) {
                qpbase[(size_t)q * n + off + prefix[q]] = idx;
                if (qlbase) qlbase[(size_t)q * n + off + prefix[q]] = i;
            }
            running[q] += tot;
        }
        __syncthreads();
    }
    if (tid < 4) qcount[4 * tile + tid] = running[tid];
}

template __global__ void k_tile_sort<GSR_SORT_SMALL_KEYS, 256>(uint32_t, uint32_t, int, const uint32_t*, const uint32_t*, const uint32_t*, unsigned long long*, uint32_t*, uint32_t*,
                                                                uint32_t*, uint32_t*, uint32_t*, const float4*, unsigned long long,
                                                                const unsigned long long*);
template __global__ void k_tile_sort<GSR_SORT_LDS_KEYS, 1024>(uint32_t, uint32_t, int, const uint32_t*, const uint32_t*, const uint32_t*, unsigned long long*, uint32_t*, uint32_t*,
                                                               uint32_t*, uint32_t*, uint32_t*, const float4*, unsigned long long,
                                                               const unsigned long long*);
template __global__ void k_tile_sort<GSR_SORT_XL_KEYS, 1024>(uint32_t, uint32_t, int, const uint32_t*, const uint32_t*, const uint32_t*, unsigned long long*, uint32_t*, uint32_t*,
                                                              uint32_t*, uint32_t*, uint32_t*, const float4*, unsigned long long,
                                                              const unsigned long long*);

// ------------------------------------------------------------------------------------------
// k_render: front-to-back compositing.  Workgroup = one 16x16 tile = 4 independent waves, wave w
// owns the 8x8 quadrant (w&1, w>>1) and walks that quadrant's record stream.  The record index is
// wave-uniform, so the loads below are scalar-unit loads: one 48-byte fetch serves all 64 pixels
// and the values sit in SGPRs -- no LDS staging, no barriers, each wave stops on its own.
// ------------------------------------------------------------------------------------------
#ifdef GSR_EXPERIMENT_TIMELINE   // `make timeline`: per-wave stamps of k_render for tools/bwd_timeline.py
__device__ unsigned long long gsr_dbg_fwd[4 * 16384];
extern "C" int gsr_debug_read_fwd(unsigned long long* host, int n) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(gsr_dbg_fwd), (size_t)n * 8); }
#endif
// CONT (fast blend only): the CONTINUATION kernel, launched right behind k_render<true, false> (gsr_api.hip).  The forward blend ends when its
// deepest quadrant's serial walk ends (round 5's timeline: the second half of the kernel is a tail of a few hundred deep walks, each alone on
// its SIMD at the single-wave issue limit), and which quadrants walk deep is only known once they have: the stream length says little (half of
// cfg3's streams hold more than 600 entries, ten are WALKED beyond 400).  So a walk that reaches entry s.cont_chunks * GSR_BWD_SEGMENT with
// pixels still open parks its state -- (T, C, last contributor) of its 64 pixels, 1280 bytes -- and leaves; this kernel's workgroups pull those
// quadrants and walk the rest of each stream FOUR CHUNKS AT A TIME: wave w takes chunks c0 + w, c0 + w + 4, ... of 60 entries; every chunk but
// the first is walked from T = 1 (its own transmittance product and colour sum), and the waves hand the true state down the chunks in order
// through LDS:  T = T_prefix * T_chunk,  C = C_prefix + T_prefix * C_chunk  for a pixel that stays open across the chunk (transmittance only
// falls, so "T_prefix * T_chunk >= 1e-4" says that no record of the chunk closed it); a pixel that closes INSIDE the chunk is evaluated again,
// from the true prefix, with the record-parallel pass of the tail mode (one wave scan over the chunk's 60 records) -- every pixel closes once.
// Which chunks are walked from T = 1 and which from the true state depends on the chunk index and on the open pixels the wave itself saw four
// chunks earlier, never on timing: the result is the same bits run to run.  Same arithmetic per record as the lone walk; what differs is the
// association of the products (1e-7 relative), inside the fast blend's tolerance.
#ifndef GSR_CONT_WAVES
#define GSR_CONT_WAVES 8   // waves of a workgroup of the continuation KERNEL (CONT == 2) = chunks of a quadrant in flight
#endif
#ifndef GSR_EXP_NO_SOLO
#define GSR_SOLO_THREADS 64
#else
#define GSR_SOLO_THREADS 256
#endif
template <bool FAST, int CONT, bool INFER>   // INFER: the instance for frames no backward can follow (GsrSettings.forward_only): no last-contributor bookkeeping in the walk
#ifndef GSR_EXP_LB
#define GSR_EXP_LB (CONT == 1 ? 5 : 1)   // (only the instance that carries the continuation workgroups needs its register budget capped: their body would take the tiles' walks from five waves per SIMD to four)
#endif
__global__ __launch_bounds__(CONT == 2 ? 64 * GSR_CONT_WAVES : (CONT == 0 ? GSR_SOLO_THREADS : 256), GSR_EXP_LB) void k_render(Settings s, const uint32_t* __restrict__ tile_order, const uint32_t* __restrict__ qstart,
                                                 const uint32_t* __restrict__ qcount,
                                                 const float4* __restrict__ grec, const uint32_t* __restrict__ qpos,
                                                 const uint32_t* __restrict__ qlist, float* __restrict__ final_T,
                                                 uint32_t* __restrict__ n_contrib, uint32_t* __restrict__ n_contrib_q,
                                                 float* __restrict__ c_final, float4* __restrict__ ck,
                                                 float* __restrict__ out_color, unsigned long long capacity,
                                                 const unsigned long long* __restrict__ total_dev, uint32_t* __restrict__ units, int tiles)
{
    static_assert(FAST || CONT == 0, "continuations belong to the fast blend");
    static_assert(!INFER || CONT == 0, "the inference instances carry no continuation code");
    const bool fwd_only = INFER || s.forward_only;
    if (*total_dev > capacity) return;
#ifdef GSR_EXPERIMENT_TIMELINE
    const unsigned long long t_start = wall_clock64();
#endif
    const int W = s.W, H = s.H;
    const int gx = (W + GSR_BLOCK_X - 1) / GSR_BLOCK_X;
    // the continuation area behind the backward's unit lists (gsr.h: GsrImageLayout.units): header (word 0: quadrants parked so far, word 32: the
    // continuation workgroups' pull cursor, words 64 + 32 k, k < 16: tile waves that have reported; all zeroed by k_preprocess), the list of parked
    // quadrants (launch position << 2 | quadrant; ~0 = not yet written: k_preprocess fills it), and 5 x 64 floats of state per quadrant slot
    const uint32_t ucap = (uint32_t)unit_list_cap((size_t)tiles);
    uint32_t* const cont_hdr = units + cont_hdr_word((size_t)tiles);
    uint32_t* const cont_list = cont_hdr + GSR_CONT_HDR_WORDS;
    float* const cont_state = reinterpret_cast<float*>(cont_list + 4 * (size_t)tiles);
    const int cont_c = (FAST && s.cont_chunks > 0) ? s.cont_chunks : 0x7fffffff;   // the chunk a lone walk hands over at
    // SOLO (round 6, the instances without continuation code): ONE wave per workgroup, workgroup 4 p + q = quadrant q of the tile at launch position p.
    // A tile's four quadrant walks share nothing; as four waves of one workgroup the slots of the three that finish first stay taken until the
    // deepest is done.
#ifndef GSR_EXP_NO_SOLO
    constexpr bool SOLO = CONT == 0;
#else
    constexpr bool SOLO = false;
#endif
    const int pwave = SOLO ? 0 : __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));       // this wave inside its workgroup
    const int lane = threadIdx.x & 63;
    // per QUADRANT (k_render: once; the continuation kernel: once per pulled quadrant)
    const bool helper = CONT == 2 || (CONT == 1 && blockIdx.x >= (uint32_t)tiles);   // a continuation workgroup (below): no tile of its own
    uint32_t lpos = SOLO ? blockIdx.x >> 2 : blockIdx.x;         // the tile's position in the launch order
    int tile = helper ? 0 : (int)tile_order[lpos];
    int tile_x = tile % gx, tile_y = tile / gx;
    int wave = SOLO ? (int)(blockIdx.x & 3u) : pwave;            // the QUADRANT this wave blends (CONT: all four waves the same one)
    int pxi = tile_x * GSR_BLOCK_X + (wave & 1) * 8 + (lane & 7);
    int pyi = tile_y * GSR_BLOCK_Y + (wave >> 1) * 8 + (lane >> 3);
    bool inside = pxi < W && pyi < H;
    float pixx = (float)pxi, pixy = (float)pyi;

    int n = helper ? 0 : (int)qcount[4 * tile + wave];
    uint32_t qs = helper ? 0u : qstart[4 * tile + wave];
    const float4* __restrict__ rec = grec;                      // the per-splat records (48 bytes each)
    const uint32_t* __restrict__ qp = qpos + qs;                // this quadrant's stream of splat indices

    // ONE transmittance register per pixel (round 4; rounds 1 - 3 carried the final T beside a working copy): Tw is the transmittance while
    // the pixel is open and MINUS the transmittance it ended with once it is closed (saturated; outside the image: 0).  A closed pixel needs no
    // flag: Tw * (1 - alpha) <= 0 fails the ">= 0.0001" acceptance test by itself, c * 0 * Tw adds a zero, and the closing select writes -|Tw|
    // (source modifiers of v_cndmask: no instruction) -- one select per record less than updating T and Tw.  |Tw| is the pixel's T throughout.
    float Tw = inside ? 1.0f : 0.0f;
    float C0 = 0.f, C1 = 0.f, C2 = 0.f;
    uint32_t last_q = 0;

    // the kernel ends when its longest stream ends: let those waves win issue arbitration on their SIMD
    if (n > 2048) __builtin_amdgcn_s_setprio(3);
    else if (n > 1024) __builtin_amdgcn_s_setprio(2);
    else if (n > 512) __builtin_amdgcn_s_setprio(1);

    // The stream is walked RB records at a time: the RB exponentials are independent (instruction-level
    // parallelism for a wave that is alone on its SIMD), only the short T/C update is sequential.
    // Skips are predicated (selects), so the arithmetic per contributing record is exactly A.3's.
    // A wave on the critical path retires one instruction per ~6.6 cycles whatever its kind (measured: tools/pmc_kernel.sh,
    // tools/bwd_timeline.py), so the loop is written for INSTRUCTION COUNT, scalar ones included: no per-record bounds
    // tests (full batches take an unmasked body), 32-bit record offsets, no exec-mask branches around the exponential.
#ifndef GSR_EXP_RB
#define GSR_EXP_RB 3
#endif
    constexpr int RB = GSR_EXP_RB;
#ifndef GSR_FAST_TAIL_LANES
#define GSR_FAST_TAIL_LANES 12
#endif
    // switch to record-parallel mode when this few pixels are still open.  Exact walk: the tail's T/C update is serial over the hits (swept
    // 0..16 on the instruction-count build: 2-4 best, 8 costs 10 %, 16 costs 50 % of the kernel).  Fast blend: the tail evaluates a pixel's
    // 60 records with one wave scan (no serial loop), ~75 instructions per (pixel, chunk) against 60 x 42 for the walk, so it takes over
    // much earlier (model on the cfg3 frame, tools/remap_model.py: 31.7 M -> 26.3 M instructions at 8..16 open pixels).
    constexpr int TAIL_LANES = FAST ? GSR_FAST_TAIL_LANES : 4;
    struct Rec4 { float a[RB][8]; float cbl[RB]; float hi[RB]; };   // a = (x, y, conic a, conic b, conic c, opacity | fast: log2 opacity, red, green); blue; fast: alpha's upper bound -- in vector registers, read from the wave's LDS stage
    // last_q is carried RELATIVE to the walk position (lq = last_q - j0 at the top of an iteration): a hit then stores a small constant,
    // which v_cndmask takes inline -- an absolute index costs a scalar add and a v_mov per record on top of the select
    int lq = 0;
    auto blend4 = [&](int jb, const Rec4& R, auto masked, auto off) {
        constexpr int OFF = decltype(off)::value;   // jb - (the iteration's j0)
        float alpha[RB];
        bool ok[RB];
#pragma unroll
        for (int u = 0; u < RB; ++u) {
            const float dx = R.a[u][0] - pixx;
            const float dy = R.a[u][1] - pixy;
            float power, a;
            if constexpr (FAST) {
                // the record holds the conic scaled by -log2(e)/2 and L = log2(opacity): power = log2(opacity * G), five instructions, and
                // alpha's raw value is the exponential itself.  Accepted iff 1/255 <= raw <= its upper bound (= opacity: `power <= 0`), as
                // ONE median + ONE comparison (a NaN -- an opacity below 1/255 has L = NaN -- fails it)
                power = __builtin_fmaf(__builtin_fmaf(R.a[u][3], dy, R.a[u][2] * dx), dx, __builtin_fmaf(R.a[u][4] * dy, dy, R.a[u][5]));
                const float raw = __builtin_amdgcn_exp2f(power);
#ifdef GSR_EXP_HI_TEST
                ok[u] = __builtin_amdgcn_fmed3f(raw, 1.0f / 255.0f, R.hi[u]) == raw;
#else
                ok[u] = raw >= 1.0f / 255.0f;   // (NaN fails)
#endif
                a = __builtin_amdgcn_fmed3f(raw, 0.0f, 0.99f);   // = min(0.99, raw) for raw >= 0.  (fminf would first canonicalise the exponential's result, one more
                                                                 // instruction; an inline-asm v_min hides the operand from the compiler's hazard recogniser, which has to
                                                                 // put a wait state between a transcendental and the VALU instruction that reads its result: measured, T > 1)
            } else {
                power = -0.5f * (R.a[u][2] * dx * dx + R.a[u][4] * dy * dy) - R.a[u][3] * dx * dy;
                a = sel_min(0.99f, R.a[u][5] * gsr_expf_blend(power));
                asm volatile("" : "+v"(a));   // evaluated for every lane: power > 0 is too rare to pay an exec-mask branch per record
                ok[u] = power <= 0.0f && a >= 1.0f / 255.0f;
            }
            if (decltype(masked)::value) ok[u] = ok[u] && (jb + u) < n;
            alpha[u] = ok[u] ? a : 0.0f;
        }
        if constexpr (FAST) {
            // (the colour sums of record u are issued between record u + 1's acceptance compare and the selects that read its mask: a VALU
            //  instruction that reads an SGPR pair as a mask needs two wait states behind the compare that wrote it -- left in source order the
            //  compiler fills them with s_nop, one issue slot per record)
            float wprev = 0.0f;
#pragma unroll
            for (int u = 0; u < RB; ++u) {
                const float test_T = __builtin_fmaf(-alpha[u], Tw, Tw);   // alpha == 0 (skipped record): exactly Tw, accepted, nothing changes
                const bool keep = test_T >= 0.0001f;                       // false at the record that saturates the pixel, and forever after
                if (u > 0) {
                    C0 = __builtin_fmaf(R.a[u - 1][6], wprev, C0);
                    C1 = __builtin_fmaf(R.a[u - 1][7], wprev, C1);
                    C2 = __builtin_fmaf(R.cbl[u - 1], wprev, C2);
                }
                const float ae = keep ? alpha[u] : 0.0f;                   // adding (c * 0) * T == +0 leaves C bit-identical: no selects on C
                wprev = ae * Tw;
                Tw = keep ? test_T : -__builtin_fabsf(Tw);
                if constexpr (!INFER) lq = ((int)keep & (int)ok[u]) ? OFF + u + 1 : lq;   // (&, not &&: with the short-circuit form the compiler kept a branch per record once the closing select carried source modifiers)
            }
            C0 = __builtin_fmaf(R.a[RB - 1][6], wprev, C0);
            C1 = __builtin_fmaf(R.a[RB - 1][7], wprev, C1);
            C2 = __builtin_fmaf(R.cbl[RB - 1], wprev, C2);
        } else {
#pragma unroll
            for (int u = 0; u < RB; ++u) {
                const float test_T = Tw * (1.0f - alpha[u]);
                const bool keep = test_T >= 0.0001f;
                const float ae = keep ? alpha[u] : 0.0f;
                C0 = C0 + R.a[u][6] * ae * Tw;
                C1 = C1 + R.a[u][7] * ae * Tw;
                C2 = C2 + R.cbl[u] * ae * Tw;
                Tw = keep ? test_T : -__builtin_fabsf(Tw);
                if constexpr (!INFER) lq = ((int)keep & (int)ok[u]) ? OFF + u + 1 : lq;
            }
        }
    };
    using Off0 = std::integral_constant<int, 0>;
    using Off1 = std::integral_constant<int, RB>;
    int j0 = 0;
    // ---- the walk, records staged through LDS (round 4) ------------------------------------------------------------------------
    // Through the scalar unit (above, kept for A/B builds) a batch's records can only be requested ONE batch ahead -- scalar loads
    // return out of order, so the only wait is lgkmcnt(0), which also waits for whatever was issued last -- and a request takes ~300 ns
    // (K-cache miss -> L2): the deepest quadrant's walk, which IS the kernel's duration, ran at 107 ns per record whether a record cost 40
    // instructions or 32 (measured: RB = 2 -> 85 us, RB = 3 -> 72 us).  Here the wave gathers a whole CHUNK of 60 records with
    // vector loads (lane l: entry 60 c + l, three 16-byte loads), one chunk ahead of the walk, parks them in its own 2 x 2880 bytes of
    // LDS and walks them with broadcast ds_read_b128: LDS returns in order (the compiler counts lgkmcnt exactly, a batch ahead costs
    // nothing), takes ~50 ns instead of ~300, and the record's fields arrive in VECTOR registers -- a VALU instruction with a scalar
    // operand issues at half rate on gfx950 (tools/valu_peak.hip), so the per-record arithmetic gets cheaper for the crowded SIMDs too.
    // Chunks are the backward's segments: the checkpoint test sits at the chunk boundary.
    constexpr int CH = GSR_BWD_SEGMENT;
    static_assert(CH <= GSR_WAVE && CH % (2 * RB) == 0, "a chunk is one gather of the wave and a whole number of double batches");
    constexpr int NWV = CONT == 2 ? GSR_CONT_WAVES : (SOLO ? 1 : 4);   // waves per workgroup
    // DIRECT (round 6, the instances without continuation code): the gather goes straight into LDS (global_load_lds_dwordx4: lane l's 16 bytes land at
    // base + 16 l, so a chunk is staged as three planes of 64 float4 -- part k of record j at [64 k + j]) instead of through twelve vector
    // registers that stay live across the whole walk of the chunk before; nothing is parked.
#ifndef GSR_EXP_NO_DIRECT
    constexpr bool DIRECT = CONT == 0;
#else
    constexpr bool DIRECT = false;
#endif
    constexpr int STAGE_F4 = DIRECT ? 3 * GSR_WAVE : CH * 3 + 3;   // (+3: the walk's read-ahead of a chunk's last double batch ends one batch past the chunk -- harmless, never used, but it has to be inside the allocation; DIRECT: the planes hold 64 entries)
    constexpr int RS = DIRECT ? 16 : 48;                           // bytes from a record to the next in the stage
    constexpr int P1 = DIRECT ? 16 * GSR_WAVE : 16, P2 = DIRECT ? 32 * GSR_WAVE : 32;   // ... and to a record's second and third part
    constexpr int BB = RB * RS;                                    // bytes per batch
    __shared__ float4 stage_all[NWV][2][STAGE_F4];
    float4(*const stage)[STAGE_F4] = stage_all[pwave];
    float4 g0, g1, g2;   // (not DIRECT) the chunk in flight: this lane's record
    uint32_t nidx = 0;   // DIRECT: this lane's stream entry of the chunk to be gathered NEXT, fetched a chunk ahead (the record loads depend on it: fetched
                         // where it is needed, every chunk boundary waits out one more round trip to memory)
    if constexpr (DIRECT) nidx = n > 0 ? qp[min(lane, n - 1)] : 0u;
    auto gather = [&](int c) {   // entries past the end re-read the last one (never walked); lanes CH.. load too (never parked)
        const size_t idx = DIRECT ? (size_t)nidx : (size_t)qp[min(c * CH + lane, n - 1)];
        if constexpr (DIRECT) {
            typedef const __attribute__((address_space(1))) void* gptr;
            typedef __attribute__((address_space(3))) void* lptr;
            const float4* p = rec + 3 * idx;
            nidx = qp[min((c + 1) * CH + lane, n - 1)];
            __builtin_amdgcn_global_load_lds((gptr)(p + 0), (lptr)&stage[c & 1][0], 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr)(p + 1), (lptr)&stage[c & 1][GSR_WAVE], 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr)(p + 2), (lptr)&stage[c & 1][2 * GSR_WAVE], 16, 0, 0);
        } else {
            g0 = rec[3 * idx + 0];
            g1 = rec[3 * idx + 1];
            g2 = rec[3 * idx + 2];
        }
    };
    auto landed = [&]() {   // DIRECT: everything gathered so far is in LDS
        if constexpr (DIRECT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    auto park = [&](int c) {
        if constexpr (!DIRECT) {
            if (lane < CH) {
                float4* d = &stage[c & 1][3 * lane];
                d[0] = g0; d[1] = g1; d[2] = g2;
            }
        }
    };
    // A batch's records come out of LDS with ds_read_b128 / ds_read_b64 at a wave-uniform address (broadcast), issued from inline
    // assembly: the reads have to be ISSUED a batch ahead of their use, behind the blend that frees their registers, and left to the
    // compiler they all end up at the top of the loop body (the fetched batch copied aside, the LDS waited for with nothing to do;
    // volatile loads turn into flat loads).  LDS returns in order, so `ready` waits with the exact count of reads issued since.
    // Rules that keep this sound: every `issue` is followed by a `ready` on the same registers before they die (the compiler
    // believes they were written at the issue), and nothing else of this wave is in flight on lgkmcnt inside the loop (no scalar
    // loads, no compiler-made LDS access: the parks sit between chunks, behind a full wait).
    typedef float v4f __attribute__((ext_vector_type(4)));
#ifdef GSR_EXP_HI_TEST
    typedef float v2f __attribute__((ext_vector_type(2)));
    typedef v2f q2_t;   // (blue, alpha's upper bound)
#define GSR_Q2_READ "ds_read_b64"
#else
    typedef float q2_t;   // blue
#define GSR_Q2_READ "ds_read_b32"
#endif
    struct RecV { v4f q0[RB], q1[RB]; q2_t q2[RB]; };
    static_assert(RB == 3, "the issue / ready assembly below is written for three records per batch");
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float4*)&stage[0][0], lds1 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float4*)&stage[1][0];
    // `addr`: the LDS byte address of the walk's current double batch (a vector register that only ever gets a constant added);
    // `O`: 0, 1 or 2 batches (BB bytes each) ahead of it, as immediates -- no address arithmetic per batch
    // (a macro: captured variables inside the operand list of an asm in a GENERIC lambda do not compile with this clang)
#define GSR_ISSUE(ADDR, O, V)                                                                                                                        \
    asm volatile("ds_read_b128 %0, %14 offset:%15\n\tds_read_b128 %1, %14 offset:%16\n\t" GSR_Q2_READ " %2, %14 offset:%17\n\t"                       \
                 "ds_read_b128 %3, %14 offset:%18\n\tds_read_b128 %4, %14 offset:%19\n\t" GSR_Q2_READ " %5, %14 offset:%20\n\t"                       \
                 "ds_read_b128 %6, %14 offset:%21\n\tds_read_b128 %7, %14 offset:%22\n\t" GSR_Q2_READ " %8, %14 offset:%23"                             \
                 : "=&v"(V.q0[0]), "=&v"(V.q1[0]), "=&v"(V.q2[0]), "=&v"(V.q0[1]), "=&v"(V.q1[1]), "=&v"(V.q2[1]), "=&v"(V.q0[2]),              \
                   "=&v"(V.q1[2]), "=&v"(V.q2[2]), /* the blend state as pass-through operands: the reads are issued BEHIND everything the */    \
                   /* previous blend computes (its instructions are free to move otherwise, and below this statement they need the old */       \
                   /* batch copied aside) */                                                                                                    \
                   "+v"(Tw), "+v"(C0), "+v"(C1), "+v"(C2), "+v"(lq)                                                                             \
                 : "v"(ADDR), "n"((O)), "n"((O) + P1), "n"((O) + P2), "n"((O) + RS), "n"((O) + RS + P1), "n"((O) + RS + P2), "n"((O) + 2 * RS),  \
                   "n"((O) + 2 * RS + P1), "n"((O) + 2 * RS + P2))
    // the batch is in its registers once at most `behind` reads issued after it are still in flight (9 = one batch)
    auto ready = [&](RecV& V, auto behind, Rec4& R) {
        if constexpr (decltype(behind)::value == 9)
            asm volatile("s_waitcnt lgkmcnt(9)" : "+v"(V.q0[0]), "+v"(V.q1[0]), "+v"(V.q2[0]), "+v"(V.q0[1]), "+v"(V.q1[1]), "+v"(V.q2[1]),
                         "+v"(V.q0[2]), "+v"(V.q1[2]), "+v"(V.q2[2]));
        else
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(V.q0[0]), "+v"(V.q1[0]), "+v"(V.q2[0]), "+v"(V.q0[1]), "+v"(V.q1[1]), "+v"(V.q2[1]),
                         "+v"(V.q0[2]), "+v"(V.q1[2]), "+v"(V.q2[2]));
#pragma unroll
        for (int u = 0; u < RB; ++u) {
            R.a[u][0] = V.q0[u].x; R.a[u][1] = V.q0[u].y; R.a[u][2] = V.q0[u].z; R.a[u][3] = V.q0[u].w;
            R.a[u][4] = V.q1[u].x; R.a[u][5] = V.q1[u].y; R.a[u][6] = V.q1[u].z; R.a[u][7] = V.q1[u].w;
#ifdef GSR_EXP_HI_TEST
            R.cbl[u] = V.q2[u].x; R.hi[u] = V.q2[u].y;
#else
            R.cbl[u] = V.q2[u]; R.hi[u] = 0.0f;
#endif
        }
    };
    using Behind9 = std::integral_constant<int, 9>;
    using Behind0 = std::integral_constant<int, 0>;

    // ---- the record-parallel pass of the fast blend (tail mode; the continuation kernel's re-evaluation of a pixel that closes inside a
    // chunk): lanes = the records c0 + lane of ONE chunk (r0, r1, r2; `valid`), the pixels of `todo` take turns two at a time.
    // T_j = T_p * prod_{i<=j} (1 - alpha_i) is one inclusive wave scan (transmittance only falls: "T >= 1e-4" is a prefix mask, the
    // pixel closes at its first failing lane), the colour is three wave sums of c * alpha * T_before; no loop over the hits.
    // Updates (Tw, C, last_q) of the pixels' own lanes; returns the pixels of `todo` that closed.
    auto tail_pairs = [&](unsigned long long todo, const int c0, const bool valid, const float4 r0, const float4 r1, const float4 r2) {
        unsigned long long closed_mask = 0ull;
        while (todo) {
            // two open pixels per pass (the second one a repeat of the first when only one is left): their scans and sums interleave,
            // which covers the two wait states a DPP operand needs behind its producer
            const int pa = __builtin_ctzll(todo);
            todo &= todo - 1;
            const bool two = todo != 0ull;
            const int pb = two ? __builtin_ctzll(todo) : pa;
            todo &= todo - 1;
            const int pp[2] = {pa, pb};
            float Tp[2], am[2], prod[2], okf[2];
            bool ok[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float ppx = (float)(tile_x * GSR_BLOCK_X + (wave & 1) * 8 + (pp[e] & 7));
                const float ppy = (float)(tile_y * GSR_BLOCK_Y + (wave >> 1) * 8 + (pp[e] >> 3));
                Tp[e] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(Tw), pp[e]));   // open: Tw == T
                const float dx = r0.x - ppx, dy = r0.y - ppy;
                const float power = __builtin_fmaf(__builtin_fmaf(r0.w, dy, r0.z * dx), dx, __builtin_fmaf(r1.x * dy, dy, r1.y));   // (the walk's expressions)
                const float raw = __builtin_amdgcn_exp2f(power);
#ifdef GSR_EXP_HI_TEST
                ok[e] = valid && __builtin_amdgcn_fmed3f(raw, 1.0f / 255.0f, r2.y) == raw;
#else
                ok[e] = valid && raw >= 1.0f / 255.0f;
#endif
                const float a = __builtin_amdgcn_fmed3f(raw, 0.0f, 0.99f);
                am[e] = ok[e] ? a : 0.0f;
                prod[e] = 1.0f - am[e];
                okf[e] = am[e] * __builtin_amdgcn_rcpf(prod[e]);   // alpha / (1 - alpha)
            }
            // inclusive prefix products over the lanes (v_mul_f32_dpp in place: lanes without a source keep theirs)
#define GSR_TAIL_STEP(CTRL) "v_mul_f32_dpp %0, %0, %0 " CTRL "\n\tv_mul_f32_dpp %1, %1, %1 " CTRL "\n\ts_nop 0\n\t"
            asm volatile("s_nop 1\n\t" GSR_TAIL_STEP("row_shr:1 row_mask:0xf bank_mask:0xf") GSR_TAIL_STEP("row_shr:2 row_mask:0xf bank_mask:0xf")
                         GSR_TAIL_STEP("row_shr:4 row_mask:0xf bank_mask:0xf") GSR_TAIL_STEP("row_shr:8 row_mask:0xf bank_mask:0xf")
                         GSR_TAIL_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf") GSR_TAIL_STEP("row_bcast:31 row_mask:0xc bank_mask:0xf")
                         : "+v"(prod[0]), "+v"(prod[1]));
#undef GSR_TAIL_STEP
            float Tfull[2], s3[2][3];
            bool keepl[2];
            unsigned long long K[2], hits[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                Tfull[e] = Tp[e] * prod[e];                         // transmittance AFTER the lane's record
                keepl[e] = Tfull[e] >= 0.0001f;                      // false from the record that would saturate the pixel onwards
                K[e] = __ballot(keepl[e]);
                hits[e] = __ballot(ok[e] && keepl[e]);
                const float wgt = keepl[e] ? Tfull[e] * okf[e] : 0.0f;   // alpha * T_before = T_after * alpha / (1 - alpha)
                s3[e][0] = r1.z * wgt; s3[e][1] = r1.w * wgt; s3[e][2] = r2.x * wgt;
            }
            // the six colour sums: halves first (v_permlane32_swap: lanes 0-31 keep pixel a's, lanes 32-63 pixel b's), then 32-lane sums
            float h3[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(s3[0][c]), __float_as_uint(s3[1][c]), false, false);
                float v = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
                v += dpp_f<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
                v += dpp_f<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
                v += dpp_f<0x141, 0xf>(v);  // row_half_mirror
                v += dpp_f<0x140, 0xf>(v);  // row_mirror
                v += dpp_f<0x142, 0xa>(v);  // row_bcast:15 -> rows 1, 3: lane 31 holds pixel a's sum, lane 63 pixel b's
                h3[c] = v;
            }
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                if (e == 1 && !two) break;
                const int src = e ? 63 : 31;
                const float A0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(h3[0]), src));
                const float A1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(h3[1]), src));
                const float A2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(h3[2]), src));
                const bool closed = K[e] != ~0ull;
                const int f = closed ? __builtin_ctzll(~K[e]) : 64;   // first lane whose record does not fit any more
                const float Tlast = f == 0 ? Tp[e] : __int_as_float(__builtin_amdgcn_readlane(__float_as_int(Tfull[e]), f - 1));
                if (lane == pp[e]) {
                    Tw = closed ? -Tlast : Tlast;
                    C0 += A0; C1 += A1; C2 += A2;
                    if (!INFER && hits[e]) last_q = (uint32_t)(c0 + 64 - __builtin_clzll(hits[e]));
                }
                if (closed) closed_mask |= 1ull << pp[e];
            }
        }
        return closed_mask;
    };

    // ---- the end of a quadrant: the backward's work units, the per-pixel state the backward reads, the image ---------------------------
    auto finish = [&]() {
        if (FAST && !fwd_only) {
            // the backward's work list (gsr.h: GsrImageLayout.units): one unit per 60-entry segment up to this quadrant's deepest last
            // contributor, appended to the list of this quadrant's launch position (one returning atomic per wave, spread over 64 counters)
            const uint32_t qmax = wave_max_u32(inside ? last_q : 0u);
            const uint32_t nseg = min((qmax + GSR_BWD_SEGMENT - 1u) / GSR_BWD_SEGMENT, (uint32_t)GSR_BWD_SEGMENTS);
            if (nseg) {
                const uint32_t w = lpos * 4u + (uint32_t)wave, list = w % (uint32_t)GSR_UNIT_LISTS;
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(units + 32u * list, nseg);   // (a 128-byte line per counter: returning atomics on one line queue up in L2)
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                if ((uint32_t)lane < nseg)
                    units[32u * GSR_UNIT_LISTS + list * ucap + base + (uint32_t)lane] = (uint32_t)tile << 6 | (uint32_t)wave << 4 | (uint32_t)lane;
            }
        }
        if (inside) {
            const int pix_id = W * pyi + pxi;
            // the reference's n_contrib counts positions in the TILE list: the parity modes keep those in a twin stream; in
            // production it is the position in the quadrant stream (nothing reads it: the backward walks n_contrib_q)
            const size_t HW = (size_t)H * W;
            if (!fwd_only) {   // what only the backward reads: 24 bytes per pixel (and the checkpoints above) less to store under torch.no_grad()
                const uint32_t last_contributor = !qlist ? last_q : (last_q ? (qlist + qs)[last_q - 1] + 1u : 0u);
                final_T[pix_id] = __builtin_fabsf(Tw);
                n_contrib[pix_id] = last_contributor;
                n_contrib_q[pix_id] = last_q;
                c_final[0 * HW + pix_id] = C0;
                c_final[1 * HW + pix_id] = C1;
                c_final[2 * HW + pix_id] = C2;
            }
            const float T = __builtin_fabsf(Tw);
            out_color[0 * HW + pix_id] = C0 + T * s.bg[0];
            out_color[1 * HW + pix_id] = C1 + T * s.bg[1];
            out_color[2 * HW + pix_id] = C2 + T * s.bg[2];
        }
    };

    if constexpr (CONT != 0) {
    if (helper) {
        // ================= continuation workgroups: parked quadrants, four chunks in flight =========================================================
        // CONT == 1: the LAST workgroups of k_render's own grid (dispatched behind every tile's workgroup, so they never keep one from starting):
        // they wait for quadrants to be parked while the tiles' walks are still running and leave when every tile wave has reported and the list is
        // drained.  CONT == 2: a kernel of its own behind k_render<true, 0> (everything is parked by then; nothing waits).
        __shared__ uint32_t s_pull;          // the parked quadrant this workgroup works on (~0: none left)
        __shared__ uint32_t s_seq;           // the chunk whose prefix state lies in s_hand (~0: the quadrant is finished)
        __shared__ uint32_t s_pending;       // re-evaluations of closing pixels still on their way to s_fin
        __shared__ float s_hand[5][64];      // (T signed as Tw, C0, C1, C2, last_q) of the 64 pixels before that chunk's first entry
        __shared__ float s_fin[5][64];       // the same five values of every pixel that has CLOSED (written once, by the wave that saw it close)
        const size_t HWc = (size_t)H * W;
        const uint32_t hidx = CONT == 1 ? blockIdx.x - (uint32_t)tiles : blockIdx.x, nhelp = CONT == 1 ? gridDim.x - (uint32_t)tiles : gridDim.x;
        bool first_pull = true;
        for (;;) {
            __syncthreads();                 // every wave is done with the previous quadrant (s_seq, s_hand, s_fin, the stages)
            if (threadIdx.x == 0) {
                // the first pull is the workgroup's own index (no atomic: 768 returning atomics on one word are 9 us of queueing), the later ones
                // come from the shared cursor behind those
                const uint32_t ei = first_pull ? hidx : nhelp + atomicAdd(cont_hdr + 32, 1u);
                uint32_t ent = 0xFFFFFFFFu;
                if (ei < 4u * (uint32_t)tiles) {
                    if constexpr (CONT == 2) {
                        if (ei < cont_hdr[0]) ent = cont_list[ei];
                    } else {
                        // wait for entry ei of the list, or for the proof that it will never come: every tile wave reports (after its own post, if it
                        // made one) on one of 16 counters; once they all have and the entry is still empty, the list ended below it
                        for (uint32_t spins = 0; spins < (1u << 24); ++spins) {
                            ent = __hip_atomic_load(cont_list + ei, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (ent != 0xFFFFFFFFu) break;
                            if ((spins & 7u) == 7u) {
                                uint32_t reported = 0;
                                for (int k = 0; k < 16; ++k) reported += __hip_atomic_load(cont_hdr + 64 + 32 * k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                if (reported >= 4u * (uint32_t)tiles) {
                                    ent = __hip_atomic_load(cont_list + ei, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                    break;
                                }
                            }
                            __builtin_amdgcn_s_sleep(8);
                        }
                    }
                }
                s_pull = ent;
                s_seq = (uint32_t)cont_c;
                s_pending = 0u;
            }
            first_pull = false;
            __syncthreads();
            const uint32_t ent = s_pull;
            if (ent == 0xFFFFFFFFu) return;
            lpos = ent >> 2;
            wave = (int)(ent & 3u);
            tile = (int)tile_order[lpos];
            tile_x = tile % gx; tile_y = tile / gx;
            pxi = tile_x * GSR_BLOCK_X + (wave & 1) * 8 + (lane & 7);
            pyi = tile_y * GSR_BLOCK_Y + (wave >> 1) * 8 + (lane >> 3);
            inside = pxi < W && pyi < H;
            pixx = (float)pxi; pixy = (float)pyi;
            n = (int)qcount[4 * tile + wave];
            qs = qstart[4 * tile + wave];
            qp = qpos + qs;
            const int pix_id = inside ? W * pyi + pxi : 0;
            {
                const float* st = cont_state + (size_t)(4 * tile + wave) * GSR_CONT_STATE_FLOATS;
                if constexpr (CONT == 1) {   // parked by a wave of THIS launch with write-through stores: read past this CU's L1
                    Tw = __hip_atomic_load(st + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    C0 = __hip_atomic_load(st + 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    C1 = __hip_atomic_load(st + 128 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    C2 = __hip_atomic_load(st + 192 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    last_q = __hip_atomic_load(reinterpret_cast<const uint32_t*>(st) + 256 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    Tw = st[lane]; C0 = st[64 + lane]; C1 = st[128 + lane]; C2 = st[192 + lane];
                    last_q = reinterpret_cast<const uint32_t*>(st)[256 + lane];
                }
            }
            auto to_fin = [&]() {   // this lane's pixel has closed: its five values, for whoever finishes the quadrant
                s_fin[0][lane] = Tw; s_fin[1][lane] = C0; s_fin[2][lane] = C1; s_fin[3][lane] = C2; s_fin[4][lane] = __uint_as_float(last_q);
            };
            if (pwave == 0) to_fin();        // (the pixels that were closed when the quadrant was parked; the open ones are rewritten when they close)
            __syncthreads();
            const int c_first = cont_c, nchunks = (n + CH - 1) / CH;
            unsigned long long myopen = __ballot(Tw > 0.0f);   // the open pixels as this wave last saw them
            bool have_g = false;
            for (int c = c_first + pwave; c < nchunks; c += NWV) {
                if (__hip_atomic_load(&s_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0xFFFFFFFFu) break;   // finished further up: nothing to add
                const int m = min(CH, n - c * CH);   // records of this chunk
                if (!have_g) gather(c);
                park(0);
                asm volatile("" ::: "memory");
                have_g = c + NWV < nchunks;
                if (have_g) gather(c + NWV);         // in flight behind this chunk's walk
                // wait for the state before this chunk's first entry; false: the quadrant finished further up
                auto wait_prefix = [&]() {
                    for (uint32_t spins = 0; spins < (1u << 22); ++spins) {   // (the chunk below is always on its way: the bound only keeps a logic error from hanging the device)
                        const uint32_t v = __hip_atomic_load(&s_seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                        if (v == 0xFFFFFFFFu) return false;
                        if (v >= (uint32_t)c) return true;
                        __builtin_amdgcn_s_sleep(1);
                    }
                    return false;
                };
                const bool first = c == c_first;
                const bool spec = !first && __builtin_popcountll(myopen) > TAIL_LANES;
                bool walk = spec;
                if (spec) {
                    // walked from T = 1 (pixels this wave knows to be closed: from 0) while the chunks below are still on their way
                    Tw = ((myopen >> lane) & 1ull) ? 1.0f : 0.0f;
                    C0 = 0.f; C1 = 0.f; C2 = 0.f;
                    lq = 0; j0 = c * CH;
                } else {
                    if (!first) {
                        if (!wait_prefix()) break;
                        Tw = s_hand[0][lane]; C0 = s_hand[1][lane]; C1 = s_hand[2][lane]; C2 = s_hand[3][lane];
                        last_q = __float_as_uint(s_hand[4][lane]);
                    }
                    myopen = __ballot(Tw > 0.0f);
                    walk = __builtin_popcountll(myopen) > TAIL_LANES;   // (only the first chunk can get here with many pixels open)
                    j0 = c * CH;
                    lq = (int)last_q - j0;
                }
                if (walk) {
                    // the chunk's records, one batch of three per read, a double batch per step (the lone walk's loop without its tests: pixels that are
                    // closed are inert by construction)
                    RecV VA, VB;
                    Rec4 A, B;
                    uint32_t addr = lds0;
                    GSR_ISSUE(addr, 0, VA);
                    int k = 0;
                    while (k + 2 * RB <= m) {
                        GSR_ISSUE(addr, BB, VB);
                        ready(VA, Behind9{}, A);
                        blend4(j0, A, std::false_type{}, Off0{});
                        GSR_ISSUE(addr, 2 * BB, VA);
                        ready(VB, Behind9{}, B);
                        blend4(j0 + RB, B, std::false_type{}, Off1{});
                        addr += 2 * BB;
                        j0 += 2 * RB;
                        lq -= 2 * RB;
                        k += 2 * RB;
                    }
                    ready(VA, Behind0{}, A);
                    if (k < m) {   // the stream ends inside this chunk: one or two bounds-tested batches (A holds the first)
                        blend4(j0, A, std::true_type{}, Off0{});
                        j0 += RB;
                        lq -= RB;
                        if (j0 < n) {
                            GSR_ISSUE(addr, BB, VB);
                            ready(VB, Behind0{}, B);
                            blend4(j0, B, std::true_type{}, Off0{});
                            j0 += RB;
                            lq -= RB;
                        }
                    }
                }
                unsigned long long todo;         // pixels to evaluate with the record-parallel pass, from the TRUE state in their lanes
                const unsigned long long open_before = myopen;
                bool last;                        // the state after this chunk is the quadrant's result
                if (spec) {
                    const float Tl = Tw, L0 = C0, L1 = C1, L2 = C2;
                    const uint32_t hit = (uint32_t)(lq + j0);                       // > c * CH: the chunk's last contributor of this pixel
                    if (!wait_prefix()) break;
                    const float Tp = s_hand[0][lane], P0 = s_hand[1][lane], P1 = s_hand[2][lane], P2 = s_hand[3][lane];
                    const uint32_t lqp = __float_as_uint(s_hand[4][lane]);
                    const bool open_p = Tp > 0.0f;
                    const float Tt = Tp * Tl;
                    const bool thru = open_p && Tl > 0.0f && Tt >= 0.0001f;       // open before the chunk and no record of it closes the pixel
                    todo = __ballot(open_p && !thru);                              // these close inside the chunk
                    myopen = __ballot(thru);
                    last = myopen == 0ull || c == nchunks - 1;
                    if (!last) {
                        // the chunks above only need to know WHICH pixels are still open: the state goes down the chain before the closing pixels
                        // are evaluated (their values go to s_fin, off the chain)
                        if (todo && lane == 0) __hip_atomic_fetch_add(&s_pending, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        s_hand[0][lane] = thru ? Tt : -1.0f;
                        s_hand[1][lane] = __builtin_fmaf(Tp, L0, P0); s_hand[2][lane] = __builtin_fmaf(Tp, L1, P1); s_hand[3][lane] = __builtin_fmaf(Tp, L2, P2);
                        s_hand[4][lane] = __uint_as_float(hit > (uint32_t)(c * CH) ? hit : lqp);
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        if (lane == 0) __hip_atomic_store(&s_seq, (uint32_t)(c + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                        if (!fwd_only && c + 1 <= GSR_BWD_SEGMENTS - 1 && inside && thru)   // (the backward reads a checkpoint only where the pixel goes on)
                            ck[(size_t)c * HWc + pix_id] = make_float4(Tt, __builtin_fmaf(Tp, L0, P0), __builtin_fmaf(Tp, L1, P1), __builtin_fmaf(Tp, L2, P2));
                    }
                    Tw = thru ? Tt : Tp;
                    C0 = thru ? __builtin_fmaf(Tp, L0, P0) : P0;
                    C1 = thru ? __builtin_fmaf(Tp, L1, P1) : P1;
                    C2 = thru ? __builtin_fmaf(Tp, L2, P2) : P2;
                    last_q = (thru && hit > (uint32_t)(c * CH)) ? hit : lqp;
                } else {
                    if (walk) last_q = (uint32_t)(lq + j0);
                    todo = walk ? 0ull : myopen;
                }
                if (todo) {   // records across the lanes, the pixels two at a time
                    const float4* sp = &stage[0][3 * min(lane, CH - 1)];
                    (void)tail_pairs(todo, c * CH, lane < m, sp[0], sp[1], sp[2]);
                }
                if (spec) {
                    if (todo) {
                        if ((todo >> lane) & 1ull) to_fin();
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        if (!last && lane == 0) __hip_atomic_fetch_sub(&s_pending, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                } else {
                    myopen = __ballot(Tw > 0.0f);
                    if (((open_before & ~myopen) >> lane) & 1ull) to_fin();   // closed in this chunk
                    last = myopen == 0ull || c == nchunks - 1;
                    if (!last) {
                        if (!fwd_only && c + 1 <= GSR_BWD_SEGMENTS - 1 && inside)
                            ck[(size_t)c * HWc + pix_id] = make_float4(__builtin_fabsf(Tw), C0, C1, C2);
                        s_hand[0][lane] = Tw; s_hand[1][lane] = C0; s_hand[2][lane] = C1; s_hand[3][lane] = C2; s_hand[4][lane] = __uint_as_float(last_q);
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        if (lane == 0) __hip_atomic_store(&s_seq, (uint32_t)(c + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
                if (last) {
                    // every pixel that closed on the way left its values in s_fin (the waves still evaluating theirs are counted in s_pending)
                    for (uint32_t spins = 0; spins < (1u << 22); ++spins) {
                        if (__hip_atomic_load(&s_pending, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == 0u) break;
                        __builtin_amdgcn_s_sleep(1);
                    }
                    if (!(Tw > 0.0f)) {
                        Tw = s_fin[0][lane]; C0 = s_fin[1][lane]; C1 = s_fin[2][lane]; C2 = s_fin[3][lane];
                        last_q = __float_as_uint(s_fin[4][lane]);
                    }
                    finish();
                    if (lane == 0) __hip_atomic_store(&s_seq, 0xFFFFFFFFu, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    break;
                }
            }
        }
    }
    }
    {
    auto keep_going = [&](int jb) {
        const unsigned long long open_mask = __ballot(Tw > 0.0f);
        int open;    // (through asm: the compiler widens popcountll's comparison to 64 bits and then does it on the VECTOR unit)
        asm("s_bcnt1_i32_b64 %0, %1" : "=s"(open) : "s"(open_mask) : "scc");
        if (open > TAIL_LANES) return true;
        return open != 0 && n - jb <= 2 * GSR_WAVE;   // nothing open: stop; few open pixels and a long way to go: tail mode
    };
    // Software-pipelined walk: the scalar loads of batch k+1 are issued BEFORE batch k is blended, right after the wait for
    // batch k's own loads (issued a whole batch ago) -- placed the other way round the wait would stall on the fresh loads.
    // Blend checkpoints for the backward (include/gsr.h, GsrImageLayout.ck): the state (T, C) of every pixel just before
    // stream entry s * GSR_BWD_SEGMENT, s = 1 .. GSR_BWD_SEGMENTS-1.  GSR_BWD_SEGMENT is a multiple of the two batches one
    // iteration of the walk takes, so the test sits at the top of the loop only.  The store is unconditional: lanes outside
    // the image aim at the spare slot behind the array (the layout's tail padding), closed pixels rewrite slots nobody reads.
    static_assert(GSR_BWD_SEGMENT % (2 * RB) == 0, "checkpoints sit on iteration boundaries of the walk");
    const size_t HWs = (size_t)H * W;
    float4* ck_ptr = ck + (inside ? (size_t)(W * pyi + pxi) : (size_t)(GSR_BWD_SEGMENTS - 1) * HWs);
    const size_t ck_step = inside ? HWs : 0;
    int next_ck = fwd_only ? 0x7fffffff : GSR_BWD_SEGMENT;   // (forward_only: no backward will read a checkpoint -- the test below never fires)
    auto checkpoint = [&](int jtop) {
        if (jtop == next_ck) {
            if (next_ck <= (GSR_BWD_SEGMENTS - 1) * GSR_BWD_SEGMENT) {
                *ck_ptr = make_float4(__builtin_fabsf(Tw), C0, C1, C2);
                ck_ptr += ck_step;
            }
            next_ck += GSR_BWD_SEGMENT;
        }
    };
    // a lone walk that reaches the hand-over chunk parks its state for the continuation kernel and leaves (nothing of the quadrant's
    // outputs is written here; the checkpoint before the chunk has been)
    auto park_quadrant = [&](uint32_t lastq_abs) {
        int slot = 4 * tile + wave;
        asm volatile("" : "+s"(slot));   // (the addresses are formed HERE: hoisted out of the walk they cost it eight vector registers and a wave per SIMD)
        float* st = cont_state + (size_t)slot * GSR_CONT_STATE_FLOATS;
        if constexpr (CONT == 1) {
            // read by a workgroup of THIS launch, on any XCD: write-through stores, drained, then the list entry (MI355X_MICROARCH.md, hand-off forms)
            __hip_atomic_store(st + lane, Tw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(st + 64 + lane, C0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(st + 128 + lane, C1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(st + 192 + lane, C2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(reinterpret_cast<uint32_t*>(st) + 256 + lane, lastq_abs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) {
                const uint32_t at = atomicAdd(cont_hdr, 1u);
                __hip_atomic_store(cont_list + at, lpos << 2 | (uint32_t)wave, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the entry is out before this wave reports: a waiting workgroup that has seen every report has seen every entry)
            }
        } else {
            st[lane] = Tw; st[64 + lane] = C0; st[128 + lane] = C1; st[192 + lane] = C2;
            reinterpret_cast<uint32_t*>(st)[256 + lane] = lastq_abs;
            if (lane == 0) cont_list[atomicAdd(cont_hdr, 1u)] = lpos << 2 | (uint32_t)wave;
        }
    };
    auto report = [&]() {   // CONT == 1: this tile wave is done (parked or finished): one of 16 counters the waiting continuation workgroups add up
        if constexpr (CONT == 1) {
            if (lane == 0) (void)__hip_atomic_fetch_add(cont_hdr + 64 + 32 * (blockIdx.x & 15u), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    bool hand_over = false;   // the walk reached the hand-over chunk with the stream going on
    if (n > 0) {
        gather(0);
        landed();
        park(0);
        if (n > CH) gather(1);
        bool go = true;
        for (int c = 0;; ++c) {
            const int b = c & 1;
            const int m = min(CH, n - c * CH);   // records of this chunk
            checkpoint(j0);                      // j0 == c * CH: the state before the chunk's first entry
            if (FAST && c == cont_c) { hand_over = true; break; }
            RecV VA, VB;
            Rec4 A, B;
            uint32_t addr = b ? lds1 : lds0;
            GSR_ISSUE(addr, 0, VA);
            int k = 0;
            // whole double batches: nothing masked.  The test at the BOTTOM of the loop (one block, the state in place: with the test at the
            // top the compiler copied the six state registers in and out of the body) is the cheap half of keep_going -- more than
            // TAIL_LANES pixels open --; what to do when it fails (stop, tail mode, or walk on because the stream is nearly over) is
            // decided outside, and the rare "walk on" continues in the second loop with the full test.
            auto many_open = [&]() {
                int open;
                asm("s_bcnt1_i32_b64 %0, %1" : "=s"(open) : "s"(__ballot(Tw > 0.0f)) : "scc");
                return open > TAIL_LANES;
            };
            auto double_batch = [&]() {
                GSR_ISSUE(addr, BB, VB);
                ready(VA, Behind9{}, A);
                blend4(j0, A, std::false_type{}, Off0{});
                GSR_ISSUE(addr, 2 * BB, VA);   // the next double batch's (or the remainder's) first batch; at a chunk's end: one batch past it, unused
                ready(VB, Behind9{}, B);
                blend4(j0 + RB, B, std::false_type{}, Off1{});
                addr += 2 * BB;
                j0 += 2 * RB;
                lq -= 2 * RB;
                k += 2 * RB;
            };
            if (2 * RB <= m && many_open()) {
                do double_batch(); while (k + 2 * RB <= m && many_open());
            }
            while (k + 2 * RB <= m && (go = keep_going(j0))) double_batch();   // (few pixels open, the stream nearly over)
            ready(VA, Behind0{}, A);   // (always: the reads issued last land in registers the compiler considers written)
            if (go && k < m) {   // the stream ends inside this chunk: one or two bounds-tested batches (A holds the first)
                if (keep_going(j0)) {
                    blend4(j0, A, std::true_type{}, Off0{});
                    j0 += RB;
                    lq -= RB;
                    if (j0 < n) {
                        GSR_ISSUE(addr, BB, VB);
                        ready(VB, Behind0{}, B);
                        blend4(j0, B, std::true_type{}, Off0{});
                        j0 += RB;
                        lq -= RB;
                    }
                } else {
                    go = false;
                }
            }
            if (!go || j0 >= n) break;
            landed();   // (DIRECT: chunk c + 1 is in its buffer; the gather below goes into the buffer this chunk has just been walked out of)
            park(c + 1);
            if ((c + 2) * CH < n) gather(c + 2);
        }
    }
    last_q = (uint32_t)(lq + j0);   // (lq + j0 >= 0: a pixel without a hit kept lq = -j0)
    if (hand_over) {
        park_quadrant(last_q);
        report();
        return;
    }

    if (j0 < n) checkpoint(j0);   // the walk stopped exactly on a checkpoint entry (tail mode takes over from here)
#ifdef GSR_EXPERIMENT_TIMELINE
    const unsigned long long t_main = wall_clock64();
    const int j_main = j0;
#endif
    // ---- tail mode: record-parallel --------------------------------------------------------------
    // A few pixels that never saturate (silhouettes) would otherwise drag the whole wave through the
    // rest of a long stream with 60 idle lanes.  Here the roles flip: for one open pixel at a time the
    // 64 lanes evaluate 64 *records* in parallel, and only the records that actually touch the pixel
    // (ballot) go through the sequential T/C update, in stream order -- the arithmetic per contributing
    // record and its order are unchanged, so results stay bit-identical.
    if (FAST && j0 < n) {
        // ---- fast blend: chunks of the stream up to the next checkpoint boundary, lanes = records, the open pixels two at a time (tail_pairs)
        unsigned long long open_mask = __ballot(Tw > 0.0f);
        int c0 = j0;
        while (c0 < n && open_mask) {
            if (c0 % CH == 0 && c0 / CH == cont_c) {   // (the tail's chunks end on checkpoint entries: the hand-over entry is the top of one)
                park_quadrant(last_q);
                report();
                return;
            }
            const int c1 = min(n, (c0 / GSR_BWD_SEGMENT + 1) * GSR_BWD_SEGMENT);   // the chunk ends where the next checkpoint sits (<= 60 entries)
            const int j = c0 + lane;
            const bool valid = j < c1;
            const size_t jc = (size_t)qp[valid ? j : c1 - 1];
            const float4 r0 = rec[3 * jc + 0];
            const float4 r1 = rec[3 * jc + 1];
            const float4 r2 = rec[3 * jc + 2];
            open_mask &= ~tail_pairs(open_mask, c0, valid, r0, r1, r2);
            c0 = c1;
            if (c0 < n) checkpoint(c0);   // (every lane stores its own pixel's state; closed pixels rewrite slots nobody reads)
        }
    }
    if (!FAST && j0 < n) {
        unsigned long long open_mask = __ballot(Tw > 0.0f);
        while (open_mask) {
            const int p = __builtin_ctzll(open_mask);
            open_mask &= open_mask - 1;
            const float ppx = (float)(tile_x * GSR_BLOCK_X + (wave & 1) * 8 + (p & 7));
            const float ppy = (float)(tile_y * GSR_BLOCK_Y + (wave >> 1) * 8 + (p >> 3));
            float Tp = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(Tw), p));   // open: Tw == T
            float A0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(C0), p));
            float A1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(C1), p));
            float A2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(C2), p));
            uint32_t lastqp = (uint32_t)__builtin_amdgcn_readlane((int)last_q, p);
            bool donep = false;
            int e_ck = next_ck;                               // this pixel's next checkpoint (the main loop stored the earlier ones)
            float4* at_ck = ck + (size_t)((fwd_only ? GSR_BWD_SEGMENT : e_ck) / GSR_BWD_SEGMENT - 1) * HWs + (size_t)(W * (int)ppy + (int)ppx);
            for (int c0 = j0; c0 < n && !donep; c0 += GSR_WAVE) {
                const int j = c0 + lane;
                const bool valid = j < n;
                const size_t jc = (size_t)qp[valid ? j : n - 1];
                const float4 r0 = rec[3 * jc + 0];
                const float4 r1 = rec[3 * jc + 1];
                const float4 r2 = rec[3 * jc + 2];
                const float dx = r0.x - ppx;
                const float dy = r0.y - ppy;
                float power, a;
                if constexpr (FAST) {
                    power = __builtin_fmaf(__builtin_fmaf(r0.w, dy, r0.z * dx), dx, (r1.x * dy) * dy);
                    a = __builtin_fminf(0.99f, r1.y * __builtin_amdgcn_exp2f(power));
                } else {
                    power = -0.5f * (r0.z * dx * dx + r1.x * dy * dy) - r0.w * dx * dy;
                    a = sel_min(0.99f, r1.y * gsr_expf(power));
                }
                const bool ok = valid && power <= 0.0f && a >= 1.0f / 255.0f;
                unsigned long long hits = __ballot(ok);
                while (hits) {
                    const int k = __builtin_ctzll(hits);
                    hits &= hits - 1;
                    while (c0 + k >= e_ck && e_ck <= (GSR_BWD_SEGMENTS - 1) * GSR_BWD_SEGMENT) {   // state before entry e_ck = after every hit below it
                        if (lane == 0) *at_ck = make_float4(Tp, A0, A1, A2);
                        e_ck += GSR_BWD_SEGMENT;
                        at_ck += HWs;
                    }
                    const float ak = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a), k));
                    const float test_T = FAST ? __builtin_fmaf(-ak, Tp, Tp) : Tp * (1.0f - ak);   // (the same expression as the main walk: a pixel's result does not depend on where the tail took over)
                    if (test_T < 0.0001f) { donep = true; break; }
                    if constexpr (FAST) {
                        const float wgt = ak * Tp;
                        A0 = __builtin_fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(r1.z), k)), wgt, A0);
                        A1 = __builtin_fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(r1.w), k)), wgt, A1);
                        A2 = __builtin_fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(r2.x), k)), wgt, A2);
                    } else {
                    A0 += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r1.z), k)) * ak * Tp;
                    A1 += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r1.w), k)) * ak * Tp;
                    A2 += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r2.x), k)) * ak * Tp;
                    }
                    Tp = test_T;
                    lastqp = (uint32_t)(c0 + k + 1);
                }
            }
            if (lane == p) { Tw = Tp; C0 = A0; C1 = A1; C2 = A2; last_q = lastqp; }   // (|Tw| is the pixel's T; nothing reads the sign from here on)
        }
    }
#ifdef GSR_EXPERIMENT_TIMELINE
    if (lane == 0 && (size_t)lpos * 4 + wave < 16384) {
        unsigned long long* d = gsr_dbg_fwd + 4 * ((size_t)lpos * 4 + wave);
        d[0] = t_start;
        d[1] = wall_clock64();
        d[2] = ((unsigned long long)(uint32_t)n << 32) | (uint32_t)j_main;
        d[3] = t_main;
    }
#endif
    finish();
    report();
    }   // (a tile's workgroup)
}

template __global__ void k_render<false, 0, false>(Settings, const uint32_t*, const uint32_t*, const uint32_t*, const float4*, const uint32_t*, const uint32_t*, float*, uint32_t*,
                                                uint32_t*, float*, float4*, float*, unsigned long long, const unsigned long long*, uint32_t*, int);
template __global__ void k_render<true, 0, false>(Settings, const uint32_t*, const uint32_t*, const uint32_t*, const float4*, const uint32_t*, const uint32_t*, float*, uint32_t*,
                                                uint32_t*, float*, float4*, float*, unsigned long long, const unsigned long long*, uint32_t*, int);
template __global__ void k_render<true, 2, false>(Settings, const uint32_t*, const uint32_t*, const uint32_t*, const float4*, const uint32_t*, const uint32_t*, float*, uint32_t*,
                                                uint32_t*, float*, float4*, float*, unsigned long long, const unsigned long long*, uint32_t*, int);
template __global__ void k_render<true, 1, false>(Settings, const uint32_t*, const uint32_t*, const uint32_t*, const float4*, const uint32_t*, const uint32_t*, float*, uint32_t*,
                                                uint32_t*, float*, float4*, float*, unsigned long long, const unsigned long long*, uint32_t*, int);
template __global__ void k_render<false, 0, true>(Settings, const uint32_t*, const uint32_t*, const uint32_t*, const float4*, const uint32_t*, const uint32_t*, float*, uint32_t*,
                                                uint32_t*, float*, float4*, float*, unsigned long long, const unsigned long long*, uint32_t*, int);
template __global__ void k_render<true, 0, true>(Settings, const uint32_t*, const uint32_t*, const uint32_t*, const float4*, const uint32_t*, const uint32_t*, float*, uint32_t*,
                                                uint32_t*, float*, float4*, float*, unsigned long long, const unsigned long long*, uint32_t*, int);

// ------------------------------------------------------------------------------------------
// k_mark_visible (upstream checkFrustum): present = view z > 0.2
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_mark_visible(int P, const float* __restrict__ means3D,
                                                       const float* __restrict__ vm, uint8_t* __restrict__ present)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const float mx = means3D[3 * i], my = means3D[3 * i + 1], mz = means3D[3 * i + 2];
    const float vz = vm[2] * mx + vm[6] * my + vm[10] * mz + vm[14];
    present[i] = vz > 0.2f ? 1 : 0;
}

}  // namespace gsr
