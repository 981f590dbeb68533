// gab_kernels.hip -- fused HIP kernels (gfx950, wave64) + C ABI (include/gab.h) for the FLAME-rigged
// binding half of the GaussianAvatars hot path.  The reference executes this half as ~200 ATen
// launches per frame; here it is 3 (FLAME) + 1 (face frames) + 1 (splats) launches forward and
// 3 + 1 + 1 backward; on the prepared rig (gab_flame_prepare) 1 + 1 forward and, for the mesh node, 2 backward
// (k_gather_skin_bwd: face-frame gather + skinning; k_chain_blend_bwd), the splats riding in the rasterizer's own kernels.  Every stage is HBM-/latency-bound (no GEMM-shaped work at batch 1 except
// the 24.7 MB blend-shape GEMV, which is a pure streaming read), so the kernels are plain
// coalesced VALU code: one wave per blend-shape row, one thread per vertex / face / splat.
//
// Reference semantics restated here (file:line relative to /root/reference):
//   flame_model/flame.py:511-536, flame_model/lbs.py:25-57,101-195,218-304,
//   utils/graphics_utils.py:90-135, scene/flame_gaussian_model.py:137-154,
//   scene/gaussian_model.py:113-150, roma rotmat_to_unitquat / quat_product.
#include <hip/hip_runtime.h>
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

#include <cstdarg>
#include <cstdio>

#include "../../include/gab.h"
#include "launch_prof.h"
#include "bind_math.h"

namespace gab {

struct Rig {
    int V, n_shape, n_expr;
    const float* __restrict__ v_template;
    const float* __restrict__ shapedirs;
    const float* __restrict__ posedirs;
    const float* __restrict__ J_regressor;
    const float* __restrict__ lbs_weights;
    int parents[GAB_NUM_JOINTS];
};

// workspace offsets (floats)
constexpr int WS_J = 0;      // 15
constexpr int WS_R = 16;     // 45
constexpr int WS_PF = 64;    // 36
constexpr int WS_A = 128;    // 60  A[j][r*4+c]
constexpr int WS_DA = 256;   // 60
constexpr int WS_DT = 320;   // 3
constexpr int WS_DPF = 324;  // 36
constexpr int WS_DJ = 384;   // 15

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
// 64-lane sum; the result is valid in lanes 48..63
__device__ __forceinline__ float wave_sum_hi(float v)
{
    v += dpp_f<0xB1, 0xf>(v);
    v += dpp_f<0x4E, 0xf>(v);
    v += dpp_f<0x141, 0xf>(v);
    v += dpp_f<0x140, 0xf>(v);
    v += dpp_f<0x142, 0xa>(v);
    v += dpp_f<0x143, 0xc>(v);
    return v;
}

__device__ __forceinline__ float beta_at(const float* shape, const float* expr, int n_shape, int l)
{
    return l < n_shape ? shape[l] : expr[l - n_shape];
}

// ---------------------------------------------------------------------------------------------
// F1  v_shaped = v_template + shapedirs . [shape|expr] (+ static_offset): one wave per row of betas
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_blend(Rig rig, const float* __restrict__ shape, const float* __restrict__ expr,
                                                const float* __restrict__ static_offset, float* __restrict__ v_shaped)
{
    const int lane = threadIdx.x & 63;
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int E = 3 * rig.V;
    if (e >= E) return;
    const int NB = rig.n_shape + rig.n_expr;
    const float* row = rig.shapedirs + (size_t)e * NB;
    float acc = 0.f;
    if ((NB & 3) == 0 && (rig.n_shape & 3) == 0) {
        const float4* row4 = reinterpret_cast<const float4*>(row);
        for (int c = lane; c < NB / 4; c += 64) {
            const float4 r = row4[c];
            const int l = 4 * c;
            const float4 b = l < rig.n_shape ? reinterpret_cast<const float4*>(shape)[c]
                                             : reinterpret_cast<const float4*>(expr)[c - rig.n_shape / 4];
            acc += r.x * b.x + r.y * b.y + r.z * b.z + r.w * b.w;
        }
    } else {
        for (int l = lane; l < NB; l += 64) acc += row[l] * beta_at(shape, expr, rig.n_shape, l);
    }
    acc = wave_sum_hi(acc);
    if (lane == 63) v_shaped[e] = rig.v_template[e] + acc + (static_offset ? static_offset[e] : 0.f);
}

// ---------------------------------------------------------------------------------------------
// small dense helpers for the 5-joint chain (thread-serial; 5 joints)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void rodrigues(const float* r, float* R /*9*/)
{
    const float ux = r[0] + 1e-8f, uy = r[1] + 1e-8f, uz = r[2] + 1e-8f;   // epsilon on the vector, then the norm
    const float th = sqrtf(ux * ux + uy * uy + uz * uz);
    const float dx = r[0] / th, dy = r[1] / th, dz = r[2] / th;
    const float s = sinf(th), c = cosf(th);
    const float K[9] = {0.f, -dz, dy, dz, 0.f, -dx, -dy, dx, 0.f};
    float KK[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) KK[3 * i + j] = K[3 * i] * K[j] + K[3 * i + 1] * K[3 + j] + K[3 * i + 2] * K[6 + j];
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = ((i % 4) == 0 ? 1.f : 0.f) + s * K[i] + (1.f - c) * KK[i];
}

// gradient of rodrigues w.r.t. the axis-angle vector
__device__ __forceinline__ void rodrigues_bwd(const float* r, const float* dR /*9*/, float* dr /*3*/)
{
    const float u[3] = {r[0] + 1e-8f, r[1] + 1e-8f, r[2] + 1e-8f};
    const float th = sqrtf(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
    const float d[3] = {r[0] / th, r[1] / th, r[2] / th};
    const float s = sinf(th), c = cosf(th);
    const float K[9] = {0.f, -d[2], d[1], d[2], 0.f, -d[0], -d[1], d[0], 0.f};
    float KK[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) KK[3 * i + j] = K[3 * i] * K[j] + K[3 * i + 1] * K[3 + j] + K[3 * i + 2] * K[6 + j];
    float g_s = 0.f, g_c = 0.f;
#pragma unroll
    for (int i = 0; i < 9; ++i) { g_s += dR[i] * K[i]; g_c -= dR[i] * KK[i]; }
    // dL/dK = s dR + (1-c) (dR K^T + K^T dR)
    float dK[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                a += dR[3 * i + k] * K[3 * j + k];   // dR K^T
                b += K[3 * k + i] * dR[3 * k + j];   // K^T dR
            }
            dK[3 * i + j] = s * dR[3 * i + j] + (1.f - c) * (a + b);
        }
    const float gd[3] = {dK[7] - dK[5], dK[2] - dK[6], dK[3] - dK[1]};
    float g_th = g_s * c - g_c * s;
    g_th -= (gd[0] * r[0] + gd[1] * r[1] + gd[2] * r[2]) / (th * th);
#pragma unroll
    for (int k = 0; k < 3; ++k) dr[k] = gd[k] / th + g_th * u[k] / th;
}

// forward chain: G_i = G_parent [R_i | J_i - J_parent];  A_i = [Rg_i | tg_i - Rg_i J_i]
// FLAME's kinematic tree.  The serial chain code below indexes small local arrays by parent id; with the
// tree known at compile time (and every loop unrolled) those arrays stay in registers, a run-time tree
// sends them to scratch memory (~10x slower for the one lane that runs the chain).
__device__ constexpr int kFlameParents[GAB_NUM_JOINTS] = {-1, 0, 1, 1, 1};
template <bool FLAME_TREE>
__device__ __forceinline__ int parent_of(const int* parents, int i) { return FLAME_TREE ? kFlameParents[i] : parents[i]; }

template <bool FLAME_TREE>
__device__ __forceinline__ void chain_forward(const int* parents, const float* R /*5x9*/, const float* J /*5x3*/, float* Rg /*5x9*/,
                                              float* tg /*5x3*/)
{
#pragma unroll
    for (int i = 0; i < GAB_NUM_JOINTS; ++i) {
        if (i == 0) {
#pragma unroll
            for (int k = 0; k < 9; ++k) Rg[k] = R[k];
#pragma unroll
            for (int k = 0; k < 3; ++k) tg[k] = J[k];
            continue;
        }
        const int p = parent_of<FLAME_TREE>(parents, i);
        const float rel[3] = {J[3 * i] - J[3 * p], J[3 * i + 1] - J[3 * p + 1], J[3 * i + 2] - J[3 * p + 2]};
#pragma unroll
        for (int r = 0; r < 3; ++r) {
#pragma unroll
            for (int c = 0; c < 3; ++c)
                Rg[9 * i + 3 * r + c] = Rg[9 * p + 3 * r] * R[9 * i + c] + Rg[9 * p + 3 * r + 1] * R[9 * i + 3 + c] +
                                        Rg[9 * p + 3 * r + 2] * R[9 * i + 6 + c];
            tg[3 * i + r] = Rg[9 * p + 3 * r] * rel[0] + Rg[9 * p + 3 * r + 1] * rel[1] + Rg[9 * p + 3 * r + 2] * rel[2] + tg[3 * p + r];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// F2  joints J = J_regressor . v_shaped (15 sums over V), Rodrigues, kinematic chain -> ws
// ---------------------------------------------------------------------------------------------
template <bool FLAME_TREE>
__global__ __launch_bounds__(1024) void k_joints_chain(Rig rig, const float* __restrict__ v_shaped, const float* __restrict__ rotation,
                                                        const float* __restrict__ neck, const float* __restrict__ jaw,
                                                        const float* __restrict__ eyes, float* __restrict__ ws)
{
    // One workgroup (the 15 sums feed a 5-joint serial chain), but a wide one: 1024 lanes keep the whole regression in
    // flight at once (5-6 vertices per lane), Rodrigues runs on five lanes in parallel, only the chain itself is serial,
    // and the 256-float forward half of the workspace is assembled in LDS and stored with one coalesced pass.
    __shared__ float red[16][16];
    __shared__ float out[WS_DA];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    // the backward's accumulators live in the same workspace: hand them over zeroed (k_chain_bwd zeroes them again
    // after consuming them), so no memset precedes the backward's atomics
    if (tid < GAB_FLAME_WS_FLOATS - WS_DA) ws[WS_DA + tid] = 0.f;
    if (tid < WS_DA) out[tid] = 0.f;
    float acc[15];
#pragma unroll
    for (int k = 0; k < 15; ++k) acc[k] = 0.f;
    for (int v = tid; v < rig.V; v += 1024) {
        const float x = v_shaped[3 * v], y = v_shaped[3 * v + 1], z = v_shaped[3 * v + 2];
#pragma unroll
        for (int j = 0; j < GAB_NUM_JOINTS; ++j) {
            const float w = rig.J_regressor[(size_t)j * rig.V + v];
            acc[3 * j] += w * x;
            acc[3 * j + 1] += w * y;
            acc[3 * j + 2] += w * z;
        }
    }
#pragma unroll
    for (int k = 0; k < 15; ++k) {
        const float s = wave_sum_hi(acc[k]);
        if (lane == 63) red[wid][k] = s;
    }
    __syncthreads();
    if (tid < 15) {
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) sum += red[w][tid];
        out[WS_J + tid] = sum;
    } else if (tid >= 64 && tid < 64 + GAB_NUM_JOINTS) {
        const int j = tid - 64;
        const float* pj = j == 0 ? rotation : j == 1 ? neck : j == 2 ? jaw : eyes + 3 * (j - 3);
        const float pose[3] = {pj[0], pj[1], pj[2]};
        float R[9];
        rodrigues(pose, R);
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            out[WS_R + 9 * j + k] = R[k];
            if (j >= 1) out[WS_PF + 9 * (j - 1) + k] = R[k] - ((k % 4) == 0 ? 1.f : 0.f);
        }
    }
    __syncthreads();
    if (tid == 0) {
        float J[15], R[45], Rg[45], tg[15];
#pragma unroll
        for (int k = 0; k < 15; ++k) J[k] = out[WS_J + k];
#pragma unroll
        for (int k = 0; k < 45; ++k) R[k] = out[WS_R + k];
        chain_forward<FLAME_TREE>(rig.parents, R, J, Rg, tg);
#pragma unroll
        for (int j = 0; j < GAB_NUM_JOINTS; ++j)
#pragma unroll
            for (int r = 0; r < 3; ++r) {
#pragma unroll
                for (int c = 0; c < 3; ++c) out[WS_A + 12 * j + 4 * r + c] = Rg[9 * j + 3 * r + c];
                out[WS_A + 12 * j + 4 * r + 3] = tg[3 * j + r] - (Rg[9 * j + 3 * r] * J[3 * j] + Rg[9 * j + 3 * r + 1] * J[3 * j + 1] +
                                                                  Rg[9 * j + 3 * r + 2] * J[3 * j + 2]);
            }
    }
    __syncthreads();
    if (tid < WS_DA) ws[tid] = out[tid];
}

// ---------------------------------------------------------------------------------------------
// F3  per vertex: pose-corrective offsets, blended rigid transform, translation
// ---------------------------------------------------------------------------------------------
template <bool DEEP>
__device__ __forceinline__ void vertex_posed_and_T(const Rig& rig, const float* __restrict__ ws, const float* __restrict__ v_shaped,
                                                   int v, float* vp /*3*/, float* T /*12*/)
{
    const int E = 3 * rig.V;
    float po[3] = {0.f, 0.f, 0.f};
    if (DEEP) {
        // backward: fully unrolled, the 36 x 3 strided row loads are independent and all in flight together (the rolled
        // loop paid one memory latency per pose feature: k_skin_bwd 22 -> 12 us)
#pragma unroll
        for (int p = 0; p < GAB_POSE_FEATURES; ++p) {
            const float f = ws[WS_PF + p];
            const float* row = rig.posedirs + (size_t)p * E + 3 * v;
            po[0] += f * row[0];
            po[1] += f * row[1];
            po[2] += f * row[2];
        }
    } else {
        // forward: the rolled loop measures faster (4.5 vs 10.7 us)
        for (int p = 0; p < GAB_POSE_FEATURES; ++p) {
            const float f = ws[WS_PF + p];
            const float* row = rig.posedirs + (size_t)p * E + 3 * v;
            po[0] += f * row[0];
            po[1] += f * row[1];
            po[2] += f * row[2];
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) vp[k] = po[k] + v_shaped[3 * v + k];
#pragma unroll
    for (int k = 0; k < 12; ++k) T[k] = 0.f;
#pragma unroll
    for (int j = 0; j < GAB_NUM_JOINTS; ++j) {
        const float w = rig.lbs_weights[(size_t)v * GAB_NUM_JOINTS + j];
#pragma unroll
        for (int k = 0; k < 12; ++k) T[k] += w * ws[WS_A + 12 * j + k];
    }
}

__global__ __launch_bounds__(256) void k_skin(Rig rig, const float* __restrict__ ws, const float* __restrict__ v_shaped,
                                               const float* __restrict__ translation, float* __restrict__ verts)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= rig.V) return;
    float vp[3], T[12];
    vertex_posed_and_T<false>(rig, ws, v_shaped, v, vp, T);
    for (int r = 0; r < 3; ++r)
        verts[3 * v + r] = T[4 * r] * vp[0] + T[4 * r + 1] * vp[1] + T[4 * r + 2] * vp[2] + T[4 * r + 3] + translation[r];
}

// ---------------------------------------------------------------------------------------------
// Prepared rig: everything of F1/F2 that does not depend on the per-frame parameters, folded once per (rig, shape, static_offset):
//   v_static = v_template + shapedirs[:, :n_shape] . shape (+ static_offset)          (3V)
//   J_static = J_regressor . v_static                                                  (15)
//   M        = J_regressor . shapedirs[:, n_shape:]   =>  joints = J_static + M . expr (15 x n_expr)
// With the joints a 15 x n_expr product instead of a regression over every vertex, nothing per-frame needs all the vertices any
// more, and blend shapes + chain + skinning run as ONE launch (k_flame_fused) instead of three dependent ones.
// Layout (floats): [0, 3V) v_static | [jo, jo + 16) J_static | [jo + 16, ...) M row-major, jo = 3V rounded up to 4.
// ---------------------------------------------------------------------------------------------
__host__ __device__ inline int prep_joint_offset(int V) { return (3 * V + 3) & ~3; }

__global__ __launch_bounds__(256) void k_prep_rows(Rig rig, const float* __restrict__ shape, const float* __restrict__ static_offset,
                                                    float* __restrict__ prepared)
{
    const int lane = threadIdx.x & 63;
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (e >= 3 * rig.V) return;
    const float* row = rig.shapedirs + (size_t)e * (rig.n_shape + rig.n_expr);
    float acc = 0.f;
    for (int l = lane; l < rig.n_shape; l += 64) acc += row[l] * shape[l];
    acc = wave_sum_hi(acc);
    if (lane == 63) prepared[e] = rig.v_template[e] + acc + (static_offset ? static_offset[e] : 0.f);
}
// one wave per output: o < 15: J_static[o]; else M[k][l], o - 15 = k * n_expr + l
__global__ __launch_bounds__(256) void k_prep_joints(Rig rig, float* __restrict__ prepared)
{
    const int lane = threadIdx.x & 63;
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (o >= 15 + 15 * rig.n_expr) return;
    const int NB = rig.n_shape + rig.n_expr;
    const int k = o < 15 ? o : (o - 15) / rig.n_expr, l = o < 15 ? 0 : (o - 15) % rig.n_expr;
    const int j = k / 3, c = k - 3 * j;
    float acc = 0.f;
    for (int v = lane; v < rig.V; v += 64) {
        const float w = rig.J_regressor[(size_t)j * rig.V + v];
        if (w != 0.f) acc += w * (o < 15 ? prepared[3 * v + c] : rig.shapedirs[(size_t)(3 * v + c) * NB + rig.n_shape + l]);
    }
    acc = wave_sum_hi(acc);
    const int jo = prep_joint_offset(rig.V);
    if (lane == 63) prepared[o < 15 ? jo + o : jo + 16 + (o - 15)] = acc;
}

// F1+F2+F3 in one launch.  Workgroup = GAB_FUSED_VERTS vertices.  Every workgroup works the (tiny) joint / Rodrigues / chain
// problem out for itself -- 15 x n_expr multiply-adds, five Rodrigues, one serial 5-joint chain -- while its blend-shape rows are
// in flight: a half-wave per row of the expression block (25 float4 of the 400-byte run that starts at column n_shape), all
// rows of a half-wave issued before the first is consumed.  Then 16 lanes per vertex gather the 36 pose-corrective rows and one
// lane per vertex applies the blended rigid transform.  Workgroup 0 also leaves the workspace the backward reads.
#ifndef GAB_FUSED_VERTS
#define GAB_FUSED_VERTS 8   // swept 8 / 16 / 32 (round 4, rocprofv3): 8.8 / 10.5 / 13.7 us -- 643 workgroups of 24 blend-shape rows each
#endif
template <bool FLAME_TREE>
__global__ __launch_bounds__(256) void k_flame_fused(Rig rig, const float* __restrict__ prepared, const float* __restrict__ expr,
                                                      const float* __restrict__ rotation, const float* __restrict__ neck,
                                                      const float* __restrict__ jaw, const float* __restrict__ eyes,
                                                      const float* __restrict__ translation, float* __restrict__ verts,
                                                      float* __restrict__ v_shaped, float* __restrict__ ws, const float* __restrict__ vs_pre)
{
    // vs_pre != NULL: this frame's row of the SEQUENCE table (k_blend_seq_mfma: v_shaped of every frame of an expression sequence as one
    // fp32-MFMA product) -- the expression block of the blend shapes (6 MB per frame) is then neither read nor multiplied here
    constexpr int VW = GAB_FUSED_VERTS, ROWS = 3 * VW, PER = ROWS / 8;   // rows per half-wave (8 half-waves)
    __shared__ float out[WS_DA];        // the forward half of the workspace: J, R, pose features, A
    __shared__ float vs[ROWS];          // this workgroup's rows of v_shaped
    __shared__ float po[ROWS];          // pose-corrective offsets
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int E = 3 * rig.V, NB = rig.n_shape + rig.n_expr;
    const int e0 = blockIdx.x * ROWS;
    const int jo = prep_joint_offset(rig.V);
    // ---- blend-shape rows: issue the loads first
    const int hw = 2 * wid + (lane >> 5), hl = lane & 31;   // half-wave 0..7, lane inside it
    const int nq = rig.n_expr >> 2;                          // float4 per row of the expression block
    const bool vec = (rig.n_expr & 3) == 0 && (rig.n_shape & 3) == 0 && (NB & 3) == 0 && nq <= 32;
    float acc[PER];
#pragma unroll
    for (int r = 0; r < PER; ++r) acc[r] = 0.f;
    if (vs_pre) {
        // nothing to multiply
    } else if (vec) {
        float4 rw[PER];
        const float4 b = hl < nq ? reinterpret_cast<const float4*>(expr)[hl] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int r = 0; r < PER; ++r) {
            const int e = e0 + hw * PER + r;
            rw[r] = (e < E && hl < nq) ? reinterpret_cast<const float4*>(rig.shapedirs + (size_t)e * NB + rig.n_shape)[hl]
                                       : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int r = 0; r < PER; ++r) acc[r] = rw[r].x * b.x + rw[r].y * b.y + rw[r].z * b.z + rw[r].w * b.w;
    } else {
#pragma unroll
        for (int r = 0; r < PER; ++r) {
            const int e = e0 + hw * PER + r;
            if (e < E)
                for (int l = hl; l < rig.n_expr; l += 32) acc[r] += rig.shapedirs[(size_t)e * NB + rig.n_shape + l] * expr[l];
        }
    }
    // ---- joints = J_static + M . expr: 16 lanes per output (240 threads), Rodrigues on five more lanes
    if (tid < WS_DA) out[tid] = 0.f;
    __syncthreads();
    if (tid < 240) {
        const int k = tid >> 4, q = tid & 15;
        const float* mrow = prepared + jo + 16 + (size_t)k * rig.n_expr;
        float a = 0.f;
        for (int l = q; l < rig.n_expr; l += 16) a += mrow[l] * expr[l];
        a += dpp_f<0xB1, 0xf>(a);
        a += dpp_f<0x4E, 0xf>(a);
        a += dpp_f<0x141, 0xf>(a);
        a += dpp_f<0x140, 0xf>(a);   // every lane of the row of 16 holds the sum
        if (q == 0) out[WS_J + k] = prepared[jo + k] + a;
    } else if (tid < 240 + GAB_NUM_JOINTS) {
        const int j = tid - 240;
        const float* pj = j == 0 ? rotation : j == 1 ? neck : j == 2 ? jaw : eyes + 3 * (j - 3);
        const float pose[3] = {pj[0], pj[1], pj[2]};
        float R[9];
        rodrigues(pose, R);
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            out[WS_R + 9 * j + k] = R[k];
            if (j >= 1) out[WS_PF + 9 * (j - 1) + k] = R[k] - ((k % 4) == 0 ? 1.f : 0.f);
        }
    }
    // ---- finish the rows: 32-lane sums (rows of 16 by DPP, row 0 -> 1 and 2 -> 3 by row_bcast:15)
#pragma unroll
    for (int r = 0; r < PER; ++r) {
        float a = acc[r];
        a += dpp_f<0xB1, 0xf>(a);
        a += dpp_f<0x4E, 0xf>(a);
        a += dpp_f<0x141, 0xf>(a);
        a += dpp_f<0x140, 0xf>(a);
        a += dpp_f<0x142, 0xa>(a);   // lanes 16..31 / 48..63 now hold their half-wave's sum
        const int e = e0 + hw * PER + r;
        if (hl == 31 && e < E) {
            const float x = vs_pre ? vs_pre[e] : prepared[e] + a;
            vs[hw * PER + r] = x;
            v_shaped[e] = x;
        }
    }
    __syncthreads();
    if (tid == 0) {
        float J[15], R[45], Rg[45], tg[15];
#pragma unroll
        for (int k = 0; k < 15; ++k) J[k] = out[WS_J + k];
#pragma unroll
        for (int k = 0; k < 45; ++k) R[k] = out[WS_R + k];
        chain_forward<FLAME_TREE>(rig.parents, R, J, Rg, tg);
#pragma unroll
        for (int j = 0; j < GAB_NUM_JOINTS; ++j)
#pragma unroll
            for (int r = 0; r < 3; ++r) {
#pragma unroll
                for (int c = 0; c < 3; ++c) out[WS_A + 12 * j + 4 * r + c] = Rg[9 * j + 3 * r + c];
                out[WS_A + 12 * j + 4 * r + 3] = tg[3 * j + r] - (Rg[9 * j + 3 * r] * J[3 * j] + Rg[9 * j + 3 * r + 1] * J[3 * j + 1] +
                                                                  Rg[9 * j + 3 * r + 2] * J[3 * j + 2]);
            }
    } else if (tid >= 64) {
        // meanwhile: pose-corrective offsets, 16 lanes per vertex (lanes 64..255 = 12 vertices per pass)
        for (int vl = (tid - 64) >> 4; vl < VW; vl += 12) {
            const int v = blockIdx.x * VW + vl, q = tid & 15;
            float o[3] = {0.f, 0.f, 0.f};
            if (v < rig.V)
                for (int p = q; p < GAB_POSE_FEATURES; p += 16) {
                    const float f = out[WS_PF + p];
                    const float* row = rig.posedirs + (size_t)p * E + 3 * v;
                    o[0] += f * row[0]; o[1] += f * row[1]; o[2] += f * row[2];
                }
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                float a = o[k];
                a += dpp_f<0xB1, 0xf>(a);
                a += dpp_f<0x4E, 0xf>(a);
                a += dpp_f<0x141, 0xf>(a);
                a += dpp_f<0x140, 0xf>(a);
                if (q == 0) po[3 * vl + k] = a;
            }
        }
    }
    __syncthreads();
    if (blockIdx.x == 0) {   // the workspace the backward reads (and its accumulators, handed over zeroed)
        if (tid < WS_DA) ws[tid] = out[tid];
        for (int k = WS_DA + tid; k < GAB_FLAME_WS_FLOATS; k += 256) ws[k] = 0.f;
    }
    if (tid < VW) {
        const int v = blockIdx.x * VW + tid;
        if (v < rig.V) {
            float T[12];
#pragma unroll
            for (int k = 0; k < 12; ++k) T[k] = 0.f;
#pragma unroll
            for (int j = 0; j < GAB_NUM_JOINTS; ++j) {
                const float w = rig.lbs_weights[(size_t)v * GAB_NUM_JOINTS + j];
#pragma unroll
                for (int k = 0; k < 12; ++k) T[k] += w * out[WS_A + 12 * j + k];
            }
            const float vp[3] = {po[3 * tid] + vs[3 * tid], po[3 * tid + 1] + vs[3 * tid + 1], po[3 * tid + 2] + vs[3 * tid + 2]};
#pragma unroll
            for (int r = 0; r < 3; ++r)
                verts[3 * v + r] = T[4 * r] * vp[0] + T[4 * r + 1] * vp[1] + T[4 * r + 2] * vp[2] + T[4 * r + 3] + translation[r];
        }
    }
}

struct ZeroSpec { float* p[8]; int n[8]; int count; };   // up to 8 buffers a kernel zero-fills on the side

// ---------------------------------------------------------------------------------------------
// B1  skinning backward: dL/dv_posed per vertex -> scratch; block-reduced sums for dA (60),
//     d translation (3), d pose_feature (36) -> ws
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_skin_bwd(Rig rig, float* __restrict__ ws, const float* __restrict__ v_shaped,
                                                   const float* __restrict__ dL_dverts, float* __restrict__ g_vs /*(V,3)*/, ZeroSpec zero)
{
    __shared__ float red[4][100];   // per-wave partials of the 99 sums (no LDS atomics)
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int v = blockIdx.x * blockDim.x + tid;
    // first kernel of the FLAME backward: zero-fill the caller's gradient tables on the side (rows of them are written by the
    // kernels that follow), instead of a launch of their own
    for (int b = 0; b < zero.count; ++b)
        for (int i = v; i < zero.n[b]; i += (int)(gridDim.x * blockDim.x)) zero.p[b][i] = 0.f;
    const bool ok = v < rig.V;
    const int E = 3 * rig.V;
    float g[3] = {0.f, 0.f, 0.f}, vp[3] = {0.f, 0.f, 0.f}, T[12], gvp[3] = {0.f, 0.f, 0.f};
    float w[GAB_NUM_JOINTS] = {0.f, 0.f, 0.f, 0.f, 0.f};
    float pf[GAB_POSE_FEATURES];
#pragma unroll
    for (int p = 0; p < GAB_POSE_FEATURES; ++p) pf[p] = 0.f;
    if (ok) {
        vertex_posed_and_T<true>(rig, ws, v_shaped, v, vp, T);
#pragma unroll
        for (int k = 0; k < 3; ++k) g[k] = dL_dverts[3 * v + k];
#pragma unroll
        for (int c = 0; c < 3; ++c) gvp[c] = T[c] * g[0] + T[4 + c] * g[1] + T[8 + c] * g[2];
#pragma unroll
        for (int k = 0; k < 3; ++k) g_vs[3 * v + k] = gvp[k];
#pragma unroll
        for (int j = 0; j < GAB_NUM_JOINTS; ++j) w[j] = rig.lbs_weights[(size_t)v * GAB_NUM_JOINTS + j];
#pragma unroll
        for (int p = 0; p < GAB_POSE_FEATURES; ++p) {   // rows are L1/L2-resident from vertex_posed_and_T; all loads in flight
            const float* row = rig.posedirs + (size_t)p * E + 3 * v;
            pf[p] = row[0] * gvp[0] + row[1] * gvp[1] + row[2] * gvp[2];
        }
    }
    const float vph[4] = {vp[0], vp[1], vp[2], ok ? 1.f : 0.f};
    // dA[j][r][c] += w_j g_r vph_c
#pragma unroll
    for (int j = 0; j < GAB_NUM_JOINTS; ++j)
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float s = wave_sum_hi(w[j] * g[r] * vph[c]);
                if (lane == 63) red[wid][12 * j + 4 * r + c] = s;
            }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const float s = wave_sum_hi(g[r]);
        if (lane == 63) red[wid][60 + r] = s;
    }
#pragma unroll
    for (int p = 0; p < GAB_POSE_FEATURES; ++p) {
        const float s = wave_sum_hi(pf[p]);
        if (lane == 63) red[wid][63 + p] = s;
    }
    __syncthreads();
    if (tid < 99) {
        const float s = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
        float* dst = tid < 60 ? &ws[WS_DA + tid] : (tid < 63 ? &ws[WS_DT + tid - 60] : &ws[WS_DPF + tid - 63]);
        unsafeAtomicAdd(dst, s);
    }
}

// ---------------------------------------------------------------------------------------------
// B2  chain backward (one thread; 5 joints): dA, d pose_feature -> d pose (15), dJ (15)
// ---------------------------------------------------------------------------------------------
// NT threads run it (one workgroup).  Mmat == nullptr: the classic three-kernel backward -- d_expr / d_shape are zeroed here and
// k_blend_bwd (next launch) adds J_regressor^T dJ through the vertices.  Mmat != nullptr (prepared rig): dJ reaches d_expr as
// M^T dJ right here, d_expr was zeroed by k_skin_bwd, and the blend workgroups of the same launch never look at dJ.
template <bool FLAME_TREE, int NT>
__device__ __forceinline__ void chain_bwd_body(Rig rig, float* __restrict__ ws, const float* __restrict__ rotation, const float* __restrict__ neck,
                            const float* __restrict__ jaw, const float* __restrict__ eyes, float* __restrict__ d_rotation,
                            float* __restrict__ d_neck, float* __restrict__ d_jaw, float* __restrict__ d_eyes,
                            float* __restrict__ d_translation, float* __restrict__ d_expr, float* __restrict__ d_shape,
                            const float* __restrict__ Mmat)
{
    // stage the whole workspace through LDS with one coalesced pass, then a single lane runs the
    // (inherently serial, 5-joint) chain out of LDS instead of ~250 dependent global loads
    __shared__ float sw[GAB_FLAME_WS_FLOATS];
    __shared__ float sdJ[16];
    for (int kk = threadIdx.x; kk < GAB_FLAME_WS_FLOATS; kk += NT) sw[kk] = ws[kk];
    __syncthreads();
    // consumed: leave the accumulators zeroed for the next backward, and zero the targets k_blend_bwd adds into
    for (int kk = WS_DA + (int)threadIdx.x; kk < WS_DJ; kk += NT) ws[kk] = 0.f;
    if (!Mmat) {
        for (int kk = threadIdx.x; kk < rig.n_expr; kk += NT) d_expr[kk] = 0.f;
        if (d_shape)
            for (int kk = threadIdx.x; kk < rig.n_shape; kk += NT) d_shape[kk] = 0.f;
    }
    __shared__ float sdR[45];
    if (threadIdx.x == 0) {
    float J[15], R[45], Rg[45], tg[15];
#pragma unroll
    for (int k = 0; k < 15; ++k) J[k] = sw[WS_J + k];
#pragma unroll
    for (int k = 0; k < 45; ++k) R[k] = sw[WS_R + k];
    chain_forward<FLAME_TREE>(rig.parents, R, J, Rg, tg);
    float dRg[45], dtg[15], dJ[15], dR[45];
#pragma unroll
    for (int k = 0; k < 45; ++k) { dRg[k] = 0.f; dR[k] = 0.f; }
#pragma unroll
    for (int k = 0; k < 15; ++k) { dtg[k] = 0.f; dJ[k] = 0.f; }
    // A_i = [Rg_i | tg_i - Rg_i J_i]
#pragma unroll
    for (int i = 0; i < GAB_NUM_JOINTS; ++i) {
        const float* dA = sw + WS_DA + 12 * i;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const float dAt = dA[4 * r + 3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                dRg[9 * i + 3 * r + c] += dA[4 * r + c] - dAt * J[3 * i + c];
                dJ[3 * i + c] -= Rg[9 * i + 3 * r + c] * dAt;
            }
            dtg[3 * i + r] += dAt;
        }
    }
#pragma unroll
    for (int i = GAB_NUM_JOINTS - 1; i >= 1; --i) {
        const int p = parent_of<FLAME_TREE>(rig.parents, i);
        const float rel[3] = {J[3 * i] - J[3 * p], J[3 * i + 1] - J[3 * p + 1], J[3 * i + 2] - J[3 * p + 2]};
        float drel[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 3; ++r) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                dRg[9 * p + 3 * r + c] += dtg[3 * i + r] * rel[c];
                drel[c] += Rg[9 * p + 3 * r + c] * dtg[3 * i + r];
            }
            dtg[3 * p + r] += dtg[3 * i + r];
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) { dJ[3 * i + c] += drel[c]; dJ[3 * p + c] -= drel[c]; }
        // Rg_i = Rg_p R_i
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float a = 0.f, b = 0.f;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    a += dRg[9 * i + 3 * r + k] * R[9 * i + 3 * c + k];      // dRg_i R_i^T
                    b += Rg[9 * p + 3 * k + r] * dRg[9 * i + 3 * k + c];     // Rg_p^T dRg_i
                }
                dRg[9 * p + 3 * r + c] += a;
                dR[9 * i + 3 * r + c] += b;
            }
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) dR[k] += dRg[k];
#pragma unroll
    for (int c = 0; c < 3; ++c) dJ[c] += dtg[c];
#pragma unroll
    for (int j = 1; j < GAB_NUM_JOINTS; ++j)
#pragma unroll
        for (int k = 0; k < 9; ++k) dR[9 * j + k] += sw[WS_DPF + 9 * (j - 1) + k];
#pragma unroll
    for (int k = 0; k < 45; ++k) sdR[k] = dR[k];
#pragma unroll
    for (int k = 0; k < 15; ++k) { ws[WS_DJ + k] = dJ[k]; sdJ[k] = dJ[k]; }
    }
    __syncthreads();
    if (Mmat)   // d_expr += M^T dJ (the joints' dependence on the expression coefficients)
        for (int l = threadIdx.x; l < rig.n_expr; l += NT) {
            float a = 0.f;
#pragma unroll
            for (int k = 0; k < 15; ++k) a += Mmat[(size_t)k * rig.n_expr + l] * sdJ[k];
            unsafeAtomicAdd(&d_expr[l], a);
        }
    // Rodrigues backward: one lane per joint
    const int j = threadIdx.x;
    if (j < GAB_NUM_JOINTS) {
        const float* pj = j == 0 ? rotation : j == 1 ? neck : j == 2 ? jaw : eyes + 3 * (j - 3);
        float* dj = j == 0 ? d_rotation : j == 1 ? d_neck : j == 2 ? d_jaw : d_eyes + 3 * (j - 3);
        const float pose[3] = {pj[0], pj[1], pj[2]};
        float dRj[9], dpose[3];
#pragma unroll
        for (int k = 0; k < 9; ++k) dRj[k] = sdR[9 * j + k];
        rodrigues_bwd(pose, dRj, dpose);
        dj[0] = dpose[0]; dj[1] = dpose[1]; dj[2] = dpose[2];
    } else if (j >= 8 && j < 11) {
        d_translation[j - 8] = sw[WS_DT + j - 8];
    }
}
template <bool FLAME_TREE>
__global__ __launch_bounds__(64) void k_chain_bwd(Rig rig, float* __restrict__ ws, const float* __restrict__ rotation, const float* __restrict__ neck,
                            const float* __restrict__ jaw, const float* __restrict__ eyes, float* __restrict__ d_rotation,
                            float* __restrict__ d_neck, float* __restrict__ d_jaw, float* __restrict__ d_eyes,
                            float* __restrict__ d_translation, float* __restrict__ d_expr, float* __restrict__ d_shape)
{
    chain_bwd_body<FLAME_TREE, 64>(rig, ws, rotation, neck, jaw, eyes, d_rotation, d_neck, d_jaw, d_eyes, d_translation, d_expr, d_shape, nullptr);
}

// ---------------------------------------------------------------------------------------------
// B3  blend backward: g_total = dL/dv_posed + J_regressor^T dJ (+ external dL/dv_shaped);
//     d static_offset = g_total; d betas = shapedirs^T g_total (128 rows per workgroup)
// ---------------------------------------------------------------------------------------------
#define GAB_BLEND_BWD_ROWS 128   // swept 64 / 128 / 256: 11.0 / 8.7 / 9.7 us (fewer, less contended atomics vs parallelism)
// ws == nullptr: no J_regressor^T dJ term (prepared rig: it went into d_expr as M^T dJ)
__device__ __forceinline__ void blend_bwd_body(Rig rig, const float* __restrict__ ws, const float* __restrict__ g_vs,
                                               const float* __restrict__ dL_dv_shaped, float* __restrict__ d_static_offset,
                                               float* __restrict__ d_shape, float* __restrict__ d_expr, int block)
{
    __shared__ float g[GAB_BLEND_BWD_ROWS];
    const int tid = threadIdx.x;
    const int E = 3 * rig.V;
    const int e0 = block * GAB_BLEND_BWD_ROWS;
    if (tid < GAB_BLEND_BWD_ROWS) {
        const int e = e0 + tid;
        float x = 0.f;
        if (e < E) {
            const int v = e / 3, k = e - 3 * v;
            x = g_vs[e];
            if (ws)
                for (int j = 0; j < GAB_NUM_JOINTS; ++j) x += rig.J_regressor[(size_t)j * rig.V + v] * ws[WS_DJ + 3 * j + k];
            if (dL_dv_shaped) x += dL_dv_shaped[e];
            if (d_static_offset) d_static_offset[e] = x;
        }
        g[tid] = x;
    }
    __syncthreads();
    const int NB = rig.n_shape + rig.n_expr;
    const int rows = min(GAB_BLEND_BWD_ROWS, E - e0);
    // columns to produce: all betas, or only the expression block when the (constant) shape is not optimised -- the
    // reference never optimises it, so 3/4 of the stream is skipped.  128 column lanes x 2 row halves: every lane is
    // busy, a lane's 64 row loads are independent (unrolled by 8) and consecutive lanes read consecutive columns.
    const int c0 = d_shape ? 0 : rig.n_shape, nC = NB - c0;
    __shared__ float part[128];
    const int cl = tid & 127, half = tid >> 7;
    const int r0 = half * (GAB_BLEND_BWD_ROWS / 2), r1 = min(rows, r0 + GAB_BLEND_BWD_ROWS / 2);
    for (int cb = 0; cb < nC; cb += 128) {
        const int c = cb + cl;
        float acc = 0.f;
        if (c < nC) {
            const float* col = rig.shapedirs + (size_t)e0 * NB + c0 + c;
            int r = r0;
            for (; r + 8 <= r1; r += 8) {
                float x[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) x[k] = col[(size_t)(r + k) * NB];
#pragma unroll
                for (int k = 0; k < 8; ++k) acc += x[k] * g[r + k];
            }
            for (; r < r1; ++r) acc += col[(size_t)r * NB] * g[r];
        }
        if (half == 1) part[cl] = acc;
        __syncthreads();
        if (half == 0 && c < nC) {
            const int l = c0 + c;
            float* dst = l < rig.n_shape ? d_shape + l : d_expr + (l - rig.n_shape);
            unsafeAtomicAdd(dst, acc + part[cl]);
        }
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void k_blend_bwd(Rig rig, const float* __restrict__ ws, const float* __restrict__ g_vs,
                                                    const float* __restrict__ dL_dv_shaped, float* __restrict__ d_static_offset,
                                                    float* __restrict__ d_shape, float* __restrict__ d_expr)
{
    blend_bwd_body(rig, ws, g_vs, dL_dv_shaped, d_static_offset, d_shape, d_expr, (int)blockIdx.x);
}
// prepared rig: B2 and B3 side by side in one launch -- workgroup 0 runs the chain (and adds M^T dJ), the others the
// expression block of shapedirs^T g; neither waits for the other
template <bool FLAME_TREE>
__global__ __launch_bounds__(256) void k_chain_blend_bwd(Rig rig, float* __restrict__ ws, const float* __restrict__ g_vs, const float* __restrict__ Mmat,
                                                          const float* __restrict__ rotation, const float* __restrict__ neck,
                                                          const float* __restrict__ jaw, const float* __restrict__ eyes,
                                                          float* __restrict__ d_rotation, float* __restrict__ d_neck, float* __restrict__ d_jaw,
                                                          float* __restrict__ d_eyes, float* __restrict__ d_translation, float* __restrict__ d_expr)
{
    if (blockIdx.x == 0)
        chain_bwd_body<FLAME_TREE, 256>(rig, ws, rotation, neck, jaw, eyes, d_rotation, d_neck, d_jaw, d_eyes, d_translation, d_expr, nullptr, Mmat);
    else
        blend_bwd_body(rig, nullptr, g_vs, nullptr, nullptr, nullptr, d_expr, (int)blockIdx.x - 1);
}

// ---------------------------------------------------------------------------------------------
// per-face frames
// ---------------------------------------------------------------------------------------------
struct Vec3 { float x, y, z; };
__device__ __forceinline__ Vec3 operator-(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ Vec3 operator+(Vec3 a, Vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ Vec3 operator*(Vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ float dot(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ Vec3 cross(Vec3 a, Vec3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ Vec3 ld3(const float* p, long long i) { return {p[3 * i], p[3 * i + 1], p[3 * i + 2]}; }

__device__ __forceinline__ long long index_at(const void* idx, int is64, long long i)
{
    return is64 ? ((const long long*)idx)[i] : (long long)((const int*)idx)[i];
}

struct Frame {
    Vec3 e1, e2, a0, a1, a2, n, m;
    float l1, ln, lm, d;
    bool c1, cn, cm;  // clamp active
};
__device__ __forceinline__ Frame make_frame(Vec3 v0, Vec3 v1, Vec3 v2)
{
    Frame f;
    const float eps = 1e-20f;
    f.e1 = v1 - v0;
    f.e2 = v2 - v0;
    float q = dot(f.e1, f.e1);
    f.c1 = q < eps; f.l1 = sqrtf(f.c1 ? eps : q);
    f.a0 = f.e1 * (1.f / f.l1);
    f.n = cross(f.a0, f.e2);
    q = dot(f.n, f.n);
    f.cn = q < eps; f.ln = sqrtf(f.cn ? eps : q);
    f.a1 = f.n * (1.f / f.ln);
    f.m = cross(f.a1, f.a0);
    q = dot(f.m, f.m);
    f.cm = q < eps; f.lm = sqrtf(f.cm ? eps : q);
    f.a2 = f.m * (-1.f / f.lm);
    f.d = dot(f.a2, f.e2);
    return f;
}

// R (row-major, columns a0 a1 a2) -> unnormalised XYZW quaternion + branch (SciPy/roma algorithm)
__device__ __forceinline__ int quat_raw(const float* R, float* q)
{
    const float tr = R[0] + R[4] + R[8];
    const float dec[4] = {R[0], R[4], R[8], tr};
    int c = 0;
    for (int k = 1; k < 4; ++k) if (dec[k] > dec[c]) c = k;
    if (c < 3) {
        const int i = c, j = (i + 1) % 3, k = (j + 1) % 3;
        q[i] = 1.f - tr + 2.f * R[4 * i];
        q[j] = R[3 * j + i] + R[3 * i + j];
        q[k] = R[3 * k + i] + R[3 * i + k];
        q[3] = R[3 * k + j] - R[3 * j + k];
    } else {
        q[0] = R[7] - R[5];
        q[1] = R[2] - R[6];
        q[2] = R[3] - R[1];
        q[3] = 1.f + tr;
    }
    return c;
}

__global__ __launch_bounds__(256) void k_face_frames(int F, const float* __restrict__ verts, const void* __restrict__ faces, int is64,
                                                      float* __restrict__ center, float* __restrict__ Rm, float* __restrict__ scaling,
                                                      float* __restrict__ quat, float* __restrict__ zero_fill, int zero_count)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (zero_fill)   // the backward's accumulation target, prepared here so that no memset precedes its atomics
        for (int k = f; k < zero_count; k += (int)(gridDim.x * blockDim.x)) zero_fill[k] = 0.f;
    if (f >= F) return;
    const Vec3 v0 = ld3(verts, index_at(faces, is64, 3ll * f)), v1 = ld3(verts, index_at(faces, is64, 3ll * f + 1)),
               v2 = ld3(verts, index_at(faces, is64, 3ll * f + 2));
    const Frame fr = make_frame(v0, v1, v2);
    const Vec3 c = (v0 + v1 + v2) * (1.f / 3.f);
    center[3 * f] = c.x; center[3 * f + 1] = c.y; center[3 * f + 2] = c.z;
    const float R[9] = {fr.a0.x, fr.a1.x, fr.a2.x, fr.a0.y, fr.a1.y, fr.a2.y, fr.a0.z, fr.a1.z, fr.a2.z};
    for (int k = 0; k < 9; ++k) Rm[9 * f + k] = R[k];
    scaling[f] = (fr.l1 + fabsf(fr.d)) * 0.5f;
    float q[4];
    quat_raw(R, q);
    const float inv = 1.f / sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    quat[4 * f] = q[3] * inv;       // WXYZ
    quat[4 * f + 1] = q[0] * inv;
    quat[4 * f + 2] = q[1] * inv;
    quat[4 * f + 3] = q[2] * inv;
}

__device__ __forceinline__ Vec3 unit_bwd(Vec3 g_unit, Vec3 unit, float len, bool clamped)
{
    // y = x / len(x): dL/dx = (g - y (y.g)) / len; with the clamp active len is a constant
    if (clamped) return g_unit * (1.f / len);
    return (g_unit - unit * dot(unit, g_unit)) * (1.f / len);
}

// gradients of one face's frame outputs -> gradients of its three corners (g0, g1, g2), the corners' vertices given
__device__ __forceinline__ void face_frame_bwd(int f, Vec3 v0, Vec3 v1, Vec3 v2, const float* __restrict__ d_center, const float* __restrict__ d_R,
                                               const float* __restrict__ d_scaling, const float* __restrict__ d_quat, Vec3& g0, Vec3& g1, Vec3& g2)
{
    const Frame fr = make_frame(v0, v1, v2);
    const float R[9] = {fr.a0.x, fr.a1.x, fr.a2.x, fr.a0.y, fr.a1.y, fr.a2.y, fr.a0.z, fr.a1.z, fr.a2.z};
    float gR[9];
    for (int k = 0; k < 9; ++k) gR[k] = d_R ? d_R[9 * f + k] : 0.f;
    if (d_quat) {
        float q[4];
        const int c = quat_raw(R, q);
        const float len = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
        const float qn[4] = {q[0] / len, q[1] / len, q[2] / len, q[3] / len};
        const float g[4] = {d_quat[4 * f + 1], d_quat[4 * f + 2], d_quat[4 * f + 3], d_quat[4 * f]};   // WXYZ -> XYZW
        const float gd = qn[0] * g[0] + qn[1] * g[1] + qn[2] * g[2] + qn[3] * g[3];
        float gq[4];
        for (int k = 0; k < 4; ++k) gq[k] = (g[k] - qn[k] * gd) / len;
        if (c < 3) {
            const int i = c, j = (i + 1) % 3, k = (j + 1) % 3;
            gR[0] -= gq[i]; gR[4] -= gq[i]; gR[8] -= gq[i];
            gR[4 * i] += 2.f * gq[i];
            gR[3 * j + i] += gq[j]; gR[3 * i + j] += gq[j];
            gR[3 * k + i] += gq[k]; gR[3 * i + k] += gq[k];
            gR[3 * k + j] += gq[3]; gR[3 * j + k] -= gq[3];
        } else {
            gR[7] += gq[0]; gR[5] -= gq[0];
            gR[2] += gq[1]; gR[6] -= gq[1];
            gR[3] += gq[2]; gR[1] -= gq[2];
            gR[0] += gq[3]; gR[4] += gq[3]; gR[8] += gq[3];
        }
    }
    Vec3 g_a0 = {gR[0], gR[3], gR[6]}, g_a1 = {gR[1], gR[4], gR[7]}, g_a2 = {gR[2], gR[5], gR[8]};
    Vec3 g_e2 = {0.f, 0.f, 0.f};
    float g_l1 = 0.f;
    if (d_scaling) {
        const float gs = d_scaling[f] * 0.5f;
        g_l1 = gs;
        const float sg = fr.d > 0.f ? 1.f : (fr.d < 0.f ? -1.f : 0.f);
        g_a2 = g_a2 + fr.e2 * (gs * sg);
        g_e2 = g_e2 + fr.a2 * (gs * sg);
    }
    // a2 = -(m / |m|)
    const Vec3 u = fr.a2 * -1.f;
    const Vec3 g_m = unit_bwd(g_a2 * -1.f, u, fr.lm, fr.cm);
    // m = a1 x a0
    g_a1 = g_a1 + cross(fr.a0, g_m);
    g_a0 = g_a0 + cross(g_m, fr.a1);
    // a1 = n / |n|,  n = a0 x e2
    const Vec3 g_n = unit_bwd(g_a1, fr.a1, fr.ln, fr.cn);
    g_a0 = g_a0 + cross(fr.e2, g_n);
    g_e2 = g_e2 + cross(g_n, fr.a0);
    // a0 = e1 / |e1|, l1 = |e1|
    Vec3 g_e1 = unit_bwd(g_a0, fr.a0, fr.l1, fr.c1);
    if (!fr.c1) g_e1 = g_e1 + fr.a0 * g_l1;
    Vec3 gc = {0.f, 0.f, 0.f};
    if (d_center) gc = Vec3{d_center[3 * f], d_center[3 * f + 1], d_center[3 * f + 2]} * (1.f / 3.f);
    g0 = gc - g_e1 - g_e2; g1 = gc + g_e1; g2 = gc + g_e2;
}

__global__ __launch_bounds__(256) void k_face_frames_bwd(int F, const float* __restrict__ verts, const void* __restrict__ faces, int is64,
                                                          const float* __restrict__ d_center, const float* __restrict__ d_R,
                                                          const float* __restrict__ d_scaling, const float* __restrict__ d_quat,
                                                          float* __restrict__ d_verts)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    const long long i0 = index_at(faces, is64, 3ll * f), i1 = index_at(faces, is64, 3ll * f + 1), i2 = index_at(faces, is64, 3ll * f + 2);
    Vec3 g0, g1, g2;
    face_frame_bwd(f, ld3(verts, i0), ld3(verts, i1), ld3(verts, i2), d_center, d_R, d_scaling, d_quat, g0, g1, g2);
    unsafeAtomicAdd(&d_verts[3 * i0], g0.x); unsafeAtomicAdd(&d_verts[3 * i0 + 1], g0.y); unsafeAtomicAdd(&d_verts[3 * i0 + 2], g0.z);
    unsafeAtomicAdd(&d_verts[3 * i1], g1.x); unsafeAtomicAdd(&d_verts[3 * i1 + 1], g1.y); unsafeAtomicAdd(&d_verts[3 * i1 + 2], g1.z);
    unsafeAtomicAdd(&d_verts[3 * i2], g2.x); unsafeAtomicAdd(&d_verts[3 * i2 + 1], g2.y); unsafeAtomicAdd(&d_verts[3 * i2 + 2], g2.z);
}

// ---------------------------------------------------------------------------------------------
// k_gather_skin_bwd: k_face_frames_bwd + k_skin_bwd in one launch, for select_mesh_by_timestep + update_mesh_properties as one
// autograd node.  The pair of launches it replaces pays two launch floors (4 us each), 91 k float atomics into the vertices and a
// zero-fill of their target.  Here a workgroup owns GAB_MESH_VPB consecutive vertices and GATHERS instead: the (face, corner) pairs
// of its vertices come from a vertex -> corner table of the (static) topology, 16 bytes per pair = (4 f + c, i0, i1, i2), so that a
// pair costs two dependent loads (entry, then vertices and gradients); one thread per pair re-derives the face's frame backward (each
// face three times over the grid: ~250 flops) and adds its corner's share into LDS -- no global atomics, no d_verts buffer.
// Then the skinning backward of k_skin_bwd for those vertices, the four waves sharing the 99 block sums.
// (One launch for the WHOLE mesh backward was built and measured first: the grid-wide hand-over it needs -- fence, ticket, fence, a
// last workgroup summing 81 rows of partials -- cost 6 + 18 us against the 4 us launch floor it saved; see DESIGN.md.)
// ---------------------------------------------------------------------------------------------
#ifndef GAB_MESH_VPB
#define GAB_MESH_VPB 32
#endif
struct GatherSkinArgs {
    const float* v_shaped; float* ws;
    const float* verts;
    const int* vf_begin; const int4* vf_list;                       // vertex -> (4 f + c, i0, i1, i2), rows sorted by vertex
    const float* d_center; const float* d_R; const float* d_scaling; const float* d_quat;
    const float* g_verts;                                           // optional external dL/d(posed vertices), added in
    float* g_vs;                                                    // (V,3) out, for k_chain_blend_bwd
};
__global__ __launch_bounds__(256) void k_gather_skin_bwd(Rig rig, GatherSkinArgs a, ZeroSpec zero)
{
    constexpr int VPB = GAB_MESH_VPB, ROWS = 3 * VPB;
    static_assert(VPB == 16 || VPB == 32 || VPB == 64, "a wave holds the workgroup's vertices");
    __shared__ float gv[ROWS];          // dL/d(posed vertex) of this workgroup's vertices
    __shared__ float red[100];          // the 99 block sums
    // Round 6: the kernel is a chain of memory round trips behind a launch floor (corner entry -> vertices and face gradients -> LDS; then, for the
    // skinning backward, the pose-corrective rows, the shaped vertices and the weights): the second chain does not depend on the first, so its loads
    // are ISSUED before the gather and only waited for behind it -- the workgroup's block of the pose-corrective table (36 rows x 3 VPB floats) once,
    // cooperatively, into LDS (every wave used to fetch it for itself, twice), the per-vertex values into registers.
    constexpr int PDN = GAB_POSE_FEATURES * ROWS, PDK = (PDN + 255) / 256;
    __shared__ float pd[GAB_POSE_FEATURES][ROWS];
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int gid = blockIdx.x * 256 + tid;
    for (int b = 0; b < zero.count; ++b)
        for (int i = gid; i < zero.n[b]; i += (int)(gridDim.x * 256)) zero.p[b][i] = 0.f;
    const int vbase = blockIdx.x * VPB, vend = min(vbase + VPB, rig.V);
    const int E = 3 * rig.V;
    const int p0 = a.vf_begin[vbase], p1 = a.vf_begin[vend];
    // ---- loads of the skinning half, in flight across the gather
    float pdr[PDK];
#pragma unroll
    for (int i = 0; i < PDK; ++i) {
        const int k = tid + 256 * i, p = k / ROWS, c = k - p * ROWS;
        pdr[i] = (k < PDN && 3 * vbase + c < E) ? rig.posedirs[(size_t)p * E + 3 * vbase + c] : 0.f;
    }
    const int v = vbase + lane;
    const bool ok = lane < VPB && v < rig.V;
    float vs[3] = {0.f, 0.f, 0.f}, gx[3] = {0.f, 0.f, 0.f};
    float w[GAB_NUM_JOINTS] = {0.f, 0.f, 0.f, 0.f, 0.f};
    if (ok) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { vs[k] = a.v_shaped[3 * v + k]; gx[k] = a.g_verts ? a.g_verts[3 * v + k] : 0.f; }
#pragma unroll
        for (int j = 0; j < GAB_NUM_JOINTS; ++j) w[j] = rig.lbs_weights[(size_t)v * GAB_NUM_JOINTS + j];
    }
    for (int k = tid; k < ROWS; k += 256) gv[k] = 0.f;
    __syncthreads();
    // ---- gather the corners
    // d(translation) is the plain sum of the vertex gradients, and of those the frames' orientation / scale parts -- translation invariant,
    // hundreds of times larger than the rest -- cancel exactly in mathematics but not in fp32 (a run-to-run spread of 1.6e-4 of the row's
    // max in arrival-order atomics, tools/mesh_bwd_repeat.py).  What survives the sum is each face centre's gradient, a third per corner:
    // that is summed here directly (tc), and the vertex sums below only add the external vertex gradient to it.
    float tc[3] = {0.f, 0.f, 0.f};
    for (int p = p0 + tid; p < p1; p += 256) {
        const int4 e = a.vf_list[p];
        const int f = e.x >> 2, c = e.x & 3;
        Vec3 g0, g1, g2;
        face_frame_bwd(f, ld3(a.verts, e.y), ld3(a.verts, e.z), ld3(a.verts, e.w), a.d_center, a.d_R, a.d_scaling, a.d_quat, g0, g1, g2);
        const Vec3 g = c == 0 ? g0 : (c == 1 ? g1 : g2);
        const int vl = (c == 0 ? e.y : (c == 1 ? e.z : e.w)) - vbase;
        atomicAdd(&gv[3 * vl], g.x); atomicAdd(&gv[3 * vl + 1], g.y); atomicAdd(&gv[3 * vl + 2], g.z);
        if (a.d_center) {
#pragma unroll
            for (int r = 0; r < 3; ++r) tc[r] += a.d_center[3 * f + r] * (1.0f / 3.0f);
        }
    }
    __shared__ float tcs[4][3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const float s = wave_sum_hi(tc[r]);
        if (lane == 63) tcs[wid][r] = s;
    }
#pragma unroll
    for (int i = 0; i < PDK; ++i) {
        const int k = tid + 256 * i;
        if (k < PDN) (&pd[0][0])[k] = pdr[i];
    }
    __syncthreads();
    // ---- skinning backward: every wave evaluates the workgroup's vertices (lane = vertex) from LDS, each reduces a quarter of the sums
    float g[3] = {0.f, 0.f, 0.f}, vp[3] = {0.f, 0.f, 0.f}, T[12], gvp[3] = {0.f, 0.f, 0.f};
    float pf[GAB_POSE_FEATURES];
#pragma unroll
    for (int p = 0; p < GAB_POSE_FEATURES; ++p) pf[p] = 0.f;
    if (ok) {
        // the posed-before-skinning vertex and its blended transform (vertex_posed_and_T's arithmetic, the rows from LDS: same order of the sums)
        float po[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int p = 0; p < GAB_POSE_FEATURES; ++p) {
            const float f = a.ws[WS_PF + p];
            po[0] += f * pd[p][3 * lane]; po[1] += f * pd[p][3 * lane + 1]; po[2] += f * pd[p][3 * lane + 2];
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) vp[k] = po[k] + vs[k];
#pragma unroll
        for (int k = 0; k < 12; ++k) T[k] = 0.f;
#pragma unroll
        for (int j = 0; j < GAB_NUM_JOINTS; ++j)
#pragma unroll
            for (int k = 0; k < 12; ++k) T[k] += w[j] * a.ws[WS_A + 12 * j + k];
#pragma unroll
        for (int k = 0; k < 3; ++k) g[k] = gv[3 * lane + k] + gx[k];
#pragma unroll
        for (int c = 0; c < 3; ++c) gvp[c] = T[c] * g[0] + T[4 + c] * g[1] + T[8 + c] * g[2];
#pragma unroll
        for (int p = 0; p < GAB_POSE_FEATURES; ++p) pf[p] = pd[p][3 * lane] * gvp[0] + pd[p][3 * lane + 1] * gvp[1] + pd[p][3 * lane + 2] * gvp[2];
        if (wid == 0) { a.g_vs[3 * v] = gvp[0]; a.g_vs[3 * v + 1] = gvp[1]; a.g_vs[3 * v + 2] = gvp[2]; }
    }
    const float vph[4] = {vp[0], vp[1], vp[2], ok ? 1.f : 0.f};
    // wave 0: joints 0, 1; wave 1: joints 2, 3; wave 2: joint 4, d translation, pose features 0..8; wave 3: pose features 9..35
#pragma unroll
    for (int j = 0; j < GAB_NUM_JOINTS; ++j) {
        if (wid != (j >> 1)) continue;
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float s = wave_sum_hi(w[j] * g[r] * vph[c]);
                if (lane == 63) red[12 * j + 4 * r + c] = s;
            }
    }
    if (wid == 2) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const float s = wave_sum_hi(gx[r]);   // the external vertex gradient; the frames' share is the face centres' (tcs)
            if (lane == 63) red[60 + r] = s + ((tcs[0][r] + tcs[1][r]) + (tcs[2][r] + tcs[3][r]));
        }
    }
#pragma unroll
    for (int p = 0; p < GAB_POSE_FEATURES; ++p) {
        if (wid != (p < 9 ? 2 : 3)) continue;
        const float s = wave_sum_hi(pf[p]);
        if (lane == 63) red[63 + p] = s;
    }
    __syncthreads();
    if (tid < 99) {
        float* dst = tid < 60 ? &a.ws[WS_DA + tid] : (tid < 63 ? &a.ws[WS_DT + tid - 60] : &a.ws[WS_DPF + tid - 63]);
        unsafeAtomicAdd(dst, red[tid]);
    }
}

// ---------------------------------------------------------------------------------------------
// per-splat mesh-local -> world (get_xyz, get_scaling, get_rotation in one pass)
// ---------------------------------------------------------------------------------------------
using bindm::qmul;            // the transform itself lives in bind_math.h, shared with the rasterizer's bound entry
using bindm::qconj;
using bindm::qnorm_clamped;

__global__ __launch_bounds__(256) void k_bind(int N, const float* __restrict__ xyz, const float* __restrict__ log_scaling,
                                               const float* __restrict__ rotation, const void* __restrict__ binding, int is64,
                                               const float* __restrict__ fc, const float* __restrict__ fR, const float* __restrict__ fs,
                                               const float* __restrict__ fq, float* __restrict__ out_xyz, float* __restrict__ out_scaling,
                                               float* __restrict__ out_rotation, const float* __restrict__ opacity_logit,
                                               float* __restrict__ out_opacity)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    if (opacity_logit) out_opacity[i] = bindm::sigmoid(opacity_logit[i]);   // get_opacity (scene/gaussian_model.py:158-160) on the side
    const long long f = index_at(binding, is64, i);
    const float s = fs[f];
    float w[3];
    bindm::world_xyz(fR + 9 * f, s, fc + 3 * f, xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], w);
    for (int r = 0; r < 3; ++r) out_xyz[3 * i + r] = w[r];
    for (int k = 0; k < 3; ++k) out_scaling[3 * i + k] = bindm::world_scaling(log_scaling[3 * i + k], s);
    reinterpret_cast<float4*>(out_rotation)[i] = bindm::world_rotation(reinterpret_cast<const float4*>(fq)[f], reinterpret_cast<const float4*>(rotation)[i]);
}

__global__ __launch_bounds__(256) void k_bind_bwd(int N, const float* __restrict__ xyz, const float* __restrict__ log_scaling,
                                                   const float* __restrict__ rotation, const void* __restrict__ binding, int is64,
                                                   const float* __restrict__ fR, const float* __restrict__ fs, const float* __restrict__ fq,
                                                   const float* __restrict__ g_xyz, const float* __restrict__ g_scaling,
                                                   const float* __restrict__ g_rot, float* __restrict__ d_xyz,
                                                   float* __restrict__ d_log_scaling, float* __restrict__ d_rotation, float* __restrict__ d_face, int F,
                                                   const float* __restrict__ out_opacity, const float* __restrict__ g_opacity,
                                                   float* __restrict__ d_opacity_logit)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    if (d_opacity_logit) {
        const float o = out_opacity[i];
        d_opacity_logit[i] = (g_opacity ? g_opacity[i] : 0.f) * o * (1.f - o);
    }
    const long long f = index_at(binding, is64, i);
    const float s = fs[f];
    const float* R = fR + 9 * f;
    // d_face: four contiguous blocks  center (F,3) | orien_mat (F,9) | scaling (F,1) | orien_quat (F,4)
    float* dfc = d_face + 3 * f;
    float* dfR = d_face + (size_t)3 * F + 9 * f;
    float* dfs = d_face + (size_t)12 * F + f;
    float* dfq = d_face + (size_t)13 * F + 4 * f;
    float ds = 0.f;
    const float x[3] = {xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
    const float gx[3] = {g_xyz ? g_xyz[3 * i] : 0.f, g_xyz ? g_xyz[3 * i + 1] : 0.f, g_xyz ? g_xyz[3 * i + 2] : 0.f};
    for (int c = 0; c < 3; ++c) d_xyz[3 * i + c] = s * (R[c] * gx[0] + R[3 + c] * gx[1] + R[6 + c] * gx[2]);
    if (g_xyz) {
        for (int r = 0; r < 3; ++r) {
            unsafeAtomicAdd(&dfc[r], gx[r]);
            const float rx = R[3 * r] * x[0] + R[3 * r + 1] * x[1] + R[3 * r + 2] * x[2];
            ds += gx[r] * rx;
            for (int c = 0; c < 3; ++c) unsafeAtomicAdd(&dfR[3 * r + c], s * gx[r] * x[c]);
        }
    }
    for (int k = 0; k < 3; ++k) {
        const float e = expf(log_scaling[3 * i + k]);
        const float g = g_scaling ? g_scaling[3 * i + k] : 0.f;
        d_log_scaling[3 * i + k] = g * e * s;
        ds += g * e;
    }
    unsafeAtomicAdd(dfs, ds);
    const float4 qf = reinterpret_cast<const float4*>(fq)[f];
    const float4 q = reinterpret_cast<const float4*>(rotation)[i];
    const float na = qnorm_clamped(qf), nb = qnorm_clamped(q);
    const float4 a = make_float4(qf.x / na, qf.y / na, qf.z / na, qf.w / na);
    const float4 b = make_float4(q.x / nb, q.y / nb, q.z / nb, q.w / nb);
    const float4 g = g_rot ? reinterpret_cast<const float4*>(g_rot)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 da = qmul(g, qconj(b));   // <g, a*b> = <g*conj(b), a>
    const float4 db = qmul(qconj(a), g);   //           = <conj(a)*g, b>
    const float ada = a.x * da.x + a.y * da.y + a.z * da.z + a.w * da.w;
    const float bdb = b.x * db.x + b.y * db.y + b.z * db.z + b.w * db.w;
    reinterpret_cast<float4*>(d_rotation)[i] = make_float4((db.x - b.x * bdb) / nb, (db.y - b.y * bdb) / nb, (db.z - b.z * bdb) / nb, (db.w - b.w * bdb) / nb);
    if (g_rot) {
        unsafeAtomicAdd(&dfq[0], (da.x - a.x * ada) / na);
        unsafeAtomicAdd(&dfq[1], (da.y - a.y * ada) / na);
        unsafeAtomicAdd(&dfq[2], (da.z - a.z * ada) / na);
        unsafeAtomicAdd(&dfq[3], (da.w - a.w * ada) / na);
    }
}

// Atomic-free variant: splats are visited face by face through a CSR (order, face_begin) built once per
// binding change.  16 lanes share a face (lane k takes splats k, k+16, ...; ~10 splats per face, so nearly every
// splat has its own lane and all gathers are in flight at once), the 17 per-face sums are reduced
// over the 16 lanes on the DPP network and written once -- deterministic, no memset, no L2 atomics.
__global__ __launch_bounds__(256) void k_bind_bwd_csr(int F, const float* __restrict__ xyz, const float* __restrict__ log_scaling,
                                                       const float* __restrict__ rotation, const float* __restrict__ fR,
                                                       const float* __restrict__ fs, const float* __restrict__ fq,
                                                       const float* __restrict__ g_xyz, const float* __restrict__ g_scaling,
                                                       const float* __restrict__ g_rot, const int* __restrict__ order,
                                                       const int* __restrict__ face_begin, float* __restrict__ d_xyz,
                                                       float* __restrict__ d_log_scaling, float* __restrict__ d_rotation,
                                                       float* __restrict__ d_face, const float* __restrict__ out_opacity,
                                                       const float* __restrict__ g_opacity, float* __restrict__ d_opacity_logit)
{
    const int gtid = blockIdx.x * blockDim.x + threadIdx.x;
    const int f = gtid >> 4, sub = gtid & 15;
    const bool okf = f < F;
    float acc[17];
#pragma unroll
    for (int k = 0; k < 17; ++k) acc[k] = 0.f;
    if (okf) {
        const float s = fs[f];
        float R[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) R[k] = fR[9 * f + k];
        const float4 qf = reinterpret_cast<const float4*>(fq)[f];
        const float na = qnorm_clamped(qf);
        const float4 a = make_float4(qf.x / na, qf.y / na, qf.z / na, qf.w / na);
        const int b0 = face_begin[f], b1 = face_begin[f + 1];
        for (int j = b0 + sub; j < b1; j += 16) {
            const int i = order[j];
            if (d_opacity_logit) {
                const float o = out_opacity[i];
                d_opacity_logit[i] = (g_opacity ? g_opacity[i] : 0.f) * o * (1.f - o);
            }
            const float x[3] = {xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
            const float gx[3] = {g_xyz ? g_xyz[3 * i] : 0.f, g_xyz ? g_xyz[3 * i + 1] : 0.f, g_xyz ? g_xyz[3 * i + 2] : 0.f};
#pragma unroll
            for (int c = 0; c < 3; ++c) d_xyz[3 * i + c] = s * (R[c] * gx[0] + R[3 + c] * gx[1] + R[6 + c] * gx[2]);
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                acc[r] += gx[r];
                acc[12] += gx[r] * (R[3 * r] * x[0] + R[3 * r + 1] * x[1] + R[3 * r + 2] * x[2]);
#pragma unroll
                for (int c = 0; c < 3; ++c) acc[3 + 3 * r + c] += s * gx[r] * x[c];
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float e = expf(log_scaling[3 * i + k]);
                const float g = g_scaling ? g_scaling[3 * i + k] : 0.f;
                d_log_scaling[3 * i + k] = g * e * s;
                acc[12] += g * e;
            }
            const float4 q = reinterpret_cast<const float4*>(rotation)[i];
            const float nb = qnorm_clamped(q);
            const float4 b = make_float4(q.x / nb, q.y / nb, q.z / nb, q.w / nb);
            const float4 g = g_rot ? reinterpret_cast<const float4*>(g_rot)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 da = qmul(g, qconj(b));
            const float4 db = qmul(qconj(a), g);
            const float ada = a.x * da.x + a.y * da.y + a.z * da.z + a.w * da.w;
            const float bdb = b.x * db.x + b.y * db.y + b.z * db.z + b.w * db.w;
            reinterpret_cast<float4*>(d_rotation)[i] =
                make_float4((db.x - b.x * bdb) / nb, (db.y - b.y * bdb) / nb, (db.z - b.z * bdb) / nb, (db.w - b.w * bdb) / nb);
            acc[13] += (da.x - a.x * ada) / na;
            acc[14] += (da.y - a.y * ada) / na;
            acc[15] += (da.z - a.z * ada) / na;
            acc[16] += (da.w - a.w * ada) / na;
        }
    }
    // 16-lane sums (quad xor 1, xor 2, half-row mirror, row mirror); every lane of the group ends with the total
#pragma unroll
    for (int k = 0; k < 17; ++k) {
        float v = acc[k];
        v += dpp_f<0xB1, 0xf>(v);
        v += dpp_f<0x4E, 0xf>(v);
        v += dpp_f<0x141, 0xf>(v);
        v += dpp_f<0x140, 0xf>(v);
        acc[k] = v;
    }
    if (okf) {
        // the 16 lanes of the group share the 17 stores; d_face is four contiguous blocks
        // center (F,3) | orien_mat (F,9) | scaling (F,1) | orien_quat (F,4)
#pragma unroll
        for (int k = 0; k < 17; ++k) {
            if ((k & 15) != sub) continue;
            float* dst = k < 3 ? d_face + 3 * f + k
                       : k < 12 ? d_face + (size_t)3 * F + 9 * f + (k - 3)
                       : k < 13 ? d_face + (size_t)12 * F + f
                                : d_face + (size_t)13 * F + 4 * f + (k - 13);
            *dst = acc[k];
        }
    }
}

// Two-pass form of the CSR backward (what gab_bind_backward_csr runs when it is given splat_face / slot / rows).  The
// one-pass kernel above visits the splats in FACE order, so each of its eight per-splat reads and four writes is a
// scattered 4..16-byte access (12 sectors per splat: 78 MB of HBM transactions for 8 MB of data, and the kernel is bound
// by exactly that).  Here pass 1 runs in SPLAT order -- every per-splat array is read and written coalesced, the small
// per-face tables are gathered (L2-resident) -- and parks the splat's 17 contributions to its face as one 80-byte row at
// the splat's CSR position; pass 2 sums each face's contiguous rows (16 lanes per face, coalesced) on the DPP network.
// Deterministic like the one-pass kernel: the same values are added in the same CSR order.
#define GAB_BIND_ROW 20   // floats per row: 17 used, padded to a multiple of 16 bytes
__global__ __launch_bounds__(256) void k_bind_bwd_rows(int N, const float* __restrict__ xyz, const float* __restrict__ log_scaling,
                                                        const float* __restrict__ rotation, const int* __restrict__ splat_face,
                                                        const int* __restrict__ slot, const float* __restrict__ fR,
                                                        const float* __restrict__ fs, const float* __restrict__ fq,
                                                        const float* __restrict__ g_xyz, const float* __restrict__ g_scaling,
                                                        const float* __restrict__ g_rot, float* __restrict__ d_xyz,
                                                        float* __restrict__ d_log_scaling, float* __restrict__ d_rotation,
                                                        float* __restrict__ rows, const float* __restrict__ out_opacity,
                                                        const float* __restrict__ g_opacity, float* __restrict__ d_opacity_logit)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    if (d_opacity_logit) {
        const float o = out_opacity[i];
        d_opacity_logit[i] = (g_opacity ? g_opacity[i] : 0.f) * o * (1.f - o);
    }
    const int f = splat_face[i];
    const float s = fs[f];
    float R[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) R[k] = fR[9 * f + k];
    float acc[GAB_BIND_ROW];
#pragma unroll
    for (int k = 0; k < GAB_BIND_ROW; ++k) acc[k] = 0.f;
    const float x[3] = {xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
    const float gx[3] = {g_xyz ? g_xyz[3 * i] : 0.f, g_xyz ? g_xyz[3 * i + 1] : 0.f, g_xyz ? g_xyz[3 * i + 2] : 0.f};
#pragma unroll
    for (int c = 0; c < 3; ++c) d_xyz[3 * i + c] = s * (R[c] * gx[0] + R[3 + c] * gx[1] + R[6 + c] * gx[2]);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        acc[r] = gx[r];
        acc[12] += gx[r] * (R[3 * r] * x[0] + R[3 * r + 1] * x[1] + R[3 * r + 2] * x[2]);
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[3 + 3 * r + c] = s * gx[r] * x[c];
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float e = expf(log_scaling[3 * i + k]);
        const float g = g_scaling ? g_scaling[3 * i + k] : 0.f;
        d_log_scaling[3 * i + k] = g * e * s;
        acc[12] += g * e;
    }
    const float4 qf = reinterpret_cast<const float4*>(fq)[f];
    const float na = qnorm_clamped(qf);
    const float4 a = make_float4(qf.x / na, qf.y / na, qf.z / na, qf.w / na);
    const float4 q = reinterpret_cast<const float4*>(rotation)[i];
    const float nb = qnorm_clamped(q);
    const float4 b = make_float4(q.x / nb, q.y / nb, q.z / nb, q.w / nb);
    const float4 g = g_rot ? reinterpret_cast<const float4*>(g_rot)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 da = qmul(g, qconj(b));
    const float4 db = qmul(qconj(a), g);
    const float ada = a.x * da.x + a.y * da.y + a.z * da.z + a.w * da.w;
    const float bdb = b.x * db.x + b.y * db.y + b.z * db.z + b.w * db.w;
    reinterpret_cast<float4*>(d_rotation)[i] =
        make_float4((db.x - b.x * bdb) / nb, (db.y - b.y * bdb) / nb, (db.z - b.z * bdb) / nb, (db.w - b.w * bdb) / nb);
    acc[13] = (da.x - a.x * ada) / na;
    acc[14] = (da.y - a.y * ada) / na;
    acc[15] = (da.z - a.z * ada) / na;
    acc[16] = (da.w - a.w * ada) / na;
    float4* row = reinterpret_cast<float4*>(rows + (size_t)GAB_BIND_ROW * slot[i]);
#pragma unroll
    for (int k = 0; k < GAB_BIND_ROW / 4; ++k) row[k] = make_float4(acc[4 * k], acc[4 * k + 1], acc[4 * k + 2], acc[4 * k + 3]);
}

__global__ __launch_bounds__(256) void k_bind_bwd_faces(int F, const int* __restrict__ face_begin, const float* __restrict__ rows,
                                                         float* __restrict__ d_face)
{
    const int gtid = blockIdx.x * blockDim.x + threadIdx.x;
    const int f = gtid >> 4, sub = gtid & 15;
    const bool okf = f < F;
    float acc[GAB_BIND_ROW];
#pragma unroll
    for (int k = 0; k < GAB_BIND_ROW; ++k) acc[k] = 0.f;
    if (okf) {
        const int b0 = face_begin[f], b1 = face_begin[f + 1];
        for (int j = b0 + sub; j < b1; j += 16) {
            const float4* row = reinterpret_cast<const float4*>(rows + (size_t)GAB_BIND_ROW * j);
#pragma unroll
            for (int k = 0; k < GAB_BIND_ROW / 4; ++k) {
                const float4 v = row[k];
                acc[4 * k] += v.x; acc[4 * k + 1] += v.y; acc[4 * k + 2] += v.z; acc[4 * k + 3] += v.w;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 17; ++k) {
        float v = acc[k];
        v += dpp_f<0xB1, 0xf>(v);
        v += dpp_f<0x4E, 0xf>(v);
        v += dpp_f<0x141, 0xf>(v);
        v += dpp_f<0x140, 0xf>(v);
        acc[k] = v;
    }
    if (okf) {
#pragma unroll
        for (int k = 0; k < 17; ++k) {
            if ((k & 15) != sub) continue;
            float* dst = k < 3 ? d_face + 3 * f + k
                       : k < 12 ? d_face + (size_t)3 * F + 9 * f + (k - 3)
                       : k < 13 ? d_face + (size_t)12 * F + f
                                : d_face + (size_t)13 * F + 4 * f + (k - 13);
            *dst = acc[k];
        }
    }
}

// one launch that zero-fills up to 8 small buffers (the full-table gradients of the per-timestep FLAME rows)
// The frame feed of a recorded step (graphs.py): row schedule[cursor % n] of the packed per-timestep table -> the static one-row table the
// recorded kernels read; then the cursor moves on.  One workgroup; every thread reads the cursor before thread 0 advances it.
// ---------------------------------------------------------------------------------------------
// k_blend_seq_mfma: v_shaped of EVERY frame of an expression sequence in one launch,
//     out[t][e] = prepared[e] + sum_l shapedirs[e][n_shape + l] * expr[t][l]      (T x n_expr) . (n_expr x 3V), fp32,
// on the matrix cores: v_mfma_f32_32x32x2_f32, one wave per 32 frames x 32 outputs.  This is the one GEMM-shaped product of the path
// (SURVEY.md H9): per frame it is a GEMV that re-reads the 6 MB expression block, batched over the T frames of a sequence it is
// 2 T n_expr 3V flops (0.93 GFLOP at T = 300) on 6 MB + the 4 T 3V bytes it writes.  Operand layout of the instruction (one float per lane):
// A[m][k]: m = lane % 32, k = lane / 32;  B[k][n]: k = lane / 32, n = lane % 32;  D[i][j] in 16 registers: j = lane % 32,
// i = 8 (v / 4) + 4 (lane / 32) + v % 4.  The two k-slots of a step are fed from the two HALVES of the k range (lane half h walks
// k = h K/2 + s: any bijection of k onto (step, slot) gives the same sum), so that every lane streams through contiguous floats of its row.
// fp32 in, fp32 accumulate: the instruction is an exact fma chain; against the per-frame kernel (float4 partial sums + DPP) the sums differ
// in order only (tests: <= 1e-6 of the value range).
// ---------------------------------------------------------------------------------------------
typedef float gab_v16f __attribute__((ext_vector_type(16)));
#define GAB_SEQ_KMAX 100          // expression coefficients the LDS-staged product holds per tile row (FLAME: 100)
__global__ __launch_bounds__(256) void k_blend_seq_mfma(Rig rig, const float* __restrict__ prepared, const float* __restrict__ expr, int T, int MT,
                                                         float* __restrict__ out)
{
    // One workgroup = 128 outputs (four waves, 32 each) x MT tiles of 32 frames.  The operand tiles go through LDS: a lane's operands walk
    // along ITS row (32 rows per wave), which as global loads is 64 cache lines per instruction -- the texture path, not the matrix core,
    // paced the first version (54 us at T = 300) --; staged, the rows arrive as coalesced float4 streams and are read back bank-conflict-free
    // (row stride 101 floats).  The table tile (51 KB) is staged ONCE and kept for all MT frame tiles -- re-staging it per frame tile made
    // the kernel L2-bandwidth-bound (64 KB in for 1.3 us of matrix work per workgroup: 38 us at T = 300) --; the coefficient tile of the
    // next frame tile is in flight in registers while the matrix core works on the current one (two LDS buffers, one barrier per tile).
    __shared__ float tiles[6][32][GAB_SEQ_KMAX + 1];   // [0], [1]: the two coefficient buffers; [2 + wave]: the table tile of each wave
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int E = 3 * rig.V, K = rig.n_expr, NB = rig.n_shape + rig.n_expr;
    const int nb0 = (int)blockIdx.x * 128, n0 = nb0 + wid * 32;
    const int half = lane >> 5, l32 = lane & 31;
    const int Kh = (K + 1) / 2, kbeg = half * Kh, kend = min(K, kbeg + Kh);
    const bool staged = K <= GAB_SEQ_KMAX && (K & 3) == 0 && (rig.n_shape & 3) == 0 && (NB & 3) == 0;
    const int e = n0 + l32;
    const float base = e < E ? prepared[e] : 0.f;
    const int mt0 = (int)blockIdx.y * MT, mt1 = min(mt0 + MT, (T + 31) / 32);
    if (staged) {
        const int q = K >> 2;   // float4 per row
        constexpr int A_IT = (32 * GAB_SEQ_KMAX / 4 + 255) / 256, B_IT = (128 * GAB_SEQ_KMAX / 4 + 255) / 256, KH = GAB_SEQ_KMAX / 2;
        // LDS column of coefficient k: the lane half that owns it (k / Kh) times KH, plus its step.  Columns no coefficient maps to stay
        // zero, so the product loop below is KH unmasked steps whatever K is (K = 100: the identity).
        auto col = [&](int k) { return k >= Kh ? KH + k - Kh : k; };
        if (K < GAB_SEQ_KMAX) {
            for (int i = tid; i < 6 * 32 * (GAB_SEQ_KMAX + 1); i += 256) (&tiles[0][0][0])[i] = 0.f;
            __syncthreads();
        }
        float4 va[A_IT];
        auto fetch_a = [&](int mt) {
#pragma unroll
            for (int it = 0; it < A_IT; ++it) {
                const int i = tid + 256 * it, r = i / q, c = i - r * q;
                if (i < 32 * q) va[it] = *reinterpret_cast<const float4*>(expr + (size_t)min(32 * mt + r, T - 1) * K + 4 * c);
            }
        };
        auto park_a = [&](int buf) {
#pragma unroll
            for (int it = 0; it < A_IT; ++it) {
                const int i = tid + 256 * it, r = i / q, c = i - r * q;
                if (i < 32 * q) { float* d = tiles[buf][r]; d[col(4 * c)] = va[it].x; d[col(4 * c + 1)] = va[it].y; d[col(4 * c + 2)] = va[it].z; d[col(4 * c + 3)] = va[it].w; }
            }
        };
        fetch_a(mt0);
        {
            float4 vb[B_IT];   // every load in flight before the first LDS store: one memory latency per tile, not one per row
#pragma unroll
            for (int it = 0; it < B_IT; ++it) {
                const int i = tid + 256 * it, r = i / q, c = i - r * q;
                if (i < 128 * q) vb[it] = *reinterpret_cast<const float4*>(rig.shapedirs + (size_t)min(nb0 + r, E - 1) * NB + rig.n_shape + 4 * c);
            }
#pragma unroll
            for (int it = 0; it < B_IT; ++it) {
                const int i = tid + 256 * it, r = i / q, c = i - r * q;
                if (i < 128 * q) { float* d = tiles[2 + (r >> 5)][r & 31]; d[col(4 * c)] = vb[it].x; d[col(4 * c + 1)] = vb[it].y; d[col(4 * c + 2)] = vb[it].z; d[col(4 * c + 3)] = vb[it].w; }
            }
        }
        int buf = 0;
        for (int mt = mt0; mt < mt1; ++mt, buf ^= 1) {
            park_a(buf);
            __syncthreads();
            if (mt + 1 < mt1) fetch_a(mt + 1);
            gab_v16f acc;
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[v] = 0.f;
            const float* __restrict__ ap = &tiles[buf][l32][half * KH];
            const float* __restrict__ bp = &tiles[2 + wid][l32][half * KH];
#pragma unroll
            for (int g = 0; g < KH; g += 10) {   // ten steps' operands in flight ahead of their matrix instructions
                float a[10], b[10];
#pragma unroll
                for (int u = 0; u < 10; ++u) { a[u] = ap[g + u]; b[u] = bp[g + u]; }
#pragma unroll
                for (int u = 0; u < 10; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc, 0, 0, 0);
            }
            if (e < E) {
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int t = 32 * mt + 8 * (v >> 2) + 4 * half + (v & 3);
                    if (t < T) out[(size_t)t * E + e] = base + acc[v];
                }
            }
        }
    } else {
        if (n0 >= E) return;
        const float* __restrict__ brow = rig.shapedirs + (size_t)min(n0 + l32, E - 1) * NB + rig.n_shape;   // B[k][n = l32]
        for (int mt = mt0; mt < mt1; ++mt) {
            const float* __restrict__ arow = expr + (size_t)min(32 * mt + l32, T - 1) * K;                      // A[m = l32][k]
            gab_v16f acc;
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[v] = 0.f;
            for (int s = 0; s < Kh; ++s) {
                const int k = kbeg + s;
                const float a = k < kend ? arow[k] : 0.f, b = k < kend ? brow[k] : 0.f;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
            }
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int t = 32 * mt + 8 * (v >> 2) + 4 * half + (v & 3);
                if (t < T && e < E) out[(size_t)t * E + e] = base + acc[v];
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_feed_row(const float* __restrict__ packed, int T, int width, const int* __restrict__ schedule, int n_sched,
                                                   int* __restrict__ cursor, float* __restrict__ row)
{
    const unsigned c = (unsigned)*cursor;   // kept inside [0, period): a cursor that only grew would overflow after 2^31 feeds
    const unsigned period = schedule ? (unsigned)n_sched : (unsigned)T;
    int t = schedule ? schedule[c % period] : (int)c;
    t = (int)((unsigned)t % (unsigned)T);
    for (int k = threadIdx.x; k < width; k += 256) row[k] = packed[(size_t)t * width + k];
    __syncthreads();
    if (threadIdx.x == 0) *cursor = (int)((c + 1u) % period);
}

__global__ __launch_bounds__(256) void k_zero_many(ZeroSpec z)
{
    const int b = blockIdx.y;
    if (b >= z.count) return;
    float* p = z.p[b];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < z.n[b]; i += gridDim.x * blockDim.x) p[i] = 0.f;
}

}  // namespace gab

// =================================================================================================
// C ABI
// =================================================================================================
namespace {
thread_local char g_err[512] = "";
int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
#define HIP_TRY(expr)                                                                                    \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess) return fail(GAB_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));     \
    } while (0)
#define LAUNCH_CHECK(name)                                                                               \
    do {                                                                                                 \
        hipError_t e_ = hipGetLastError();                                                               \
        if (e_ != hipSuccess) return fail(GAB_E_HIP, "launch of %s failed: %s", name, hipGetErrorString(e_)); \
    } while (0)

int to_rig(const GabRig* r, gab::Rig* o)
{
    if (!r) return fail(GAB_E_ARG, "rig is NULL");
    if (r->V <= 0 || r->n_shape < 0 || r->n_expr < 0 || r->n_shape + r->n_expr <= 0) return fail(GAB_E_ARG, "bad rig sizes");
    if (!r->v_template || !r->shapedirs || !r->posedirs || !r->J_regressor || !r->lbs_weights) return fail(GAB_E_ARG, "NULL rig buffer");
    if (r->parents[0] != -1) return fail(GAB_E_ARG, "parents[0] must be -1");
    for (int i = 1; i < GAB_NUM_JOINTS; ++i)
        if (r->parents[i] < 0 || r->parents[i] >= i) return fail(GAB_E_ARG, "parents must be topologically ordered");
    o->V = r->V; o->n_shape = r->n_shape; o->n_expr = r->n_expr;
    o->v_template = r->v_template; o->shapedirs = r->shapedirs; o->posedirs = r->posedirs;
    o->J_regressor = r->J_regressor; o->lbs_weights = r->lbs_weights;
    for (int i = 0; i < GAB_NUM_JOINTS; ++i) o->parents[i] = r->parents[i];
    return 0;
}
}  // namespace

extern "C" {

int gab_abi_version(void) { return GAB_ABI_VERSION; }
const char* gab_last_error(void) { return g_err; }

int gab_flame_forward(const GabRig* rig_, const float* shape, const float* expr, const float* rotation, const float* neck,
                      const float* jaw, const float* eyes, const float* translation, const float* static_offset, float* verts,
                      float* v_shaped, float* ws, void* stream_)
{
    gab::Rig rig;
    if (int rc = to_rig(rig_, &rig)) return rc;
    if ((rig.n_shape && !shape) || (rig.n_expr && !expr) || !rotation || !neck || !jaw || !eyes || !translation || !verts || !v_shaped || !ws)
        return fail(GAB_E_ARG, "gab_flame_forward: NULL buffer");
    hipStream_t st = (hipStream_t)stream_;
    const int E = 3 * rig.V;
    PROF_LAUNCH(gab::k_blend, dim3((E + 3) / 4), dim3(256), 0, st, rig, shape, expr, static_offset, v_shaped);
    LAUNCH_CHECK("k_blend");
    const bool flame_tree = rig.parents[1] == 0 && rig.parents[2] == 1 && rig.parents[3] == 1 && rig.parents[4] == 1;
    if (flame_tree)
        PROF_LAUNCH(gab::k_joints_chain<true>, dim3(1), dim3(1024), 0, st, rig, (const float*)v_shaped, rotation, neck, jaw, eyes, ws);
    else
        PROF_LAUNCH(gab::k_joints_chain<false>, dim3(1), dim3(1024), 0, st, rig, (const float*)v_shaped, rotation, neck, jaw, eyes, ws);
    LAUNCH_CHECK("k_joints_chain");
    PROF_LAUNCH(gab::k_skin, dim3((rig.V + 255) / 256), dim3(256), 0, st, rig, (const float*)ws, (const float*)v_shaped, translation, verts);
    LAUNCH_CHECK("k_skin");
    return GAB_OK;
}

int gab_flame_backward(const GabRig* rig_, const float* shape, const float* expr, const float* rotation, const float* neck,
                       const float* jaw, const float* eyes, const float* translation, const float* static_offset,
                       const float* v_shaped, float* ws, const float* dL_dverts, const float* dL_dv_shaped, float* d_shape,
                       float* d_expr, float* d_rotation, float* d_neck, float* d_jaw, float* d_eyes, float* d_translation,
                       float* d_static_offset, float* scratch, int32_t zero_count, float* const* zero_buffers_host,
                       const int32_t* zero_sizes_host, void* stream_)
{
    (void)shape; (void)expr; (void)translation; (void)static_offset;
    if (zero_count < 0 || zero_count > 8 || (zero_count > 0 && (!zero_buffers_host || !zero_sizes_host)))
        return fail(GAB_E_ARG, "gab_flame_backward: 0..8 zero-fill buffers");
    gab::ZeroSpec zs;
    zs.count = zero_count;
    for (int i = 0; i < 8; ++i) {
        zs.p[i] = i < zero_count ? zero_buffers_host[i] : nullptr;
        zs.n[i] = i < zero_count ? zero_sizes_host[i] : 0;
        if (i < zero_count && (zs.n[i] < 0 || (zs.n[i] > 0 && !zs.p[i]))) return fail(GAB_E_ARG, "gab_flame_backward: bad zero-fill buffer %d", i);
    }
    gab::Rig rig;
    if (int rc = to_rig(rig_, &rig)) return rc;
    if (!rotation || !neck || !jaw || !eyes || !v_shaped || !ws || !dL_dverts || !d_expr || !d_rotation || !d_neck || !d_jaw || !d_eyes ||
        !d_translation || !scratch)
        return fail(GAB_E_ARG, "gab_flame_backward: NULL buffer");
    hipStream_t st = (hipStream_t)stream_;
    const int E = 3 * rig.V;
    PROF_LAUNCH(gab::k_skin_bwd, dim3((rig.V + 255) / 256), dim3(256), 0, st, rig, ws, v_shaped, dL_dverts, scratch, zs);
    LAUNCH_CHECK("k_skin_bwd");
    const bool flame_tree = rig.parents[1] == 0 && rig.parents[2] == 1 && rig.parents[3] == 1 && rig.parents[4] == 1;
    if (flame_tree)
        PROF_LAUNCH(gab::k_chain_bwd<true>, dim3(1), dim3(64), 0, st, rig, ws, rotation, neck, jaw, eyes, d_rotation, d_neck, d_jaw, d_eyes, d_translation, d_expr, d_shape);
    else
        PROF_LAUNCH(gab::k_chain_bwd<false>, dim3(1), dim3(64), 0, st, rig, ws, rotation, neck, jaw, eyes, d_rotation, d_neck, d_jaw, d_eyes, d_translation, d_expr, d_shape);
    LAUNCH_CHECK("k_chain_bwd");
    PROF_LAUNCH(gab::k_blend_bwd, dim3((E + GAB_BLEND_BWD_ROWS - 1) / GAB_BLEND_BWD_ROWS), dim3(256), 0, st, rig, (const float*)ws,
                       (const float*)scratch, dL_dv_shaped, d_static_offset, d_shape, d_expr);
    LAUNCH_CHECK("k_blend_bwd");
    return GAB_OK;
}

int64_t gab_flame_prepared_floats(const GabRig* rig_)
{
    if (!rig_ || rig_->V <= 0 || rig_->n_expr < 0) return -1;
    return (int64_t)gab::prep_joint_offset(rig_->V) + 16 + 15 * (int64_t)rig_->n_expr;
}

int gab_flame_prepare(const GabRig* rig_, const float* shape, const float* static_offset, float* prepared, void* stream_)
{
    gab::Rig rig;
    if (int rc = to_rig(rig_, &rig)) return rc;
    if ((rig.n_shape && !shape) || !prepared) return fail(GAB_E_ARG, "gab_flame_prepare: NULL buffer");
    hipStream_t st = (hipStream_t)stream_;
    const int E = 3 * rig.V, outs = 15 + 15 * rig.n_expr;
    PROF_LAUNCH(gab::k_prep_rows, dim3((E + 3) / 4), dim3(256), 0, st, rig, shape, static_offset, prepared);
    LAUNCH_CHECK("k_prep_rows");
    PROF_LAUNCH(gab::k_prep_joints, dim3((outs + 3) / 4), dim3(256), 0, st, rig, prepared);
    LAUNCH_CHECK("k_prep_joints");
    return GAB_OK;
}

int gab_flame_forward_prepared(const GabRig* rig_, const float* prepared, const float* expr, const float* rotation, const float* neck,
                               const float* jaw, const float* eyes, const float* translation, float* verts, float* v_shaped, float* ws,
                               void* stream_)
{
    gab::Rig rig;
    if (int rc = to_rig(rig_, &rig)) return rc;
    if (!prepared || (rig.n_expr && !expr) || !rotation || !neck || !jaw || !eyes || !translation || !verts || !v_shaped || !ws)
        return fail(GAB_E_ARG, "gab_flame_forward_prepared: NULL buffer");
    hipStream_t st = (hipStream_t)stream_;
    const int blocks = (rig.V + GAB_FUSED_VERTS - 1) / GAB_FUSED_VERTS;
    const bool flame_tree = rig.parents[1] == 0 && rig.parents[2] == 1 && rig.parents[3] == 1 && rig.parents[4] == 1;
    if (flame_tree)
        PROF_LAUNCH(gab::k_flame_fused<true>, dim3(blocks), dim3(256), 0, st, rig, prepared, expr, rotation, neck, jaw, eyes, translation, verts, v_shaped, ws, (const float*)nullptr);
    else
        PROF_LAUNCH(gab::k_flame_fused<false>, dim3(blocks), dim3(256), 0, st, rig, prepared, expr, rotation, neck, jaw, eyes, translation, verts, v_shaped, ws, (const float*)nullptr);
    LAUNCH_CHECK("k_flame_fused");
    return GAB_OK;
}

int gab_blend_sequence(const GabRig* rig_, const float* prepared, const float* expr_table, int32_t T, float* v_shaped_seq, void* stream_)
{
    gab::Rig rig;
    if (int rc = to_rig(rig_, &rig)) return rc;
    if (T < 0 || (T > 0 && (!prepared || !expr_table || !v_shaped_seq))) return fail(GAB_E_ARG, "gab_blend_sequence: bad arguments");
    if (T == 0) return GAB_OK;
    if (rig.n_expr <= 0) return fail(GAB_E_ARG, "gab_blend_sequence: the rig has no expression block");
    const int E = 3 * rig.V;
    // frame tiles per workgroup: as many as keep the launch at about two workgroups per CU (what the LDS tiles allow), so that the table
    // tile each workgroup stages is re-used instead of re-read
    const int gx = ((E + 31) / 32 + 3) / 4, mtiles = (T + 31) / 32;
    const int MT = std::min(16, std::max(1, (mtiles * gx + 511) / 512));
    dim3 grid((unsigned)gx, (unsigned)((mtiles + MT - 1) / MT));
    PROF_LAUNCH(gab::k_blend_seq_mfma, grid, dim3(256), 0, (hipStream_t)stream_, rig, prepared, expr_table, (int)T, MT, v_shaped_seq);
    LAUNCH_CHECK("k_blend_seq_mfma");
    return GAB_OK;
}

int gab_flame_forward_sequence(const GabRig* rig_, const float* prepared, const float* v_shaped_row, const float* expr, const float* rotation,
                               const float* neck, const float* jaw, const float* eyes, const float* translation, float* verts, float* v_shaped,
                               float* ws, void* stream_)
{
    gab::Rig rig;
    if (int rc = to_rig(rig_, &rig)) return rc;
    if (!prepared || !v_shaped_row || (rig.n_expr && !expr) || !rotation || !neck || !jaw || !eyes || !translation || !verts || !v_shaped || !ws)
        return fail(GAB_E_ARG, "gab_flame_forward_sequence: NULL buffer");
    hipStream_t st = (hipStream_t)stream_;
    const int blocks = (rig.V + GAB_FUSED_VERTS - 1) / GAB_FUSED_VERTS;
    const bool flame_tree = rig.parents[1] == 0 && rig.parents[2] == 1 && rig.parents[3] == 1 && rig.parents[4] == 1;
    if (flame_tree)
        PROF_LAUNCH(gab::k_flame_fused<true>, dim3(blocks), dim3(256), 0, st, rig, prepared, expr, rotation, neck, jaw, eyes, translation, verts, v_shaped, ws, v_shaped_row);
    else
        PROF_LAUNCH(gab::k_flame_fused<false>, dim3(blocks), dim3(256), 0, st, rig, prepared, expr, rotation, neck, jaw, eyes, translation, verts, v_shaped, ws, v_shaped_row);
    LAUNCH_CHECK("k_flame_fused");
    return GAB_OK;
}

int gab_flame_backward_prepared(const GabRig* rig_, const float* prepared, const float* rotation, const float* neck, const float* jaw,
                                const float* eyes, const float* v_shaped, float* ws, const float* dL_dverts, float* d_expr,
                                float* d_rotation, float* d_neck, float* d_jaw, float* d_eyes, float* d_translation, float* scratch,
                                int32_t zero_count, float* const* zero_buffers_host, const int32_t* zero_sizes_host, void* stream_)
{
    if (zero_count < 0 || zero_count > 7 || (zero_count > 0 && (!zero_buffers_host || !zero_sizes_host)))
        return fail(GAB_E_ARG, "gab_flame_backward_prepared: 0..7 zero-fill buffers");
    gab::Rig rig;
    if (int rc = to_rig(rig_, &rig)) return rc;
    if (!prepared || !rotation || !neck || !jaw || !eyes || !v_shaped || !ws || !dL_dverts || !d_expr || !d_rotation || !d_neck || !d_jaw ||
        !d_eyes || !d_translation || !scratch)
        return fail(GAB_E_ARG, "gab_flame_backward_prepared: NULL buffer");
    gab::ZeroSpec zs;
    bool covers_expr = false;   // d_expr is added into by both roles of the second launch: it must be zero before that launch
    for (int i = 0; i < 8; ++i) {
        zs.p[i] = i < zero_count ? zero_buffers_host[i] : nullptr;
        zs.n[i] = i < zero_count ? zero_sizes_host[i] : 0;
        if (i < zero_count && (zs.n[i] < 0 || (zs.n[i] > 0 && !zs.p[i]))) return fail(GAB_E_ARG, "gab_flame_backward_prepared: bad zero-fill buffer %d", i);
        if (i < zero_count && d_expr >= zs.p[i] && d_expr + rig.n_expr <= zs.p[i] + zs.n[i]) covers_expr = true;
    }
    zs.count = zero_count;
    if (!covers_expr) { zs.p[zs.count] = d_expr; zs.n[zs.count] = rig.n_expr; ++zs.count; }
    hipStream_t st = (hipStream_t)stream_;
    const int E = 3 * rig.V;
    PROF_LAUNCH(gab::k_skin_bwd, dim3((rig.V + 255) / 256), dim3(256), 0, st, rig, ws, v_shaped, dL_dverts, scratch, zs);
    LAUNCH_CHECK("k_skin_bwd");
    const float* Mmat = prepared + gab::prep_joint_offset(rig.V) + 16;
    const int blocks = 1 + (E + GAB_BLEND_BWD_ROWS - 1) / GAB_BLEND_BWD_ROWS;
    const bool flame_tree = rig.parents[1] == 0 && rig.parents[2] == 1 && rig.parents[3] == 1 && rig.parents[4] == 1;
    if (flame_tree)
        PROF_LAUNCH(gab::k_chain_blend_bwd<true>, dim3(blocks), dim3(256), 0, st, rig, ws, (const float*)scratch, Mmat, rotation, neck, jaw, eyes,
                           d_rotation, d_neck, d_jaw, d_eyes, d_translation, d_expr);
    else
        PROF_LAUNCH(gab::k_chain_blend_bwd<false>, dim3(blocks), dim3(256), 0, st, rig, ws, (const float*)scratch, Mmat, rotation, neck, jaw, eyes,
                           d_rotation, d_neck, d_jaw, d_eyes, d_translation, d_expr);
    LAUNCH_CHECK("k_chain_blend_bwd");
    return GAB_OK;
}

int gab_mesh_backward_prepared(const GabRig* rig_, const float* prepared, const float* rotation, const float* neck, const float* jaw,
                               const float* eyes, const float* v_shaped, float* ws, const float* verts, const int32_t* vf_begin,
                               const int32_t* vf_list, const float* d_center, const float* d_orien_mat, const float* d_scaling,
                               const float* d_orien_quat, const float* dL_dverts, float* d_expr, float* d_rotation, float* d_neck,
                               float* d_jaw, float* d_eyes, float* d_translation, float* scratch, int32_t zero_count,
                               float* const* zero_buffers_host, const int32_t* zero_sizes_host, void* stream_)
{
    if (zero_count < 0 || zero_count > 7 || (zero_count > 0 && (!zero_buffers_host || !zero_sizes_host)))
        return fail(GAB_E_ARG, "gab_mesh_backward_prepared: 0..7 zero-fill buffers");
    gab::Rig rig;
    if (int rc = to_rig(rig_, &rig)) return rc;
    if (!prepared || !rotation || !neck || !jaw || !eyes || !v_shaped || !ws || !verts || !vf_begin || !vf_list || !d_expr || !d_rotation ||
        !d_neck || !d_jaw || !d_eyes || !d_translation || !scratch)
        return fail(GAB_E_ARG, "gab_mesh_backward_prepared: NULL buffer");
    gab::ZeroSpec zs;
    bool covers_expr = false;   // d_expr is added into by both roles of the second launch: it must be zero before that launch
    for (int i = 0; i < 8; ++i) {
        zs.p[i] = i < zero_count ? zero_buffers_host[i] : nullptr;
        zs.n[i] = i < zero_count ? zero_sizes_host[i] : 0;
        if (i < zero_count && (zs.n[i] < 0 || (zs.n[i] > 0 && !zs.p[i]))) return fail(GAB_E_ARG, "gab_mesh_backward_prepared: bad zero-fill buffer %d", i);
        if (i < zero_count && d_expr >= zs.p[i] && d_expr + rig.n_expr <= zs.p[i] + zs.n[i]) covers_expr = true;
    }
    zs.count = zero_count;
    if (!covers_expr) { zs.p[zs.count] = d_expr; zs.n[zs.count] = rig.n_expr; ++zs.count; }
    gab::GatherSkinArgs a;
    a.v_shaped = v_shaped; a.ws = ws; a.verts = verts;
    a.vf_begin = vf_begin; a.vf_list = reinterpret_cast<const int4*>(vf_list);
    a.d_center = d_center; a.d_R = d_orien_mat; a.d_scaling = d_scaling; a.d_quat = d_orien_quat; a.g_verts = dL_dverts;
    a.g_vs = scratch;
    hipStream_t st = (hipStream_t)stream_;
    PROF_LAUNCH(gab::k_gather_skin_bwd, dim3((rig.V + GAB_MESH_VPB - 1) / GAB_MESH_VPB), dim3(256), 0, st, rig, a, zs);
    LAUNCH_CHECK("k_gather_skin_bwd");
    const int E = 3 * rig.V;
    const float* Mmat = prepared + gab::prep_joint_offset(rig.V) + 16;
    const int blocks = 1 + (E + GAB_BLEND_BWD_ROWS - 1) / GAB_BLEND_BWD_ROWS;
    const bool flame_tree = rig.parents[1] == 0 && rig.parents[2] == 1 && rig.parents[3] == 1 && rig.parents[4] == 1;
    if (flame_tree)
        PROF_LAUNCH(gab::k_chain_blend_bwd<true>, dim3(blocks), dim3(256), 0, st, rig, ws, (const float*)scratch, Mmat, rotation, neck, jaw, eyes,
                           d_rotation, d_neck, d_jaw, d_eyes, d_translation, d_expr);
    else
        PROF_LAUNCH(gab::k_chain_blend_bwd<false>, dim3(blocks), dim3(256), 0, st, rig, ws, (const float*)scratch, Mmat, rotation, neck, jaw, eyes,
                           d_rotation, d_neck, d_jaw, d_eyes, d_translation, d_expr);
    LAUNCH_CHECK("k_chain_blend_bwd");
    return GAB_OK;
}

int gab_face_frames_forward(int32_t V, int32_t F, const float* verts, const void* faces, int32_t is64, float* center, float* orien_mat,
                            float* scaling, float* orien_quat, float* d_verts_zeroed, void* stream_)
{
    if (V <= 0 || F < 0) return fail(GAB_E_ARG, "bad sizes");
    if (F == 0) {
        if (d_verts_zeroed) HIP_TRY(hipMemsetAsync(d_verts_zeroed, 0, (size_t)V * 3 * sizeof(float), (hipStream_t)stream_));
        return GAB_OK;
    }
    if (!verts || !faces || !center || !orien_mat || !scaling || !orien_quat) return fail(GAB_E_ARG, "gab_face_frames_forward: NULL buffer");
    PROF_LAUNCH(gab::k_face_frames, dim3((F + 255) / 256), dim3(256), 0, (hipStream_t)stream_, F, verts, faces, is64, center, orien_mat,
                       scaling, orien_quat, d_verts_zeroed, 3 * V);
    LAUNCH_CHECK("k_face_frames");
    return GAB_OK;
}

int gab_face_frames_backward(int32_t V, int32_t F, const float* verts, const void* faces, int32_t is64, const float* d_center,
                             const float* d_orien_mat, const float* d_scaling, const float* d_orien_quat, float* d_verts,
                             int32_t d_verts_is_zero, void* stream_)
{
    if (V <= 0 || F < 0) return fail(GAB_E_ARG, "bad sizes");
    if (!d_verts) return fail(GAB_E_ARG, "d_verts is NULL");
    hipStream_t st = (hipStream_t)stream_;
    if (!d_verts_is_zero) HIP_TRY(hipMemsetAsync(d_verts, 0, (size_t)V * 3 * sizeof(float), st));
    if (F == 0) return GAB_OK;
    if (!verts || !faces) return fail(GAB_E_ARG, "gab_face_frames_backward: NULL buffer");
    PROF_LAUNCH(gab::k_face_frames_bwd, dim3((F + 255) / 256), dim3(256), 0, st, F, verts, faces, is64, d_center, d_orien_mat, d_scaling,
                       d_orien_quat, d_verts);
    LAUNCH_CHECK("k_face_frames_bwd");
    return GAB_OK;
}

int gab_bind_forward(int32_t N, int32_t F, const float* xyz, const float* log_scaling, const float* rotation, const void* binding,
                     int32_t is64, const float* face_center, const float* face_orien_mat, const float* face_scaling,
                     const float* face_orien_quat, float* out_xyz, float* out_scaling, float* out_rotation, const float* opacity_logit,
                     float* out_opacity, void* stream_)
{
    if (N < 0 || F <= 0) return fail(GAB_E_ARG, "bad sizes");
    if (N == 0) return GAB_OK;
    if ((opacity_logit == nullptr) != (out_opacity == nullptr)) return fail(GAB_E_ARG, "opacity_logit and out_opacity go together");
    if (!xyz || !log_scaling || !rotation || !binding || !face_center || !face_orien_mat || !face_scaling || !face_orien_quat || !out_xyz ||
        !out_scaling || !out_rotation)
        return fail(GAB_E_ARG, "gab_bind_forward: NULL buffer");
    PROF_LAUNCH(gab::k_bind, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream_, N, xyz, log_scaling, rotation, binding, is64,
                       face_center, face_orien_mat, face_scaling, face_orien_quat, out_xyz, out_scaling, out_rotation, opacity_logit, out_opacity);
    LAUNCH_CHECK("k_bind");
    return GAB_OK;
}

int gab_bind_backward(int32_t N, int32_t F, const float* xyz, const float* log_scaling, const float* rotation, const void* binding,
                      int32_t is64, const float* face_center, const float* face_orien_mat, const float* face_scaling,
                      const float* face_orien_quat, const float* d_out_xyz, const float* d_out_scaling, const float* d_out_rotation,
                      float* d_xyz, float* d_log_scaling, float* d_rotation, float* d_face, const float* out_opacity,
                      const float* d_out_opacity, float* d_opacity_logit, void* stream_)
{
    (void)face_center;
    if (d_opacity_logit && !out_opacity) return fail(GAB_E_ARG, "d_opacity_logit needs out_opacity");
    if (N < 0 || F <= 0) return fail(GAB_E_ARG, "bad sizes");
    if (!d_face) return fail(GAB_E_ARG, "d_face is NULL");
    hipStream_t st = (hipStream_t)stream_;
    HIP_TRY(hipMemsetAsync(d_face, 0, (size_t)F * 17 * sizeof(float), st));
    if (N == 0) return GAB_OK;
    if (!xyz || !log_scaling || !rotation || !binding || !face_orien_mat || !face_scaling || !face_orien_quat || !d_xyz || !d_log_scaling ||
        !d_rotation)
        return fail(GAB_E_ARG, "gab_bind_backward: NULL buffer");
    PROF_LAUNCH(gab::k_bind_bwd, dim3((N + 255) / 256), dim3(256), 0, st, N, xyz, log_scaling, rotation, binding, is64, face_orien_mat,
                       face_scaling, face_orien_quat, d_out_xyz, d_out_scaling, d_out_rotation, d_xyz, d_log_scaling, d_rotation, d_face, F,
                       out_opacity, d_out_opacity, d_opacity_logit);
    LAUNCH_CHECK("k_bind_bwd");
    return GAB_OK;
}

int gab_bind_backward_csr(int32_t N, int32_t F, const float* xyz, const float* log_scaling, const float* rotation,
                          const float* face_orien_mat, const float* face_scaling, const float* face_orien_quat,
                          const float* d_out_xyz, const float* d_out_scaling, const float* d_out_rotation, const int32_t* order,
                          const int32_t* face_begin, float* d_xyz, float* d_log_scaling, float* d_rotation, float* d_face,
                          const float* out_opacity, const float* d_out_opacity, float* d_opacity_logit,
                          const int32_t* splat_face, const int32_t* slot, float* rows, void* stream_)
{
    if (d_opacity_logit && !out_opacity) return fail(GAB_E_ARG, "d_opacity_logit needs out_opacity");
    if (N < 0 || F <= 0) return fail(GAB_E_ARG, "bad sizes");
    if (!d_face || !order || !face_begin) return fail(GAB_E_ARG, "gab_bind_backward_csr: NULL buffer");
    if (N > 0 && (!xyz || !log_scaling || !rotation || !face_orien_mat || !face_scaling || !face_orien_quat || !d_xyz || !d_log_scaling ||
                  !d_rotation))
        return fail(GAB_E_ARG, "gab_bind_backward_csr: NULL buffer");
    const long long threads = 16ll * F;
    if (splat_face && slot && rows) {   // two passes: splat order (coalesced), then face order over contiguous rows
        if (N > 0) {
            PROF_LAUNCH(gab::k_bind_bwd_rows, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, N, xyz, log_scaling,
                               rotation, splat_face, slot, face_orien_mat, face_scaling, face_orien_quat, d_out_xyz, d_out_scaling,
                               d_out_rotation, d_xyz, d_log_scaling, d_rotation, rows, out_opacity, d_out_opacity, d_opacity_logit);
            LAUNCH_CHECK("k_bind_bwd_rows");
        }
        PROF_LAUNCH(gab::k_bind_bwd_faces, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, F, face_begin,
                           (const float*)rows, d_face);
        LAUNCH_CHECK("k_bind_bwd_faces");
        return GAB_OK;
    }
    PROF_LAUNCH(gab::k_bind_bwd_csr, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, F, xyz, log_scaling,
                       rotation, face_orien_mat, face_scaling, face_orien_quat, d_out_xyz, d_out_scaling, d_out_rotation, order, face_begin,
                       d_xyz, d_log_scaling, d_rotation, d_face, out_opacity, d_out_opacity, d_opacity_logit);
    LAUNCH_CHECK("k_bind_bwd_csr");
    return GAB_OK;
}

int gab_bind_backward_faces(int32_t F, const int32_t* face_begin, const float* rows, float* d_face, void* stream_)
{
    if (F <= 0 || !face_begin || !rows || !d_face) return fail(GAB_E_ARG, "gab_bind_backward_faces: bad arguments");
    const long long threads = 16ll * F;
    PROF_LAUNCH(gab::k_bind_bwd_faces, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, F, face_begin, rows, d_face);
    LAUNCH_CHECK("k_bind_bwd_faces");
    return GAB_OK;
}

int gab_feed_row(const float* packed, int32_t T, int32_t width, const int32_t* schedule, int32_t n_sched, int32_t* cursor, float* row,
                 void* stream_)
{
    if (T <= 0 || width <= 0 || !packed || !cursor || !row || (schedule && n_sched <= 0)) return fail(GAB_E_ARG, "gab_feed_row: bad arguments");
    PROF_LAUNCH(gab::k_feed_row, dim3(1), dim3(256), 0, (hipStream_t)stream_, packed, T, width, schedule, n_sched, cursor, row);
    LAUNCH_CHECK("k_feed_row");
    return GAB_OK;
}

int gab_zero_buffers(int32_t count, float* const* buffers_host, const int32_t* sizes_host, void* stream_)
{
    if (count < 0 || count > 8 || (count > 0 && (!buffers_host || !sizes_host))) return fail(GAB_E_ARG, "gab_zero_buffers: 0..8 buffers");
    if (count == 0) return GAB_OK;
    gab::ZeroSpec z;
    z.count = count;
    int mx = 0;
    for (int i = 0; i < 8; ++i) {
        z.p[i] = i < count ? buffers_host[i] : nullptr;
        z.n[i] = i < count ? sizes_host[i] : 0;
        if (i < count && (sizes_host[i] < 0 || (sizes_host[i] > 0 && !buffers_host[i]))) return fail(GAB_E_ARG, "gab_zero_buffers: bad buffer %d", i);
        if (z.n[i] > mx) mx = z.n[i];
    }
    int bx = (mx + 255) / 256;
    if (bx < 1) bx = 1;
    if (bx > 64) bx = 64;
    PROF_LAUNCH(gab::k_zero_many, dim3(bx, count), dim3(256), 0, (hipStream_t)stream_, z);
    LAUNCH_CHECK("k_zero_many");
    return GAB_OK;
}

int gab_profile_enable(int on)
{
    lprof::g.on.store(on ? 1 : 0);
    return 0;
}
int gab_profile_collect(void) { return lprof::collect(); }
int gab_profile_entry(int32_t index, const char** name, double* total_ms, int64_t* launches)
{
    long long n = 0;
    const int rc = lprof::entry(index, name, total_ms, &n);
    if (launches) *launches = (int64_t)n;
    return rc;
}
int gab_profile_reset(void)
{
    lprof::reset();
    return 0;
}

}  // extern "C"
