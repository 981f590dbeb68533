// bind_math.h -- the per-splat mesh-local -> world transform of a bound model (scene/gaussian_model.py:113-160):
//     xyz      = (face_orien_mat . _xyz) * face_scaling + face_center
//     scaling  = exp(_scaling) * face_scaling
//     rotation = quat_product(normalize(face_orien_quat), normalize(_rotation))        (WXYZ)
//     opacity  = sigmoid(_opacity)
// shared by gab::k_bind (libgab_hip.so: the accessors) and gsr::k_preprocess (libgsr_hip.so: the bound rasterizer entry, which
// evaluates it in place and never materialises the world-space tensors).  Every multiply-add is spelled out (fmaf or a lone
// operation), so both translation units produce the SAME bits whatever their -ffp-contract setting.
#pragma once
#include <hip/hip_runtime.h>

namespace bindm {

__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// exp(x) from fmaf / rint / ldexp only (the rasterizer's gsr_expf, ~1 ulp): the library expf is expanded by the compiler under each
// translation unit's own floating-point flags and does not give the same bits in both libraries
__device__ __forceinline__ float exp_(float x)
{
    if (!(x > -87.0f)) return 0.0f;
    if (x > 88.0f) x = 88.0f;
    const float l2e_hi = 1.44269502162933349609375f;
    const float l2e_lo = 1.92596299112661746e-8f;
    const float n = __builtin_rintf(x * l2e_hi);
    float f = fma_(x, l2e_hi, -n);
    f = fma_(x, l2e_lo, f);
    float p = 1.52527338040598402800e-5f;
    p = fma_(p, f, 1.54035303933816099544e-4f);
    p = fma_(p, f, 1.33335581464284434234e-3f);
    p = fma_(p, f, 9.61812910762847716197e-3f);
    p = fma_(p, f, 5.55041086648215799532e-2f);
    p = fma_(p, f, 2.40226506959100712334e-1f);
    p = fma_(p, f, 6.93147180559945309417e-1f);
    p = fma_(p, f, 1.0f);
    return __builtin_ldexpf(p, (int)n);
}

__device__ __forceinline__ float4 qmul(float4 a, float4 b)   // Hamilton product, WXYZ in .x.y.z.w
{
    return make_float4(fma_(-a.w, b.w, fma_(-a.z, b.z, fma_(-a.y, b.y, a.x * b.x))),
                       fma_(a.x, b.y, fma_(b.x, a.y, fma_(a.z, b.w, -(a.w * b.z)))),
                       fma_(a.x, b.z, fma_(b.x, a.z, fma_(a.w, b.y, -(a.y * b.w)))),
                       fma_(a.x, b.w, fma_(b.x, a.w, fma_(a.y, b.z, -(a.z * b.y)))));
}
__device__ __forceinline__ float4 qconj(float4 a) { return make_float4(a.x, -a.y, -a.z, -a.w); }
__device__ __forceinline__ float qnorm_clamped(float4 q)
{
    const float n = sqrtf(fma_(q.w, q.w, fma_(q.z, q.z, fma_(q.y, q.y, q.x * q.x))));
    return n > 1e-12f ? n : 1e-12f;
}
// R: the face's 3x3 row-major orientation, s its scale, c its centre
__device__ __forceinline__ void world_xyz(const float* __restrict__ R, float s, const float* __restrict__ c, float x, float y, float z, float* out)
{
#pragma unroll
    for (int r = 0; r < 3; ++r) out[r] = fma_(fma_(R[3 * r + 2], z, fma_(R[3 * r + 1], y, R[3 * r] * x)), s, c[r]);
}
__device__ __forceinline__ float world_scaling(float log_scaling, float s) { return exp_(log_scaling) * s; }
__device__ __forceinline__ float4 world_rotation(float4 face_quat, float4 local_rot)
{
    const float na = 1.f / qnorm_clamped(face_quat), nb = 1.f / qnorm_clamped(local_rot);
    const float4 a = make_float4(face_quat.x * na, face_quat.y * na, face_quat.z * na, face_quat.w * na);
    const float4 b = make_float4(local_rot.x * nb, local_rot.y * nb, local_rot.z * nb, local_rot.w * nb);
    return qmul(a, b);
}
__device__ __forceinline__ float sigmoid(float x) { return 1.f / (1.f + exp_(-x)); }
// an UNBOUND model's activations (scene/gaussian_model.py:113-160 without a binding): exp, normalize, sigmoid
__device__ __forceinline__ float4 unit_rotation(float4 q)
{
    const float nb = 1.f / qnorm_clamped(q);
    return make_float4(q.x * nb, q.y * nb, q.z * nb, q.w * nb);
}
// backward of unit_rotation: g is the gradient w.r.t. the normalised quaternion
__device__ __forceinline__ float4 unit_rotation_backward(float4 q, float4 g)
{
    const float nb = qnorm_clamped(q);
    const float4 b = make_float4(q.x / nb, q.y / nb, q.z / nb, q.w / nb);
    const float bg = b.x * g.x + b.y * g.y + b.z * g.z + b.w * g.w;
    return make_float4((g.x - b.x * bg) / nb, (g.y - b.y * bg) / nb, (g.z - b.z * bg) / nb, (g.w - b.w * bg) / nb);
}

// ---- backward of the four, for one splat.  g_*: gradients w.r.t. the world-space values; outputs: gradients of the splat's own
// leaves and `row`, its 17 contributions to its face's gradients (d_center 3 | d_orien_mat 9 | d_scaling 1 | d_orien_quat 4):
// the layout gab's per-face reduction (k_bind_bwd_faces) sums.
#define BINDM_ROW 20   // floats per row: 17 used, padded to a multiple of 16 bytes (= GAB_BIND_ROW_FLOATS)
__device__ __forceinline__ void bind_backward(const float* __restrict__ R, float s, float4 qf, const float* x /*3 local*/, const float* ls /*3 log scales*/,
                                              float4 q /*local rot*/, const float* gx /*3*/, const float* gs /*3*/, float4 g /*rot*/,
                                              float* d_xyz /*3*/, float* d_log_scaling /*3*/, float4* d_rotation, float* acc /*BINDM_ROW*/)
{
#pragma unroll
    for (int k = 0; k < BINDM_ROW; ++k) acc[k] = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) d_xyz[c] = s * (R[c] * gx[0] + R[3 + c] * gx[1] + R[6 + c] * gx[2]);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        acc[r] = gx[r];
        acc[12] += gx[r] * (R[3 * r] * x[0] + R[3 * r + 1] * x[1] + R[3 * r + 2] * x[2]);
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[3 + 3 * r + c] = s * gx[r] * x[c];
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float e = exp_(ls[k]);
        d_log_scaling[k] = gs[k] * e * s;
        acc[12] += gs[k] * e;
    }
    const float na = qnorm_clamped(qf);
    const float4 a = make_float4(qf.x / na, qf.y / na, qf.z / na, qf.w / na);
    const float nb = qnorm_clamped(q);
    const float4 b = make_float4(q.x / nb, q.y / nb, q.z / nb, q.w / nb);
    const float4 da = qmul(g, qconj(b));   // <g, a*b> = <g*conj(b), a>
    const float4 db = qmul(qconj(a), g);   //           = <conj(a)*g, b>
    const float ada = a.x * da.x + a.y * da.y + a.z * da.z + a.w * da.w;
    const float bdb = b.x * db.x + b.y * db.y + b.z * db.z + b.w * db.w;
    *d_rotation = make_float4((db.x - b.x * bdb) / nb, (db.y - b.y * bdb) / nb, (db.z - b.z * bdb) / nb, (db.w - b.w * bdb) / nb);
    acc[13] = (da.x - a.x * ada) / na;
    acc[14] = (da.y - a.y * ada) / na;
    acc[15] = (da.z - a.z * ada) / na;
    acc[16] = (da.w - a.w * ada) / na;
}

}  // namespace bindm
