/*
 * gsr_oracle.c -- CPU restatement of the tile-based differentiable Gaussian-splat
 * rasterizer that GaussianAvatars calls through `diff_gaussian_rasterization`
 * (reference call site: gaussian_renderer/__init__.py:15,37-52,86-94).
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product path is the HIP library
 * under gaussianavatars_amd/csrc and never links, includes or calls anything here.
 *
 * PARITY STATUS: **parity unpinned** against the CUDA rasterizer.  Its source
 * (submodules/diff-gaussian-rasterization, .gitmodules:4-6) is an empty un-vendored
 * submodule in /root/reference, the reference ships no tests or golden vectors, and no
 * NVIDIA device exists here.  What IS pinned (tests/test_oracle_pins.py + tests/golden):
 *   - SH basis / constants / sign pattern  vs utils/sh_utils.py:26-112 (eval_sh)
 *   - quaternion->R and the 6-float covariance packing vs utils/general_utils.py:64-110
 *   - camera / projection conventions vs utils/graphics_utils.py:38-71,
 *     utils/viewer_utils.py:20-70,142-170, scene/cameras.py:44-47
 *   - the analytic backward vs autograd of an independent fp64 torch restatement
 *     (oracle/torch_ref.py) and finite differences.
 * Everything else follows SURVEY.md Appendix A (a behavioural restatement of the classic
 * 12-field-settings / (color, radii) generation of the upstream rasterizer).
 *
 * Numerics contract (shared with the HIP kernels, restated independently there):
 *   fp32 everywhere, no FMA contraction (-ffp-contract=off), expressions evaluated
 *   left-to-right exactly as written, IEEE-correct division and sqrtf, and exp()
 *   replaced by ora_expf() below (pure fp32 fmaf/ldexp arithmetic) so that a CPU and a
 *   GPU produce the same bits.  The one double-precision spot is ndc2pix (upstream writes
 *   it with double literals).
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <omp.h>

#define ORA_BLOCK_X 16
#define ORA_BLOCK_Y 16
#define ORA_BLOCK_SIZE (ORA_BLOCK_X * ORA_BLOCK_Y)

typedef struct {
    int32_t image_height;
    int32_t image_width;
    float tanfovx;
    float tanfovy;
    float bg[3];
    float scale_modifier;
    float viewmatrix[16]; /* row-major storage of (W2C)^T: m[0],m[4],m[8],m[12] is the x row */
    float projmatrix[16]; /* row-major storage of (P*W2C)^T */
    int32_t sh_degree;
    float campos[3];
    int32_t prefiltered;
    int32_t debug;
    int32_t exact_scale_grad; /* 0 (default): upstream's dL/dscale, i.e. the gradient w.r.t. (scale_modifier * scale);
                                 1: the chain rule through the modifier as well (x scale_modifier) */
} OraSettings;

/* ---- SH constants: values of utils/sh_utils.py:26-43 rounded to fp32 ------------------ */
static const float SH_C0 = 0.28209479177387814f;
static const float SH_C1 = 0.4886025119029199f;
static const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                               -1.0925484305920792f, 0.5462742152960396f};
static const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                               0.3731763325901154f,  -0.4570457994644658f, 1.445305721320277f,
                               -0.5900435899266435f};

/* ---- reproducible expf -------------------------------------------------------------- */
/* exp(x) = 2^(x*log2e).  n = rint(x*log2e); f = x*log2e - n evaluated with a hi/lo split
 * of log2e through two fmaf; 2^f by a degree-7 Taylor polynomial in f*ln2 folded into the
 * coefficients; scaled by 2^n with ldexpf.  Inputs below -87 return 0.  ~1 ulp. */
float ora_expf(float x)
{
    const float LOG2E = 1.44269502162933349609375f;      /* fp32(log2 e) */
    const float LOG2E_LO = 1.92596299112661746e-8f;      /* log2 e - fp32(log2 e) */
    if (!(x > -87.0f)) return 0.0f;
    if (x > 88.0f) x = 88.0f;
    float t = x * LOG2E;
    float n = rintf(t);
    float f = fmaf(x, LOG2E, -n);
    f = fmaf(x, LOG2E_LO, f);
    /* c_k = ln(2)^k / k! */
    float p = 1.52527338040598402800e-5f;            /* k=7 */
    p = fmaf(p, f, 1.54035303933816099544e-4f);      /* k=6 */
    p = fmaf(p, f, 1.33335581464284434234e-3f);      /* k=5 */
    p = fmaf(p, f, 9.61812910762847716197e-3f);      /* k=4 */
    p = fmaf(p, f, 5.55041086648215799532e-2f);      /* k=3 */
    p = fmaf(p, f, 2.40226506959100712334e-1f);      /* k=2 */
    p = fmaf(p, f, 6.93147180559945309417e-1f);      /* k=1 */
    p = fmaf(p, f, 1.0f);
    return ldexpf(p, (int)n);
}

/* float -> int with CUDA/AMD-style saturation (C's cast is UB out of range) */
static int f2i_sat(float v)
{
    if (v != v) return 0;
    if (v >= 2147483648.0f) return 2147483647;
    if (v <= -2147483648.0f) return (-2147483647 - 1);
    return (int)v;
}
static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }
static float fminf_(float a, float b) { return a < b ? a : b; }
static float fmaxf_(float a, float b) { return a > b ? a : b; }

/* upstream writes ndc2Pix with double literals: evaluated in fp64, rounded once */
static float ndc2pix(float v, int S) { return (float)((((double)v + 1.0) * (double)S - 1.0) * 0.5); }

static void get_rect(float px, float py, int max_radius, int gx, int gy, int* rmin, int* rmax)
{
    float r = (float)max_radius;
    rmin[0] = imin(gx, imax(0, f2i_sat((px - r) / (float)ORA_BLOCK_X)));
    rmin[1] = imin(gy, imax(0, f2i_sat((py - r) / (float)ORA_BLOCK_Y)));
    rmax[0] = imin(gx, imax(0, f2i_sat((px + r + (float)ORA_BLOCK_X - 1.0f) / (float)ORA_BLOCK_X)));
    rmax[1] = imin(gy, imax(0, f2i_sat((py + r + (float)ORA_BLOCK_Y - 1.0f) / (float)ORA_BLOCK_Y)));
}

/* M*[p,1] with the transposed-storage convention, 3 rows */
static void xform4x3(const float* p, const float* m, float* o)
{
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
}
static void xform4x4(const float* p, const float* m, float* o)
{
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
    o[3] = m[3] * p[0] + m[7] * p[1] + m[11] * p[2] + m[15];
}

/* quaternion (r,x,y,z) used as-is (NOT normalised) -> R, entries as utils/general_utils.py:90-98 */
static void quat_to_R(const float* q, float R[3][3])
{
    float r = q[0], x = q[1], y = q[2], z = q[3];
    R[0][0] = 1.f - 2.f * (y * y + z * z);
    R[0][1] = 2.f * (x * y - r * z);
    R[0][2] = 2.f * (x * z + r * y);
    R[1][0] = 2.f * (x * y + r * z);
    R[1][1] = 1.f - 2.f * (x * x + z * z);
    R[1][2] = 2.f * (y * z - r * x);
    R[2][0] = 2.f * (x * z - r * y);
    R[2][1] = 2.f * (y * z + r * x);
    R[2][2] = 1.f - 2.f * (x * x + y * y);
}

/* Sigma = R diag(s^2) R^T computed as M^T M with M[k][i] = s_k * R[i][k]; packed xx,xy,xz,yy,yz,zz */
static void compute_cov3d(const float* scale, float mod, const float* rot, float* cov6)
{
    float R[3][3];
    quat_to_R(rot, R);
    float s[3] = {mod * scale[0], mod * scale[1], mod * scale[2]};
    float M[3][3]; /* M[k][i] */
    for (int k = 0; k < 3; ++k)
        for (int i = 0; i < 3; ++i) M[k][i] = s[k] * R[i][k];
    float S[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) S[i][j] = M[0][i] * M[0][j] + M[1][i] * M[1][j] + M[2][i] * M[2][j];
    cov6[0] = S[0][0];
    cov6[1] = S[0][1];
    cov6[2] = S[0][2];
    cov6[3] = S[1][1];
    cov6[4] = S[1][2];
    cov6[5] = S[2][2];
}

/* A = J * Rwc (2x3), the EWA Jacobian chain; also returns clamped t and the clamp flags */
static void ewa_A(const float* mean, const float* vm, float fx, float fy, float tanfovx, float tanfovy,
                  float A[2][3], float* t_out, int* xin, int* yin)
{
    float t[3];
    xform4x3(mean, vm, t);
    const float limx = 1.3f * tanfovx;
    const float limy = 1.3f * tanfovy;
    const float txtz = t[0] / t[2];
    const float tytz = t[1] / t[2];
    t[0] = fminf_(limx, fmaxf_(-limx, txtz)) * t[2];
    t[1] = fminf_(limy, fmaxf_(-limy, tytz)) * t[2];
    if (xin) *xin = !(txtz < -limx || txtz > limx);
    if (yin) *yin = !(tytz < -limy || tytz > limy);
    const float J00 = fx / t[2];
    const float J02 = -(fx * t[0]) / (t[2] * t[2]);
    const float J11 = fy / t[2];
    const float J12 = -(fy * t[1]) / (t[2] * t[2]);
    /* Rwc[i][j] = vm[4*j + i] */
    for (int j = 0; j < 3; ++j) {
        const float r0 = vm[4 * j + 0], r1 = vm[4 * j + 1], r2 = vm[4 * j + 2];
        A[0][j] = J00 * r0 + 0.0f * r1 + J02 * r2;
        A[1][j] = 0.0f * r0 + J11 * r1 + J12 * r2;
    }
    if (t_out) { t_out[0] = t[0]; t_out[1] = t[1]; t_out[2] = t[2]; }
}

static void compute_cov2d(const float* mean, float fx, float fy, float tanfovx, float tanfovy,
                          const float* c6, const float* vm, float* cov3)
{
    float A[2][3];
    ewa_A(mean, vm, fx, fy, tanfovx, tanfovy, A, 0, 0, 0);
    const float V[3][3] = {{c6[0], c6[1], c6[2]}, {c6[1], c6[3], c6[4]}, {c6[2], c6[4], c6[5]}};
    float AV[2][3];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 3; ++j) AV[i][j] = A[i][0] * V[0][j] + A[i][1] * V[1][j] + A[i][2] * V[2][j];
    float c00 = AV[0][0] * A[0][0] + AV[0][1] * A[0][1] + AV[0][2] * A[0][2];
    float c01 = AV[0][0] * A[1][0] + AV[0][1] * A[1][1] + AV[0][2] * A[1][2];
    float c11 = AV[1][0] * A[1][0] + AV[1][1] * A[1][1] + AV[1][2] * A[1][2];
    cov3[0] = c00 + 0.3f;
    cov3[1] = c01;
    cov3[2] = c11 + 0.3f;
}

/* SH (degree <= 3) -> rgb, utils/sh_utils.py:74-100 polynomial; returns clamp flags */
static void sh_to_rgb(int deg, int M, const float* mean, const float* campos, const float* sh /* M x 3 */,
                      float* rgb, uint8_t* clamped)
{
    float d[3] = {mean[0] - campos[0], mean[1] - campos[1], mean[2] - campos[2]};
    float len = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    float x = d[0] / len, y = d[1] / len, z = d[2] / len;
    (void)M;
    for (int c = 0; c < 3; ++c) {
#define SH(k) sh[3 * (k) + c]
        float res = SH_C0 * SH(0);
        if (deg > 0) {
            res = res - SH_C1 * y * SH(1) + SH_C1 * z * SH(2) - SH_C1 * x * SH(3);
            if (deg > 1) {
                float xx = x * x, yy = y * y, zz = z * z;
                float xy = x * y, yz = y * z, xz = x * z;
                res = res + SH_C2[0] * xy * SH(4) + SH_C2[1] * yz * SH(5) + SH_C2[2] * (2.0f * zz - xx - yy) * SH(6) +
                      SH_C2[3] * xz * SH(7) + SH_C2[4] * (xx - yy) * SH(8);
                if (deg > 2) {
                    res = res + SH_C3[0] * y * (3.0f * xx - yy) * SH(9) + SH_C3[1] * xy * z * SH(10) +
                          SH_C3[2] * y * (4.0f * zz - xx - yy) * SH(11) +
                          SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * SH(12) +
                          SH_C3[4] * x * (4.0f * zz - xx - yy) * SH(13) + SH_C3[5] * z * (xx - yy) * SH(14) +
                          SH_C3[6] * x * (xx - 3.0f * yy) * SH(15);
                }
            }
        }
#undef SH
        res += 0.5f;
        clamped[c] = (uint8_t)(res < 0.0f);
        rgb[c] = fmaxf_(res, 0.0f);
    }
}

/* =====================================================================================
 * A.1  preprocess (per splat).  All outputs are caller-allocated.
 *   depths[P] xy[2P] cov3D[6P] conic_opacity[4P] rgb[3P] radii[P] tiles_touched[P]
 *   rect[4P] = (min.x, min.y, max.x, max.y), clamped[3P]
 * ===================================================================================== */
void ora_preprocess(const OraSettings* s, int P, int M, const float* means3D, const float* shs,
                    const float* colors_precomp, const float* opacities, const float* scales,
                    const float* rotations, const float* cov3D_precomp, float* depths, float* xy,
                    float* cov3D, float* conic_opacity, float* rgb, int32_t* radii, uint32_t* tiles_touched,
                    int32_t* rect, uint8_t* clamped)
{
    const int W = s->image_width, H = s->image_height;
    const int gx = (W + ORA_BLOCK_X - 1) / ORA_BLOCK_X, gy = (H + ORA_BLOCK_Y - 1) / ORA_BLOCK_Y;
    const float fx = (float)W / (2.0f * s->tanfovx);
    const float fy = (float)H / (2.0f * s->tanfovy);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < P; ++i) {
        radii[i] = 0;
        tiles_touched[i] = 0;
        depths[i] = 0.f;
        xy[2 * i] = xy[2 * i + 1] = 0.f;
        for (int k = 0; k < 6; ++k) cov3D[6 * i + k] = 0.f;
        for (int k = 0; k < 4; ++k) conic_opacity[4 * i + k] = 0.f;
        for (int k = 0; k < 3; ++k) { rgb[3 * i + k] = 0.f; clamped[3 * i + k] = 0; }
        for (int k = 0; k < 4; ++k) rect[4 * i + k] = 0;

        const float* p = means3D + 3 * i;
        float pv[3];
        xform4x3(p, s->viewmatrix, pv);
        if (pv[2] <= 0.2f) continue;
        float ph[4];
        xform4x4(p, s->projmatrix, ph);
        float pw = 1.0f / (ph[3] + 0.0000001f);
        float pp[3] = {ph[0] * pw, ph[1] * pw, ph[2] * pw};

        float c6[6];
        if (cov3D_precomp) {
            for (int k = 0; k < 6; ++k) c6[k] = cov3D_precomp[6 * i + k];
        } else {
            compute_cov3d(scales + 3 * i, s->scale_modifier, rotations + 4 * i, c6);
        }
        float cov[3];
        compute_cov2d(p, fx, fy, s->tanfovx, s->tanfovy, c6, s->viewmatrix, cov);
        float det = cov[0] * cov[2] - cov[1] * cov[1];
        if (det == 0.0f) continue;
        float det_inv = 1.f / det;
        float conic[3] = {cov[2] * det_inv, -cov[1] * det_inv, cov[0] * det_inv};
        float mid = 0.5f * (cov[0] + cov[2]);
        float lambda1 = mid + sqrtf(fmaxf_(0.1f, mid * mid - det));
        float lambda2 = mid - sqrtf(fmaxf_(0.1f, mid * mid - det));
        float my_radius = ceilf(3.f * sqrtf(fmaxf_(lambda1, lambda2)));
        float px = ndc2pix(pp[0], W), py = ndc2pix(pp[1], H);
        int rmin[2], rmax[2];
        get_rect(px, py, f2i_sat(my_radius), gx, gy, rmin, rmax);
        if ((rmax[0] - rmin[0]) * (rmax[1] - rmin[1]) == 0) continue;

        if (colors_precomp) {
            for (int k = 0; k < 3; ++k) rgb[3 * i + k] = colors_precomp[3 * i + k];
        } else {
            sh_to_rgb(s->sh_degree, M, p, s->campos, shs + (size_t)3 * M * i, rgb + 3 * i, clamped + 3 * i);
        }
        for (int k = 0; k < 6; ++k) cov3D[6 * i + k] = c6[k];
        depths[i] = pv[2];
        radii[i] = f2i_sat(my_radius);
        xy[2 * i] = px;
        xy[2 * i + 1] = py;
        conic_opacity[4 * i + 0] = conic[0];
        conic_opacity[4 * i + 1] = conic[1];
        conic_opacity[4 * i + 2] = conic[2];
        conic_opacity[4 * i + 3] = opacities[i];
        tiles_touched[i] = (uint32_t)((rmax[1] - rmin[1]) * (rmax[0] - rmin[0]));
        rect[4 * i + 0] = rmin[0];
        rect[4 * i + 1] = rmin[1];
        rect[4 * i + 2] = rmax[0];
        rect[4 * i + 3] = rmax[1];
    }
}

/* =====================================================================================
 * A.2  binning: inclusive scan -> duplicate with keys -> stable sort -> tile ranges
 * ===================================================================================== */
int64_t ora_scan(int P, const uint32_t* tiles_touched, uint32_t* offsets)
{
    uint32_t acc = 0;
    for (int i = 0; i < P; ++i) { acc += tiles_touched[i]; offsets[i] = acc; }
    return P > 0 ? (int64_t)offsets[P - 1] : 0;
}

/* Stable sort of the (key, value) pairs by key = tile << 32 | depth bits -- what one stable radix sort over the low
 * 32 + bits(tiles) key bits produces (A.2) -- organised so that the host cores can share it: one stable counting-sort pass
 * by tile (stable_group below; the order inside a tile stays the emission order, i.e. ascending splat index), then every tile's segment is
 * sorted on its own by the 32 depth bits with a stable LSD radix (8 bits a pass; short segments by stable insertion), OpenMP
 * over tiles. */
static void sort_segment(uint64_t* k, uint32_t* v, uint64_t* tk, uint32_t* tv, int64_t n)
{
    if (n < 48) {
        for (int64_t i = 1; i < n; ++i) {
            const uint64_t kk = k[i];
            const uint32_t vv = v[i];
            int64_t j = i - 1;
            while (j >= 0 && (uint32_t)k[j] > (uint32_t)kk) { k[j + 1] = k[j]; v[j + 1] = v[j]; --j; }
            k[j + 1] = kk;
            v[j + 1] = vv;
        }
        return;
    }
    uint64_t *sk = k, *dk = tk;
    uint32_t *sv = v, *dv = tv;
    for (int shift = 0; shift < 32; shift += 8) {
        int64_t cnt[257];
        memset(cnt, 0, sizeof(cnt));
        for (int64_t i = 0; i < n; ++i) cnt[((sk[i] >> shift) & 0xFF) + 1]++;
        for (int d = 0; d < 256; ++d) cnt[d + 1] += cnt[d];
        for (int64_t i = 0; i < n; ++i) {
            const int64_t dst = cnt[(sk[i] >> shift) & 0xFF]++;
            dk[dst] = sk[i];
            dv[dst] = sv[i];
        }
        uint64_t* t1 = sk; sk = dk; dk = t1;
        uint32_t* t2 = sv; sv = dv; dv = t2;
    }
    /* four passes: the result is back in (k, v) */
}

/* perm[0..n) = the indices 0..n-1 grouped by bin[i] (ascending index inside a bin), start[0..nbins] = the groups' first positions:
 * a STABLE counting sort whose two passes the host cores share -- the index range is cut into chunks, every chunk histograms its own
 * stretch, a prefix over (bin, chunk) gives every chunk its private cursor per bin, and the chunks scatter in order.  The result does
 * not depend on the number of chunks or threads. */
static void stable_group(const uint32_t* bin, uint32_t shift_is_key_hi, const uint64_t* key64, int64_t n, int64_t nbins, int64_t* start,
                         int64_t* perm)
{
    int chunks = omp_get_max_threads();
    if (chunks > 64) chunks = 64;
    while (chunks > 1 && (int64_t)chunks * nbins > (int64_t)(1 << 25)) chunks /= 2;
    if (n < 4096) chunks = 1;
    int64_t* hist = (int64_t*)calloc((size_t)chunks * (size_t)nbins + 1, sizeof(int64_t));
    const int64_t per = (n + chunks - 1) / chunks;
#define BIN_OF(i) (shift_is_key_hi ? (int64_t)(key64[i] >> 32) : (int64_t)bin[i])
#pragma omp parallel for schedule(static, 1)
    for (int c = 0; c < chunks; ++c) {
        int64_t* h = hist + (size_t)c * (size_t)nbins;
        const int64_t lo = c * per, hi = lo + per < n ? lo + per : n;
        for (int64_t i = lo; i < hi; ++i) h[BIN_OF(i)]++;
    }
    int64_t run = 0;
    for (int64_t b = 0; b < nbins; ++b) {
        start[b] = run;
        for (int c = 0; c < chunks; ++c) {
            int64_t* h = hist + (size_t)c * (size_t)nbins + b;
            const int64_t cnt = *h;
            *h = run;
            run += cnt;
        }
    }
    start[nbins] = run;
#pragma omp parallel for schedule(static, 1)
    for (int c = 0; c < chunks; ++c) {
        int64_t* h = hist + (size_t)c * (size_t)nbins;
        const int64_t lo = c * per, hi = lo + per < n ? lo + per : n;
        for (int64_t i = lo; i < hi; ++i) perm[h[BIN_OF(i)]++] = i;
    }
#undef BIN_OF
    free(hist);
}

static void sort_pairs_by_tile_then_depth(uint64_t* keys, uint32_t* vals, uint64_t* tk, uint32_t* tv, int64_t n, int tiles)
{
    int64_t* start = (int64_t*)calloc((size_t)tiles + 1, sizeof(int64_t));
    int64_t* perm = (int64_t*)malloc((size_t)(n > 0 ? n : 1) * sizeof(int64_t));
    stable_group(NULL, 1u, keys, n, tiles, start, perm);   /* by tile; inside a tile the emission order (ascending splat index) stays */
#pragma omp parallel for schedule(static)
    for (int64_t d = 0; d < n; ++d) { tk[d] = keys[perm[d]]; tv[d] = vals[perm[d]]; }
#pragma omp parallel for schedule(static)
    for (int64_t d = 0; d < n; ++d) { keys[d] = tk[d]; vals[d] = tv[d]; }
#pragma omp parallel for schedule(dynamic, 8)
    for (int t = 0; t < tiles; ++t)
        sort_segment(keys + start[t], vals + start[t], tk + start[t], tv + start[t], start[t + 1] - start[t]);
    free(perm);
    free(start);
}

/* keys[I], vals[I] come back sorted; ranges[2*tiles] */
void ora_bin(const OraSettings* s, int P, const float* depths, const int32_t* radii, const int32_t* rect,
             const uint32_t* offsets, int64_t I, uint64_t* keys, uint32_t* vals, uint32_t* ranges)
{
    const int W = s->image_width, H = s->image_height;
    const int gx = (W + ORA_BLOCK_X - 1) / ORA_BLOCK_X, gy = (H + ORA_BLOCK_Y - 1) / ORA_BLOCK_Y;
#pragma omp parallel for schedule(dynamic, 256)   /* every splat writes its own [offsets[i-1], offsets[i]) */
    for (int i = 0; i < P; ++i) {
        if (radii[i] > 0) {
            uint32_t off = (i == 0) ? 0 : offsets[i - 1];
            uint32_t dbits;
            memcpy(&dbits, depths + i, 4);
            for (int y = rect[4 * i + 1]; y < rect[4 * i + 3]; ++y)
                for (int x = rect[4 * i + 0]; x < rect[4 * i + 2]; ++x) {
                    uint64_t key = (uint64_t)(uint32_t)(y * gx + x);
                    key <<= 32;
                    key |= dbits;
                    keys[off] = key;
                    vals[off] = (uint32_t)i;
                    off++;
                }
        }
    }
    int tiles = gx * gy;
    uint64_t* tk = (uint64_t*)malloc((size_t)(I > 0 ? I : 1) * sizeof(uint64_t));
    uint32_t* tv = (uint32_t*)malloc((size_t)(I > 0 ? I : 1) * sizeof(uint32_t));
    sort_pairs_by_tile_then_depth(keys, vals, tk, tv, I, tiles);
    free(tk);
    free(tv);
    memset(ranges, 0, (size_t)tiles * 2 * sizeof(uint32_t));
    for (int64_t idx = 0; idx < I; ++idx) {
        uint32_t cur = (uint32_t)(keys[idx] >> 32);
        if (idx == 0)
            ranges[2 * cur + 0] = 0;
        else {
            uint32_t prev = (uint32_t)(keys[idx - 1] >> 32);
            if (cur != prev) {
                ranges[2 * prev + 1] = (uint32_t)idx;
                ranges[2 * cur + 0] = (uint32_t)idx;
            }
        }
        if (idx == I - 1) ranges[2 * cur + 1] = (uint32_t)I;
    }
}

/* =====================================================================================
 * A.3  forward blend, per pixel
 * ===================================================================================== */
void ora_render_forward(const OraSettings* s, const uint32_t* ranges, const uint32_t* point_list,
                        const float* xy, const float* rgb, const float* conic_opacity, float* final_T,
                        uint32_t* n_contrib, float* out_color)
{
    const int W = s->image_width, H = s->image_height;
    const int gx = (W + ORA_BLOCK_X - 1) / ORA_BLOCK_X;
#pragma omp parallel for schedule(dynamic, 4)
    for (int py = 0; py < H; ++py) {
        for (int px = 0; px < W; ++px) {
            const int tile = (py / ORA_BLOCK_Y) * gx + (px / ORA_BLOCK_X);
            const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
            const float pixx = (float)px, pixy = (float)py;
            float T = 1.0f;
            uint32_t contributor = 0, last_contributor = 0;
            float C[3] = {0.f, 0.f, 0.f};
            for (uint32_t k = r0; k < r1; ++k) {
                contributor++;
                const uint32_t id = point_list[k];
                const float dx = xy[2 * id] - pixx;
                const float dy = xy[2 * id + 1] - pixy;
                const float* co = conic_opacity + 4 * id;
                const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                if (power > 0.0f) continue;
                const float alpha = fminf_(0.99f, co[3] * ora_expf(power));
                if (alpha < 1.0f / 255.0f) continue;
                const float test_T = T * (1.0f - alpha);
                if (test_T < 0.0001f) break;
                for (int ch = 0; ch < 3; ++ch) C[ch] += rgb[3 * id + ch] * alpha * T;
                T = test_T;
                last_contributor = contributor;
            }
            const int pix_id = W * py + px;
            final_T[pix_id] = T;
            n_contrib[pix_id] = last_contributor;
            for (int ch = 0; ch < 3; ++ch) out_color[ch * H * W + pix_id] = C[ch] + T * s->bg[ch];
        }
    }
}

/* =====================================================================================
 * A.4  backward.  Accumulators are double on purpose: the oracle is the *reference value*
 * of the sum, the fp32 device result must land within tolerance of it.
 * Parallel over tiles (OpenMP) and still deterministic: a tile owns the instances [r0, r1) of
 * the sorted list, so every (pixel, instance) term is added -- pixels row-major inside the
 * tile, back to front per pixel -- into a per-INSTANCE partial row nobody else touches; a
 * second, sequential pass folds the instance rows into the per-splat sums in list order.
 * The summation order is therefore fixed whatever the thread count.
 * ===================================================================================== */
#ifdef _OPENMP
#include <omp.h>
#endif
/* threads used by the OpenMP loops of this file from now on (0 = leave the runtime's default); returns the count that will
 * be used.  bench.py's cpu_baseline reports a 1-thread and an all-core figure with it. */
int ora_set_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
    return omp_get_max_threads();
#else
    (void)n;
    return 1;
#endif
}

void ora_render_backward(const OraSettings* s, int P, const uint32_t* ranges, const uint32_t* point_list,
                         const float* xy, const float* conic_opacity, const float* rgb, const float* final_T,
                         const uint32_t* n_contrib, const float* dL_dpix, double* dL_dmean2D /*2P*/,
                         double* dL_dconic /*3P: A,B,C*/, double* dL_dopacity /*P*/, double* dL_dcolor /*3P*/)
{
    const int W = s->image_width, H = s->image_height;
    const int gx = (W + ORA_BLOCK_X - 1) / ORA_BLOCK_X, gy = (H + ORA_BLOCK_Y - 1) / ORA_BLOCK_Y;
    memset(dL_dmean2D, 0, sizeof(double) * 2 * (size_t)P);
    memset(dL_dconic, 0, sizeof(double) * 3 * (size_t)P);
    memset(dL_dopacity, 0, sizeof(double) * (size_t)P);
    memset(dL_dcolor, 0, sizeof(double) * 3 * (size_t)P);
    const float ddelx_dx = 0.5f * (float)W;
    const float ddely_dy = 0.5f * (float)H;
    size_t I = 0;   /* instances = end of the last non-empty range */
    for (int t = 0; t < gx * gy; ++t)
        if (ranges[2 * t + 1] > I) I = ranges[2 * t + 1];
    if (I == 0) return;
    double* part = (double*)malloc((I > 0 ? I : 1) * 9 * sizeof(double));   /* per instance: mean2D 2, conic 3, opacity 1, colour 3 */
    if (!part) return;
#pragma omp parallel for schedule(static)   /* zeroed (and first touched) by every core: 166 MB at 100 k splats */
    for (int64_t k = 0; k < (int64_t)I; ++k)
        for (int c = 0; c < 9; ++c) part[9 * k + c] = 0.0;
#pragma omp parallel for schedule(dynamic, 1)
    for (int tile = 0; tile < gx * gy; ++tile) {
        const uint32_t r0 = ranges[2 * tile];
        if (ranges[2 * tile + 1] == r0) continue;
        const int ty = tile / gx, tx = tile - ty * gx;
        const int y1 = imin(H, (ty + 1) * ORA_BLOCK_Y), x1 = imin(W, (tx + 1) * ORA_BLOCK_X);
        for (int py = ty * ORA_BLOCK_Y; py < y1; ++py) {
            for (int px = tx * ORA_BLOCK_X; px < x1; ++px) {
                const int pix_id = W * py + px;
                const float pixx = (float)px, pixy = (float)py;
                const float T_final = final_T[pix_id];
                float T = T_final;
                const uint32_t last = n_contrib[pix_id];
                float accum_rec[3] = {0, 0, 0}, last_color[3] = {0, 0, 0}, last_alpha = 0.f;
                float dpix[3];
                for (int ch = 0; ch < 3; ++ch) dpix[ch] = dL_dpix[ch * H * W + pix_id];
                float bg_dot = 0.f;
                for (int ch = 0; ch < 3; ++ch) bg_dot += s->bg[ch] * dpix[ch];
                for (uint32_t c = last; c-- > 0;) {
                    const uint32_t id = point_list[r0 + c];
                    double* acc = part + 9 * (size_t)(r0 + c);
                    const float dx = xy[2 * id] - pixx;
                    const float dy = xy[2 * id + 1] - pixy;
                    const float* co = conic_opacity + 4 * id;
                    const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                    if (power > 0.0f) continue;
                    const float G = ora_expf(power);
                    const float alpha = fminf_(0.99f, co[3] * G);
                    if (alpha < 1.0f / 255.0f) continue;
                    T = T / (1.f - alpha);
                    const float dchannel_dcolor = alpha * T;
                    float dL_dalpha = 0.0f;
                    for (int ch = 0; ch < 3; ++ch) {
                        const float col = rgb[3 * id + ch];
                        accum_rec[ch] = last_alpha * last_color[ch] + (1.f - last_alpha) * accum_rec[ch];
                        last_color[ch] = col;
                        dL_dalpha += (col - accum_rec[ch]) * dpix[ch];
                        acc[6 + ch] += (double)(dchannel_dcolor * dpix[ch]);
                    }
                    dL_dalpha *= T;
                    last_alpha = alpha;
                    dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot;
                    const float dL_dG = co[3] * dL_dalpha;
                    const float gdx = G * dx, gdy = G * dy;
                    const float dG_ddelx = -gdx * co[0] - gdy * co[1];
                    const float dG_ddely = -gdy * co[2] - gdx * co[1];
                    acc[0] += (double)(dL_dG * dG_ddelx * ddelx_dx);
                    acc[1] += (double)(dL_dG * dG_ddely * ddely_dy);
                    acc[2] += (double)(-0.5f * gdx * dx * dL_dG);
                    acc[3] += (double)(-0.5f * gdx * dy * dL_dG);
                    acc[4] += (double)(-0.5f * gdy * dy * dL_dG);
                    acc[5] += (double)(G * dL_dalpha);
                }
            }
        }
    }
    /* fold per splat, in LIST order (ascending position): the sums are the same whatever the thread count.  The positions of every
     * splat come from a stable grouping of the list by splat index (the host cores share it), then the splats are independent. */
    {
        int64_t* start = (int64_t*)calloc((size_t)P + 1, sizeof(int64_t));
        int64_t* perm = (int64_t*)malloc((I > 0 ? I : 1) * sizeof(int64_t));
        stable_group(point_list, 0u, NULL, (int64_t)I, (int64_t)P, start, perm);
#pragma omp parallel for schedule(dynamic, 512)
        for (int id = 0; id < P; ++id) {
            for (int64_t d = start[id]; d < start[id + 1]; ++d) {
                const double* acc = part + 9 * (size_t)perm[d];
                dL_dmean2D[2 * id + 0] += acc[0];
                dL_dmean2D[2 * id + 1] += acc[1];
                dL_dconic[3 * id + 0] += acc[2];
                dL_dconic[3 * id + 1] += acc[3];
                dL_dconic[3 * id + 2] += acc[4];
                dL_dopacity[id] += acc[5];
                dL_dcolor[3 * id + 0] += acc[6];
                dL_dcolor[3 * id + 1] += acc[7];
                dL_dcolor[3 * id + 2] += acc[8];
            }
        }
        free(perm);
        free(start);
    }
    free(part);
}

/* K8 + K9: per visible splat.  Inputs are fp32 (the oracle driver rounds the double sums of
 * ora_render_backward to fp32 first, mirroring what the device kernel consumes). */
void ora_preprocess_backward(const OraSettings* s, int P, int M, const float* means3D, const float* shs,
                             const float* scales, const float* rotations, const float* cov3D /*6P, as used fwd*/,
                             int use_precomp_cov, int use_precomp_color, const int32_t* radii,
                             const uint8_t* clamped, const float* dL_dmean2D /*2P*/, const float* dL_dconic /*3P*/,
                             const float* dL_dcolor /*3P*/, float* dL_dmeans3D /*3P*/, float* dL_dcov3D /*6P*/,
                             float* dL_dsh /*P*M*3*/, float* dL_dscale /*3P*/, float* dL_drot /*4P*/)
{
    const int W = s->image_width, H = s->image_height;
    const float fx = (float)W / (2.0f * s->tanfovx);
    const float fy = (float)H / (2.0f * s->tanfovy);
    const float* vm = s->viewmatrix;
    const float* proj = s->projmatrix;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < P; ++i) {
        for (int k = 0; k < 3; ++k) dL_dmeans3D[3 * i + k] = 0.f;
        for (int k = 0; k < 6; ++k) dL_dcov3D[6 * i + k] = 0.f;
        if (dL_dsh) for (int k = 0; k < 3 * M; ++k) dL_dsh[(size_t)3 * M * i + k] = 0.f;
        for (int k = 0; k < 3; ++k) dL_dscale[3 * i + k] = 0.f;
        for (int k = 0; k < 4; ++k) dL_drot[4 * i + k] = 0.f;
        if (!(radii[i] > 0)) continue;
        const float* mean = means3D + 3 * i;
        const float* c6 = cov3D + 6 * i;

        /* ---- K8: conic -> cov2D -> (cov3D, mean) ---- */
        float A[2][3], t[3];
        int xin, yin;
        ewa_A(mean, vm, fx, fy, s->tanfovx, s->tanfovy, A, t, &xin, &yin);
        const float V[3][3] = {{c6[0], c6[1], c6[2]}, {c6[1], c6[3], c6[4]}, {c6[2], c6[4], c6[5]}};
        float AV[2][3];
        for (int r = 0; r < 2; ++r)
            for (int j = 0; j < 3; ++j) AV[r][j] = A[r][0] * V[0][j] + A[r][1] * V[1][j] + A[r][2] * V[2][j];
        const float a = (AV[0][0] * A[0][0] + AV[0][1] * A[0][1] + AV[0][2] * A[0][2]) + 0.3f;
        const float b = AV[0][0] * A[1][0] + AV[0][1] * A[1][1] + AV[0][2] * A[1][2];
        const float c = (AV[1][0] * A[1][0] + AV[1][1] * A[1][1] + AV[1][2] * A[1][2]) + 0.3f;
        const float gA = dL_dconic[3 * i + 0], gB = dL_dconic[3 * i + 1], gC = dL_dconic[3 * i + 2];
        const float denom = a * c - b * b;
        float dL_da = 0, dL_db = 0, dL_dc = 0;
        const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
        float gcov[6] = {0, 0, 0, 0, 0, 0};
        if (denom2inv != 0) {
            dL_da = denom2inv * (-c * c * gA + 2 * b * c * gB + (denom - a * c) * gC);
            dL_dc = denom2inv * (-a * a * gC + 2 * a * b * gB + (denom - a * c) * gA);
            dL_db = denom2inv * 2 * (b * c * gA - (denom + 2 * b * b) * gB + a * b * gC);
            /* cov2D = A V A^T ; packed symmetric V: off-diagonals carry both mirror entries */
            gcov[0] = A[0][0] * A[0][0] * dL_da + A[0][0] * A[1][0] * dL_db + A[1][0] * A[1][0] * dL_dc;
            gcov[3] = A[0][1] * A[0][1] * dL_da + A[0][1] * A[1][1] * dL_db + A[1][1] * A[1][1] * dL_dc;
            gcov[5] = A[0][2] * A[0][2] * dL_da + A[0][2] * A[1][2] * dL_db + A[1][2] * A[1][2] * dL_dc;
            gcov[1] = 2 * A[0][0] * A[0][1] * dL_da + (A[0][0] * A[1][1] + A[0][1] * A[1][0]) * dL_db +
                      2 * A[1][0] * A[1][1] * dL_dc;
            gcov[2] = 2 * A[0][0] * A[0][2] * dL_da + (A[0][0] * A[1][2] + A[0][2] * A[1][0]) * dL_db +
                      2 * A[1][0] * A[1][2] * dL_dc;
            gcov[4] = 2 * A[0][2] * A[0][1] * dL_da + (A[0][1] * A[1][2] + A[0][2] * A[1][1]) * dL_db +
                      2 * A[1][1] * A[1][2] * dL_dc;
        }
        for (int k = 0; k < 6; ++k) dL_dcov3D[6 * i + k] = gcov[k];
        /* dL/dA */
        float dA[2][3];
        for (int j = 0; j < 3; ++j) {
            dA[0][j] = 2 * (A[0][0] * V[j][0] + A[0][1] * V[j][1] + A[0][2] * V[j][2]) * dL_da +
                       (A[1][0] * V[j][0] + A[1][1] * V[j][1] + A[1][2] * V[j][2]) * dL_db;
            dA[1][j] = 2 * (A[1][0] * V[j][0] + A[1][1] * V[j][1] + A[1][2] * V[j][2]) * dL_dc +
                       (A[0][0] * V[j][0] + A[0][1] * V[j][1] + A[0][2] * V[j][2]) * dL_db;
        }
        /* Rwc[r][j] = vm[4*j + r];  dL/dJ = dL/dA * Rwc^T */
        const float dJ00 = vm[0] * dA[0][0] + vm[4] * dA[0][1] + vm[8] * dA[0][2];
        const float dJ02 = vm[2] * dA[0][0] + vm[6] * dA[0][1] + vm[10] * dA[0][2];
        const float dJ11 = vm[1] * dA[1][0] + vm[5] * dA[1][1] + vm[9] * dA[1][2];
        const float dJ12 = vm[2] * dA[1][0] + vm[6] * dA[1][1] + vm[10] * dA[1][2];
        const float tz = 1.f / t[2];
        const float tz2 = tz * tz;
        const float tz3 = tz2 * tz;
        const float x_grad_mul = xin ? 1.f : 0.f;
        const float y_grad_mul = yin ? 1.f : 0.f;
        const float dtx = x_grad_mul * -fx * tz2 * dJ02;
        const float dty = y_grad_mul * -fy * tz2 * dJ12;
        const float dtz = -fx * tz2 * dJ00 - fy * tz2 * dJ11 + (2 * fx * t[0]) * tz3 * dJ02 + (2 * fy * t[1]) * tz3 * dJ12;
        float dmean[3];
        dmean[0] = vm[0] * dtx + vm[1] * dty + vm[2] * dtz;
        dmean[1] = vm[4] * dtx + vm[5] * dty + vm[6] * dtz;
        dmean[2] = vm[8] * dtx + vm[9] * dty + vm[10] * dtz;

        /* ---- K9 (i): projective divide ---- */
        float mh[4];
        xform4x4(mean, proj, mh);
        const float m_w = 1.0f / (mh[3] + 0.0000001f);
        const float mul1 = (proj[0] * mean[0] + proj[4] * mean[1] + proj[8] * mean[2] + proj[12]) * m_w * m_w;
        const float mul2 = (proj[1] * mean[0] + proj[5] * mean[1] + proj[9] * mean[2] + proj[13]) * m_w * m_w;
        const float g2x = dL_dmean2D[2 * i], g2y = dL_dmean2D[2 * i + 1];
        dmean[0] += (proj[0] * m_w - proj[3] * mul1) * g2x + (proj[1] * m_w - proj[3] * mul2) * g2y;
        dmean[1] += (proj[4] * m_w - proj[7] * mul1) * g2x + (proj[5] * m_w - proj[7] * mul2) * g2y;
        dmean[2] += (proj[8] * m_w - proj[11] * mul1) * g2x + (proj[9] * m_w - proj[11] * mul2) * g2y;

        /* ---- K9 (ii): SH ---- */
        if (!use_precomp_color) {
            const float* sh = shs + (size_t)3 * M * i;
            float* gsh = dL_dsh + (size_t)3 * M * i;
            const int deg = s->sh_degree;
            float d0[3] = {mean[0] - s->campos[0], mean[1] - s->campos[1], mean[2] - s->campos[2]};
            float len = sqrtf(d0[0] * d0[0] + d0[1] * d0[1] + d0[2] * d0[2]);
            float x = d0[0] / len, y = d0[1] / len, z = d0[2] / len;
            float g[3];
            for (int ch = 0; ch < 3; ++ch) g[ch] = clamped[3 * i + ch] ? 0.f : dL_dcolor[3 * i + ch];
            float ddir[3] = {0, 0, 0};
            for (int ch = 0; ch < 3; ++ch) {
#define SH(k) sh[3 * (k) + ch]
#define GSH(k) gsh[3 * (k) + ch]
                float dx_ = 0, dy_ = 0, dz_ = 0;
                GSH(0) = SH_C0 * g[ch];
                if (deg > 0) {
                    GSH(1) = -SH_C1 * y * g[ch];
                    GSH(2) = SH_C1 * z * g[ch];
                    GSH(3) = -SH_C1 * x * g[ch];
                    dx_ = -SH_C1 * SH(3);
                    dy_ = -SH_C1 * SH(1);
                    dz_ = SH_C1 * SH(2);
                    if (deg > 1) {
                        float xx = x * x, yy = y * y, zz = z * z, xy_ = x * y, yz = y * z, xz = x * z;
                        GSH(4) = SH_C2[0] * xy_ * g[ch];
                        GSH(5) = SH_C2[1] * yz * g[ch];
                        GSH(6) = SH_C2[2] * (2.f * zz - xx - yy) * g[ch];
                        GSH(7) = SH_C2[3] * xz * g[ch];
                        GSH(8) = SH_C2[4] * (xx - yy) * g[ch];
                        dx_ += SH_C2[0] * y * SH(4) + SH_C2[2] * 2.f * -x * SH(6) + SH_C2[3] * z * SH(7) + SH_C2[4] * 2.f * x * SH(8);
                        dy_ += SH_C2[0] * x * SH(4) + SH_C2[1] * z * SH(5) + SH_C2[2] * 2.f * -y * SH(6) + SH_C2[4] * 2.f * -y * SH(8);
                        dz_ += SH_C2[1] * y * SH(5) + SH_C2[2] * 2.f * 2.f * z * SH(6) + SH_C2[3] * x * SH(7);
                        if (deg > 2) {
                            GSH(9) = SH_C3[0] * y * (3.f * xx - yy) * g[ch];
                            GSH(10) = SH_C3[1] * xy_ * z * g[ch];
                            GSH(11) = SH_C3[2] * y * (4.f * zz - xx - yy) * g[ch];
                            GSH(12) = SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy) * g[ch];
                            GSH(13) = SH_C3[4] * x * (4.f * zz - xx - yy) * g[ch];
                            GSH(14) = SH_C3[5] * z * (xx - yy) * g[ch];
                            GSH(15) = SH_C3[6] * x * (xx - 3.f * yy) * g[ch];
                            dx_ += SH_C3[0] * SH(9) * 3.f * 2.f * xy_ + SH_C3[1] * SH(10) * yz + SH_C3[2] * SH(11) * -2.f * xy_ +
                                   SH_C3[3] * SH(12) * -3.f * 2.f * xz + SH_C3[4] * SH(13) * (-3.f * xx + 4.f * zz - yy) +
                                   SH_C3[5] * SH(14) * 2.f * xz + SH_C3[6] * SH(15) * 3.f * (xx - yy);
                            dy_ += SH_C3[0] * SH(9) * 3.f * (xx - yy) + SH_C3[1] * SH(10) * xz +
                                   SH_C3[2] * SH(11) * (-3.f * yy + 4.f * zz - xx) + SH_C3[3] * SH(12) * -3.f * 2.f * yz +
                                   SH_C3[4] * SH(13) * -2.f * xy_ + SH_C3[5] * SH(14) * -2.f * yz +
                                   SH_C3[6] * SH(15) * -3.f * 2.f * xy_;
                            dz_ += SH_C3[1] * SH(10) * xy_ + SH_C3[2] * SH(11) * 4.f * 2.f * yz +
                                   SH_C3[3] * SH(12) * 3.f * (2.f * zz - xx - yy) + SH_C3[4] * SH(13) * 4.f * 2.f * xz +
                                   SH_C3[5] * SH(14) * (xx - yy);
                        }
                    }
                }
#undef SH
#undef GSH
                ddir[0] += dx_ * g[ch];
                ddir[1] += dy_ * g[ch];
                ddir[2] += dz_ * g[ch];
            }
            /* through dir = d0/|d0| */
            const float sum2 = d0[0] * d0[0] + d0[1] * d0[1] + d0[2] * d0[2];
            const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
            dmean[0] += ((sum2 - d0[0] * d0[0]) * ddir[0] - d0[1] * d0[0] * ddir[1] - d0[2] * d0[0] * ddir[2]) * invsum32;
            dmean[1] += (-d0[0] * d0[1] * ddir[0] + (sum2 - d0[1] * d0[1]) * ddir[1] - d0[2] * d0[1] * ddir[2]) * invsum32;
            dmean[2] += (-d0[0] * d0[2] * ddir[0] - d0[1] * d0[2] * ddir[1] + (sum2 - d0[2] * d0[2]) * ddir[2]) * invsum32;
        }
        for (int k = 0; k < 3; ++k) dL_dmeans3D[3 * i + k] = dmean[k];

        /* ---- K9 (iii): cov3D -> scale, raw quaternion ---- */
        if (!use_precomp_cov) {
            const float* q = rotations + 4 * i;
            const float mod = s->scale_modifier;
            float R[3][3];
            quat_to_R(q, R);
            const float sc[3] = {mod * scales[3 * i], mod * scales[3 * i + 1], mod * scales[3 * i + 2]};
            /* symmetric matrix form of the packed gradient: off-diagonals halved */
            const float gS[3][3] = {{gcov[0], 0.5f * gcov[1], 0.5f * gcov[2]},
                                    {0.5f * gcov[1], gcov[3], 0.5f * gcov[4]},
                                    {0.5f * gcov[2], 0.5f * gcov[4], gcov[5]}};
            /* Sigma = M^T M, M[k][j] = s_k R[j][k]  =>  dL/dM[k][j] = 2 * sum_l M[k][l] gS[l][j] */
            float dM[3][3];
            for (int k = 0; k < 3; ++k)
                for (int j = 0; j < 3; ++j)
                    dM[k][j] = 2.0f * (sc[k] * R[0][k] * gS[0][j] + sc[k] * R[1][k] * gS[1][j] + sc[k] * R[2][k] * gS[2][j]);
            /* dL/ds_k = sum_j R[j][k] dM[k][j] with s = mod * scale.  Upstream's computeCov3D backward returns THIS as
             * dL/dscale (it differentiates w.r.t. the modified scale and never multiplies by the modifier); the exact
             * chain rule (x mod) is the option.  Identical at the reference's scaling_modifier = 1.0
             * (gaussian_renderer/__init__.py:19). */
            const float smul = s->exact_scale_grad ? mod : 1.0f;
            for (int k = 0; k < 3; ++k)
                dL_dscale[3 * i + k] = smul * (R[0][k] * dM[k][0] + R[1][k] * dM[k][1] + R[2][k] * dM[k][2]);
            /* dL/dR[j][k] = s_k dM[k][j] */
            float dR[3][3];
            for (int j = 0; j < 3; ++j)
                for (int k = 0; k < 3; ++k) dR[j][k] = sc[k] * dM[k][j];
            const float r = q[0], x = q[1], y = q[2], z = q[3];
            dL_drot[4 * i + 0] = 2 * z * (dR[1][0] - dR[0][1]) + 2 * y * (dR[0][2] - dR[2][0]) + 2 * x * (dR[2][1] - dR[1][2]);
            dL_drot[4 * i + 1] = 2 * y * (dR[0][1] + dR[1][0]) + 2 * z * (dR[0][2] + dR[2][0]) + 2 * r * (dR[2][1] - dR[1][2]) -
                                 4 * x * (dR[2][2] + dR[1][1]);
            dL_drot[4 * i + 2] = 2 * x * (dR[0][1] + dR[1][0]) + 2 * r * (dR[0][2] - dR[2][0]) + 2 * z * (dR[1][2] + dR[2][1]) -
                                 4 * y * (dR[2][2] + dR[0][0]);
            dL_drot[4 * i + 3] = 2 * r * (dR[1][0] - dR[0][1]) + 2 * x * (dR[0][2] + dR[2][0]) + 2 * y * (dR[1][2] + dR[2][1]) -
                                 4 * z * (dR[1][1] + dR[0][0]);
        }
    }
}

int ora_abi_version(void) { return 1; }
