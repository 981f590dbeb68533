// gls_kernels.hip -- fused L1 + SSIM image loss (forward / backward) and the densification statistics
// update: the training-step neighbours of the render path (include/gls.h).
//
// SSIM forward, per 16x16 output tile: the 26x26 halo of both images goes to LDS once (zero padding as
// F.conv2d(padding=5)), the five windowed moments E[x], E[y], E[x^2], E[y^2], E[xy] are formed separably
// (11 horizontal taps into LDS, 11 vertical taps in registers), the SSIM map and |x-y| are reduced per
// workgroup into fixed-order partials, and the three partial derivatives the backward needs are written
// as planes.  Backward: the same separable window over those three planes, combined with x and y.
// All of it is HBM-trivial (5 MB images); the point is ~4 launches instead of ~60.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>

#include "../../include/gls.h"
#include "launch_prof.h"

namespace gls {

constexpr int WIN = GLS_SSIM_WINDOW, RAD = WIN / 2;
// Round 4: 32 x 16 output tiles (a 42 x 26 halo: 2.1 x the pixels instead of 2.6 x at 16 x 16) and REGISTER-BLOCKED taps -- a thread of the
// horizontal pass owns four neighbouring columns of a row and reads their 14 inputs once (28 LDS reads for four outputs where one output at a
// time took 22 each), a thread of the vertical pass owns two neighbouring rows of a column (12 reads per moment for two outputs instead of
// 22).  Every output still adds its eleven taps in the order k = 0 .. 10, so the values are the bits of the round-1 kernel.
// Measured (rocprofv3, train workload): k_l1_ssim_fwd 24.4 -> 24.2 us, k_l1_ssim_bwd 25.4 -> 22.6 us -- the LDS reads were not what bounds the pair: the
// forward writes 15.9 MB of derivative planes and the backward reads them back with their halos (~38 MB each way per kernel at ~3 TB/s).
constexpr int TX = 32, TY = 16, HX = TX + 2 * RAD, HY = TY + 2 * RAD, HXS = HX + 2;   // (row stride 44: 16-byte rows)
constexpr float SSIM_C1 = 0.01f * 0.01f, SSIM_C2 = 0.03f * 0.03f;

struct Window {
    float w[WIN];
};

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
// sum over the 64 lanes, valid in lane 63
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

// loads the (HY x HX) halo of one plane into LDS, zero outside the image
__device__ __forceinline__ void load_halo(float (*dst)[HXS], const float* __restrict__ plane, int H, int W, int x0, int y0, int tid)
{
    for (int i = tid; i < HY * HX; i += 256) {
        const int r = i / HX, c = i - r * HX;
        const int gy = y0 + r - RAD, gx = x0 + c - RAD;
        const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
        dst[r][c] = in ? plane[(size_t)gy * W + gx] : 0.f;
    }
}

__global__ __launch_bounds__(256) void k_l1_ssim_fwd(int H, int W, const float* __restrict__ img1, const float* __restrict__ img2,
                                                      Window win, float* __restrict__ maps, size_t map_stride,
                                                      float2* __restrict__ partial)
{
    __shared__ __attribute__((aligned(16))) float sx[HY][HXS], sy[HY][HXS];
    __shared__ __attribute__((aligned(16))) float hq[5][HY][TX];
    __shared__ float red[2][4];
    const int tid = threadIdx.x;
    const int plane = blockIdx.z, x0 = blockIdx.x * TX, y0 = blockIdx.y * TY;
    const size_t poff = (size_t)plane * H * W;
    load_halo(sx, img1 + poff, H, W, x0, y0, tid);
    load_halo(sy, img2 + poff, H, W, x0, y0, tid);
    __syncthreads();
    if (tid < HY * (TX / 4)) {   // horizontal taps: row r, columns c0 .. c0 + 3
        const int r = tid >> 3, c0 = (tid & 7) * 4;
        float xs[WIN + 3], ys[WIN + 3];
#pragma unroll
        for (int k = 0; k < WIN + 3; ++k) { xs[k] = sx[r][c0 + k]; ys[k] = sy[r][c0 + k]; }
        float o[5][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float a = 0.f, b = 0.f, aa = 0.f, bb = 0.f, ab = 0.f;
#pragma unroll
            for (int k = 0; k < WIN; ++k) {
                const float x = xs[j + k], y = ys[j + k], w = win.w[k];
                a += w * x;
                b += w * y;
                aa += w * (x * x);
                bb += w * (y * y);
                ab += w * (x * y);
            }
            o[0][j] = a; o[1][j] = b; o[2][j] = aa; o[3][j] = bb; o[4][j] = ab;
        }
#pragma unroll
        for (int m = 0; m < 5; ++m) *reinterpret_cast<float4*>(&hq[m][r][c0]) = make_float4(o[m][0], o[m][1], o[m][2], o[m][3]);
    }
    __syncthreads();
    const int tx = tid & 31, ty0 = (tid >> 5) * 2;   // vertical taps: column tx, rows ty0 and ty0 + 1
    float col[5][WIN + 1];
#pragma unroll
    for (int m = 0; m < 5; ++m)
#pragma unroll
        for (int k = 0; k < WIN + 1; ++k) col[m][k] = hq[m][ty0 + k][tx];
    float l1 = 0.f, ss = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        float mu1 = 0.f, mu2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
        for (int k = 0; k < WIN; ++k) {
            const float w = win.w[k];
            mu1 += w * col[0][j + k];
            mu2 += w * col[1][j + k];
            e11 += w * col[2][j + k];
            e22 += w * col[3][j + k];
            e12 += w * col[4][j + k];
        }
        const int ty = ty0 + j, px = x0 + tx, py = y0 + ty;
        if (px < W && py < H) {
            const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
            const float s1 = e11 - mu1_sq, s2 = e22 - mu2_sq, s12 = e12 - mu12;
            const float A = 2.f * mu12 + SSIM_C1, Bv = 2.f * s12 + SSIM_C2;
            const float Cv = mu1_sq + mu2_sq + SSIM_C1, D = s1 + s2 + SSIM_C2;
            const float inv = 1.f / (Cv * D);
            const float m = A * Bv * inv;
            ss += m;
            l1 += fabsf(sx[ty + RAD][tx + RAD] - sy[ty + RAD][tx + RAD]);
            if (maps) {
                const size_t o = poff + (size_t)py * W + px;
                maps[o] = (2.f * mu2 * (Bv - A) - m * 2.f * mu1 * (D - Cv)) * inv;   // d m / d mu1  (E[x^2], E[xy] held fixed)
                maps[o + map_stride] = -m / D;                                       // d m / d E[x^2]
                maps[o + 2 * map_stride] = 2.f * A * inv;                            // d m / d E[xy]
            }
        }
    }
    l1 = wave_sum_hi(l1);
    ss = wave_sum_hi(ss);
    if ((tid & 63) == 63) { red[0][tid >> 6] = l1; red[1][tid >> 6] = ss; }
    __syncthreads();
    if (tid == 0) {
        const size_t blk = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        partial[blk] = make_float2((red[0][0] + red[0][1]) + (red[0][2] + red[0][3]), (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]));
    }
}

// one workgroup per image: fixed-order sum of that image's partials
__global__ __launch_bounds__(256) void k_reduce_partials(const float2* __restrict__ partial, int per_image, float scale, float2* __restrict__ sums)
{
    __shared__ float red[2][4];
    const int tid = threadIdx.x;
    const float2* p = partial + (size_t)blockIdx.x * per_image;
    float a = 0.f, b = 0.f;
    for (int i = tid; i < per_image; i += 256) {
        const float2 v = p[i];
        a += v.x;
        b += v.y;
    }
    a = wave_sum_hi(a);
    b = wave_sum_hi(b);
    if ((tid & 63) == 63) { red[0][tid >> 6] = a; red[1][tid >> 6] = b; }
    __syncthreads();
    if (tid == 0)
        sums[blockIdx.x] = make_float2(scale * ((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])),
                                       scale * ((red[1][0] + red[1][1]) + (red[1][2] + red[1][3])));
}

__global__ __launch_bounds__(256) void k_l1_ssim_bwd(int C, int H, int W, const float* __restrict__ img1, const float* __restrict__ img2,
                                                      const float* __restrict__ maps, size_t map_stride, Window win,
                                                      const float* __restrict__ g_l1, const float* __restrict__ g_ssim, int g_stride, float scale,
                                                      float* __restrict__ d_img1)
{
    __shared__ __attribute__((aligned(16))) float sm[3][HY][HXS];
    __shared__ __attribute__((aligned(16))) float hq[3][HY][TX];
    const int tid = threadIdx.x;
    const int plane = blockIdx.z, x0 = blockIdx.x * TX, y0 = blockIdx.y * TY;
    const size_t poff = (size_t)plane * H * W;
#pragma unroll
    for (int q = 0; q < 3; ++q) load_halo(sm[q], maps + q * map_stride + poff, H, W, x0, y0, tid);
    __syncthreads();
    if (tid < HY * (TX / 4)) {   // horizontal taps of the three derivative planes: row r, columns c0 .. c0 + 3
        const int r = tid >> 3, c0 = (tid & 7) * 4;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            float v[WIN + 3];
#pragma unroll
            for (int k = 0; k < WIN + 3; ++k) v[k] = sm[q][r][c0 + k];
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float a = 0.f;
#pragma unroll
                for (int k = 0; k < WIN; ++k) a += win.w[k] * v[j + k];
                o[j] = a;
            }
            *reinterpret_cast<float4*>(&hq[q][r][c0]) = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
    __syncthreads();
    const int tx = tid & 31, ty0 = (tid >> 5) * 2;
    float col[3][WIN + 1];
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int k = 0; k < WIN + 1; ++k) col[q][k] = hq[q][ty0 + k][tx];
    const int img = plane / C;   // dL/d(mean |x - y|) and dL/d(mean ssim) of this image; a missing one is zero
    float2 gi = make_float2(g_l1 ? g_l1[(size_t)img * g_stride] : 0.f, g_ssim ? g_ssim[(size_t)img * g_stride] : 0.f);
    gi.x *= scale;
    gi.y *= scale;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        float ca = 0.f, cb = 0.f, cc = 0.f;
#pragma unroll
        for (int k = 0; k < WIN; ++k) {
            const float w = win.w[k];
            ca += w * col[0][j + k];
            cb += w * col[1][j + k];
            cc += w * col[2][j + k];
        }
        const int px = x0 + tx, py = y0 + ty0 + j;
        if (px < W && py < H) {
            const size_t o = poff + (size_t)py * W + px;
            const float x = img1[o], y = img2[o];
            const float df = x - y;
            const float sgn = df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f);
            d_img1[o] = gi.y * (ca + 2.f * x * cb + y * cc) + gi.x * sgn;
        }
    }
}

// ---- plain L1 -------------------------------------------------------------------------------------------------
__device__ __forceinline__ float sgn_scaled(float d, float g) { return d > 0.f ? g : (d < 0.f ? -g : 0.f); }
// GRAD: also d_a <- gs * sign(a - b), the gradient for an upstream gradient of exactly 1 (gs = the forward's mean scale): a backward seeded
// with the constant 1 (loss.backward() on the loss itself) then has nothing left to launch
// FUSED (round 6): the reduction over the workgroups without a second launch and without a fence.  Every workgroup adds its partial, as 2^-30
// fixed point, and a 1 in the top byte to ONE 64-bit word with ONE returning atomic: the atomic is the only thing the workgroups share, so the one
// that reads back (grid - 1) arrivals holds every other partial in the value it got -- it writes the loss and leaves the word zero for the
// stream's next call.  Integer adds commute: the same bits whatever order the workgroups arrive in.  (Rounds 2 - 3 measured a ticket + fence +
// re-read of the partials at the price of the launch it replaced; here nothing is re-read.)  At most 255 workgroups; |a - b| sums below 2^26.
template <bool GRAD, bool FUSED>
__global__ __launch_bounds__(1024) void k_l1_fwd(long long n, const float* __restrict__ a, const float* __restrict__ b, float2* __restrict__ partial,
                                                  float gs, float* __restrict__ d_a, unsigned long long* __restrict__ word, float* __restrict__ sum)
{
    // (blockDim.x: 256 in the two-launch form; the one-launch form is held to 255 workgroups by its arrival count and takes WIDE ones -- with
    //  256 threads it ran one wave per SIMD, five dependent trips of two loads each: 7.6 us for 16 MB)
    __shared__ float red[16];
    const int tid = threadIdx.x, nt = (int)blockDim.x;
    const long long n4 = n >> 2;
    float s = 0.f;
    for (long long i = (long long)blockIdx.x * nt + tid; i < n4; i += (long long)gridDim.x * nt) {
        const float4 u = ((const float4*)a)[i], v = ((const float4*)b)[i];
        const float dx = u.x - v.x, dy = u.y - v.y, dz = u.z - v.z, dw = u.w - v.w;
        s += (fabsf(dx) + fabsf(dy)) + (fabsf(dz) + fabsf(dw));
        if (GRAD) ((float4*)d_a)[i] = make_float4(sgn_scaled(dx, gs), sgn_scaled(dy, gs), sgn_scaled(dz, gs), sgn_scaled(dw, gs));
    }
    if (blockIdx.x == 0 && tid < (int)(n & 3)) {
        const float d = a[(n4 << 2) + tid] - b[(n4 << 2) + tid];
        s += fabsf(d);
        if (GRAD) d_a[(n4 << 2) + tid] = sgn_scaled(d, gs);
    }
    s = wave_sum_hi(s);
    if ((tid & 63) == 63) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) {
        float tot = (red[0] + red[1]) + (red[2] + red[3]);
        for (int w = 4; w < (nt >> 6); w += 4) tot += (red[w] + red[w + 1]) + (red[w + 2] + red[w + 3]);
        if constexpr (FUSED) {
            const unsigned long long mine = (unsigned long long)((double)tot * 1073741824.0 + 0.5) & 0x00FFFFFFFFFFFFFFull;
            const unsigned long long old = atomicAdd(word, mine + (1ull << 56));
            if ((old >> 56) == (unsigned long long)(gridDim.x - 1)) {
                const unsigned long long all = (old & 0x00FFFFFFFFFFFFFFull) + mine;
                *sum = gs * (float)((double)all * (1.0 / 1073741824.0));
                __hip_atomic_store(word, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (the next call on this stream starts from zero)
            }
        } else {
            partial[blockIdx.x] = make_float2(tot, 0.f);
        }
    }
}
// (plain names for the launch sites: the event profile keys its table by the text of the launch expression)
constexpr auto k_l1_fwd_1 = &k_l1_fwd<false, true>;     // one launch, loss only
constexpr auto k_l1_fwd_1g = &k_l1_fwd<true, true>;     // one launch, loss + the unit-seed gradient image
constexpr auto k_l1_fwd_2 = &k_l1_fwd<false, false>;    // partials for k_l1_reduce
constexpr auto k_l1_fwd_2g = &k_l1_fwd<true, false>;
__global__ __launch_bounds__(256) void k_l1_reduce(const float2* __restrict__ partial, int count, float scale, float* __restrict__ sum)
{
    __shared__ float red[4];
    const int tid = threadIdx.x;
    float s = 0.f;
    for (int i = tid; i < count; i += 256) s += partial[i].x;
    s = wave_sum_hi(s);
    if ((tid & 63) == 63) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) *sum = scale * ((red[0] + red[1]) + (red[2] + red[3]));
}
__global__ __launch_bounds__(256) void k_l1_bwd(long long n, const float* __restrict__ a, const float* __restrict__ b,
                                                 const float* __restrict__ g, float scale, float* __restrict__ d_a)
{
    const float gv = *g * scale;
    const long long n4 = n >> 2;
    const int tid = threadIdx.x;
    for (long long i = (long long)blockIdx.x * 256 + tid; i < n4; i += (long long)gridDim.x * 256) {
        const float4 u = ((const float4*)a)[i], v = ((const float4*)b)[i];
        ((float4*)d_a)[i] = make_float4(sgn_scaled(u.x - v.x, gv), sgn_scaled(u.y - v.y, gv), sgn_scaled(u.z - v.z, gv), sgn_scaled(u.w - v.w, gv));
    }
    if (blockIdx.x == 0 && tid < (int)(n & 3)) {
        const long long i = (n4 << 2) + tid;
        d_a[i] = sgn_scaled(a[i] - b[i], gv);
    }
}

// ---- densification statistics -----------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_densify_stats(int P, const int* __restrict__ radii, const float* __restrict__ vgrad,
                                                        float* __restrict__ max_radii2D, float* __restrict__ accum, float* __restrict__ denom)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const int r = radii[i];
    if (r <= 0) return;
    max_radii2D[i] = fmaxf(max_radii2D[i], (float)r);
    const float gx = vgrad[3 * i], gy = vgrad[3 * i + 1];
    accum[i] += sqrtf(gx * gx + gy * gy);
    denom[i] += 1.f;
}

// GaussianModel.add_densification_stats alone (scene/gaussian_model.py:517-519) with the caller's own bool filter: the form the reference's
// train.py:198 calls, rebound by patch_reference() (train.py:197's max_radii2D line is train.py's own torch code and stays there)
__global__ __launch_bounds__(256) void k_add_densify_stats(int P, const unsigned char* __restrict__ filter, const float* __restrict__ vgrad, int vstride,
                                                            float* __restrict__ accum, float* __restrict__ denom)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P || !filter[i]) return;
    const float gx = vgrad[(size_t)vstride * i], gy = vgrad[(size_t)vstride * i + 1];
    accum[i] += sqrtf(gx * gx + gy * gy);
    denom[i] += 1.f;
}

}  // namespace gls

// ---------------------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
static int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}
#define LAUNCH_CHECK(what)                                                                            \
    do {                                                                                              \
        hipError_t e_ = hipGetLastError();                                                            \
        if (e_ != hipSuccess) return fail(GLS_E_HIP, "%s: %s", what, hipGetErrorString(e_));          \
    } while (0)

static gls::Window make_window()
{
    // utils/loss_utils.py:23-25: exp(-(x - 5)^2 / (2 * 1.5^2)) held in fp32, divided by its fp32 sum
    gls::Window w;
    float sum = 0.f;
    for (int x = 0; x < gls::WIN; ++x) {
        w.w[x] = (float)std::exp(-(double)((x - gls::RAD) * (x - gls::RAD)) / (2.0 * 1.5 * 1.5));
        sum += w.w[x];
    }
    for (int x = 0; x < gls::WIN; ++x) w.w[x] /= sum;
    return w;
}
static const int kL1Blocks = 1024;

extern "C" {

int gls_abi_version(void) { return GLS_ABI_VERSION; }
const char* gls_last_error(void) { return g_err; }

static bool image_args_ok(int32_t B, int32_t C, int32_t H, int32_t W)
{
    return B > 0 && C > 0 && H > 0 && W > 0 && (int64_t)B * C <= 65535 && (H + gls::TY - 1) / gls::TY <= 65535;
}

int64_t gls_partial_floats(int32_t B, int32_t C, int32_t H, int32_t W)
{
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return 2 * kL1Blocks;
    const int64_t tiles = (int64_t)((W + gls::TX - 1) / gls::TX) * ((H + gls::TY - 1) / gls::TY) * B * C;
    return 2 * (tiles > kL1Blocks ? tiles : kL1Blocks);
}

int gls_l1_ssim_forward(int32_t B, int32_t C, int32_t H, int32_t W, const float* img1, const float* img2, float scale, float* sums,
                        float* maps, float* partial, void* stream_)
{
    if (!image_args_ok(B, C, H, W)) return fail(GLS_E_ARG, "bad image shape (%d,%d,%d,%d)", B, C, H, W);
    if (!img1 || !img2 || !sums || !partial) return fail(GLS_E_ARG, "null pointer");
    hipStream_t stream = (hipStream_t)stream_;
    const dim3 grid((W + gls::TX - 1) / gls::TX, (H + gls::TY - 1) / gls::TY, B * C);
    const size_t stride = (size_t)B * C * H * W;
    PROF_LAUNCH(gls::k_l1_ssim_fwd, grid, dim3(256), 0, stream, H, W, img1, img2, make_window(), maps, stride, (float2*)partial);
    LAUNCH_CHECK("k_l1_ssim_fwd");
    PROF_LAUNCH(gls::k_reduce_partials, dim3(B), dim3(256), 0, stream, (const float2*)partial, (int)(grid.x * grid.y * C), scale, (float2*)sums);
    LAUNCH_CHECK("k_reduce_partials");
    return GLS_OK;
}

static int l1_ssim_backward_impl(int32_t B, int32_t C, int32_t H, int32_t W, const float* img1, const float* img2, const float* maps,
                                 const float* g_l1, const float* g_ssim, int g_stride, float scale, float* d_img1, void* stream_)
{
    if (!image_args_ok(B, C, H, W)) return fail(GLS_E_ARG, "bad image shape (%d,%d,%d,%d)", B, C, H, W);
    if (!img1 || !img2 || !maps || !d_img1 || g_stride < 0) return fail(GLS_E_ARG, "null pointer");
    hipStream_t stream = (hipStream_t)stream_;
    const dim3 grid((W + gls::TX - 1) / gls::TX, (H + gls::TY - 1) / gls::TY, B * C);
    const size_t stride = (size_t)B * C * H * W;
    PROF_LAUNCH(gls::k_l1_ssim_bwd, grid, dim3(256), 0, stream, C, H, W, img1, img2, maps, stride, make_window(), g_l1, g_ssim, g_stride, scale, d_img1);
    LAUNCH_CHECK("k_l1_ssim_bwd");
    return GLS_OK;
}

int gls_l1_ssim_backward(int32_t B, int32_t C, int32_t H, int32_t W, const float* img1, const float* img2, const float* maps,
                         const float* g, float scale, float* d_img1, void* stream_)
{
    if (!g) return fail(GLS_E_ARG, "null pointer");
    return l1_ssim_backward_impl(B, C, H, W, img1, img2, maps, g, g + 1, 2, scale, d_img1, stream_);
}

int gls_l1_ssim_backward_split(int32_t B, int32_t C, int32_t H, int32_t W, const float* img1, const float* img2, const float* maps,
                               const float* g_l1, const float* g_ssim, int32_t g_stride, float scale, float* d_img1, void* stream_)
{
    return l1_ssim_backward_impl(B, C, H, W, img1, img2, maps, g_l1, g_ssim, g_stride, scale, d_img1, stream_);
}

// one accumulator word of the fused L1 forward per stream (calls on a stream are ordered, concurrent streams never share a word)
static unsigned long long* l1_word_for(hipStream_t stream)
{
    constexpr int kWords = 64;
    static std::mutex mu;
    static std::vector<hipStream_t> owners;
    static std::vector<int> owner_dev;                  // (the table is keyed by (device, stream))
    static std::vector<unsigned long long*> bases;      // one block of words per device
    static std::vector<int> base_dev;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lk(mu);
    unsigned long long* b = nullptr;
    for (size_t i = 0; i < bases.size(); ++i)
        if (base_dev[i] == dev) b = bases[i];
    if (!b) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return nullptr;   // (no allocation inside a capture: the two-launch form)
        if (hipMalloc((void**)&b, kWords * sizeof(unsigned long long)) != hipSuccess) return nullptr;
        if (hipMemset(b, 0, kWords * sizeof(unsigned long long)) != hipSuccess) return nullptr;
        bases.push_back(b); base_dev.push_back(dev);
    }
    int slot = -1, used = 0;
    for (size_t i = 0; i < owners.size(); ++i) {
        if (owner_dev[i] != dev) continue;
        if (owners[i] == stream) slot = used;
        ++used;
    }
    if (slot < 0) {
        if (used >= kWords) return nullptr;   // more streams than words: the two-launch form
        owners.push_back(stream); owner_dev.push_back(dev);
        slot = used;
    }
    return b + slot;
}

static int l1_forward_impl(int64_t n, const float* a, const float* b, float scale, float* sum, float* partial, float* d_a, bool grad, void* stream_)
{
    if (n < 0 || !sum || !partial || (n > 0 && (!a || !b || (grad && !d_a)))) return fail(GLS_E_ARG, "bad arguments");
    if ((((uintptr_t)a) | ((uintptr_t)b) | (grad ? (uintptr_t)d_a : 0)) & 15) return fail(GLS_E_ARG, "buffers must be 16-byte aligned");
    hipStream_t stream = (hipStream_t)stream_;
    int blocks = (int)(((n >> 2) + 255) / 256);
    blocks = blocks < 1 ? 1 : (blocks > kL1Blocks ? kL1Blocks : blocks);
    // one launch (round 6): <= 255 workgroups, each adds its partial to the stream's accumulator word with one returning atomic; the last arriver writes the loss
    static const bool fused_on = [] { const char* e = getenv("GLS_L1_FUSED"); return !(e && e[0] == '0'); }();
    unsigned long long* word = (fused_on && n < (1ll << 26)) ? l1_word_for(stream) : nullptr;
    if (word) {
        static const int fmax = [] { const char* e = getenv("GLS_L1_BLOCKS"); const int v = e ? atoi(e) : 0; return v > 0 && v <= 255 ? v : 255; }();   // (A/B runs)
        static const int fthreads = [] { const char* e = getenv("GLS_L1_THREADS"); const int v = e ? atoi(e) : 0; return v == 256 || v == 512 || v == 1024 ? v : 1024; }();
        const int wide = (int)(((n >> 2) + fthreads - 1) / fthreads);
        const int fb = wide < 1 ? 1 : (wide > fmax ? fmax : wide);
        if (grad) PROF_LAUNCH(gls::k_l1_fwd_1g, dim3(fb), dim3(fthreads), 0, stream, (long long)n, a, b, (float2*)partial, scale, d_a, word, sum);
        else PROF_LAUNCH(gls::k_l1_fwd_1, dim3(fb), dim3(fthreads), 0, stream, (long long)n, a, b, (float2*)partial, scale, (float*)nullptr, word, sum);
        LAUNCH_CHECK("k_l1_fwd");
        return GLS_OK;
    }
    if (grad) PROF_LAUNCH(gls::k_l1_fwd_2g, dim3(blocks), dim3(256), 0, stream, (long long)n, a, b, (float2*)partial, scale, d_a, (unsigned long long*)nullptr, (float*)nullptr);
    else PROF_LAUNCH(gls::k_l1_fwd_2, dim3(blocks), dim3(256), 0, stream, (long long)n, a, b, (float2*)partial, scale, (float*)nullptr, (unsigned long long*)nullptr, (float*)nullptr);
    LAUNCH_CHECK("k_l1_fwd");
    PROF_LAUNCH(gls::k_l1_reduce, dim3(1), dim3(256), 0, stream, (const float2*)partial, blocks, scale, sum);
    LAUNCH_CHECK("k_l1_reduce");
    return GLS_OK;
}

int gls_l1_forward(int64_t n, const float* a, const float* b, float scale, float* sum, float* partial, void* stream_)
{
    return l1_forward_impl(n, a, b, scale, sum, partial, nullptr, false, stream_);
}

int gls_l1_forward_grad(int64_t n, const float* a, const float* b, float scale, float* sum, float* partial, float* d_a, void* stream_)
{
    return l1_forward_impl(n, a, b, scale, sum, partial, d_a, true, stream_);
}

int gls_l1_backward(int64_t n, const float* a, const float* b, const float* g, float scale, float* d_a, void* stream_)
{
    if (n < 0 || !g || (n > 0 && (!a || !b || !d_a))) return fail(GLS_E_ARG, "bad arguments");
    if ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)d_a)) & 15) return fail(GLS_E_ARG, "buffers must be 16-byte aligned");
    if (n == 0) return GLS_OK;
    hipStream_t stream = (hipStream_t)stream_;
    int blocks = (int)(((n >> 2) + 255) / 256);
    blocks = blocks < 1 ? 1 : (blocks > 4096 ? 4096 : blocks);
    PROF_LAUNCH(gls::k_l1_bwd, dim3(blocks), dim3(256), 0, stream, (long long)n, a, b, g, scale, d_a);
    LAUNCH_CHECK("k_l1_bwd");
    return GLS_OK;
}

int gls_densification_stats(int32_t P, const int32_t* radii, const float* viewspace_grad, float* max_radii2D, float* xyz_gradient_accum,
                            float* denom, void* stream_)
{
    if (P < 0) return fail(GLS_E_ARG, "P < 0");
    if (P == 0) return GLS_OK;
    if (!radii || !viewspace_grad || !max_radii2D || !xyz_gradient_accum || !denom) return fail(GLS_E_ARG, "null pointer");
    PROF_LAUNCH(gls::k_densify_stats, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream_, P, radii, viewspace_grad,
                       max_radii2D, xyz_gradient_accum, denom);
    LAUNCH_CHECK("k_densify_stats");
    return GLS_OK;
}

int gls_add_densification_stats(int32_t P, const uint8_t* update_filter, const float* viewspace_grad, int32_t grad_stride, float* xyz_gradient_accum,
                                float* denom, void* stream_)
{
    if (P < 0 || grad_stride < 2) return fail(GLS_E_ARG, "P < 0 or grad_stride < 2");
    if (P == 0) return GLS_OK;
    if (!update_filter || !viewspace_grad || !xyz_gradient_accum || !denom) return fail(GLS_E_ARG, "null pointer");
    PROF_LAUNCH(gls::k_add_densify_stats, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream_, P, update_filter, viewspace_grad, grad_stride,
                       xyz_gradient_accum, denom);
    LAUNCH_CHECK("k_add_densify_stats");
    return GLS_OK;
}

int gls_profile_enable(int on)
{
    lprof::g.on.store(on ? 1 : 0);
    return 0;
}
int gls_profile_collect(void) { return lprof::collect(); }
int gls_profile_entry(int32_t index, const char** name, double* total_ms, int64_t* launches)
{
    long long n = 0;
    const int rc = lprof::entry(index, name, total_ms, &n);
    if (launches) *launches = (int64_t)n;
    return rc;
}
int gls_profile_reset(void)
{
    lprof::reset();
    return 0;
}

}  // extern "C"
