// pmc_calibrate.hip -- known-byte-count kernels in the access patterns the rasterizer's kernels use, to calibrate what rocprofv3's
// FETCH_SIZE / WRITE_SIZE report for each of them on gfx950 (MI355X_MICROARCH.md, HBM section: only the wide coalesced stream is
// calibrated there, x2).  Build + run:  bash tools/pmc_calibrate.sh   (writes profiles/<tag>_pmc_calibration.json)
//
// Every kernel touches a footprint far beyond L2 + Infinity Cache (>= 1 GiB) exactly once, so the bytes it REQUESTS are the bytes
// HBM must deliver (reads) or absorb (writes); for the record gathers the requested bytes and the 64-byte / 128-byte line footprint
// are both printed, the counter is compared with each.
#include <hip/hip_runtime.h>
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f32x8 __attribute__((ext_vector_type(8)));

__global__ void cal_stream16(const float4* __restrict__ src, float* __restrict__ sink, size_t n4)   // 16 B per lane, coalesced
{
    float acc = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = src[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 123.456f) sink[0] = acc;
}
__global__ void cal_stream4(const float* __restrict__ src, float* __restrict__ sink, size_t n)       // 4 B per lane, coalesced
{
    float acc = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += src[i];
    if (acc == 123.456f) sink[0] = acc;
}
__global__ void cal_stream8(const double* __restrict__ src, float* __restrict__ sink, size_t n)      // 8 B per lane (the sort's keys)
{
    double acc = 0.0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += src[i];
    if (acc == 123.456) sink[0] = (float)acc;
}
// per-lane gather of the first 32 bytes of a random 48-byte record (the tile sort's epilogue)
__global__ void cal_gather32(const float4* __restrict__ rec, const uint32_t* __restrict__ idx, float* __restrict__ sink, size_t n)
{
    float acc = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = idx[i];
        const float4 a = rec[3 * r], b = rec[3 * r + 1];
        acc += a.x + b.y;
    }
    if (acc == 123.456f) sink[0] = acc;
}
// wave-uniform scalar loads of 36 bytes of a random 48-byte record (the blend kernels): s_load_dwordx8 + s_load_dword
__global__ void cal_sload36(const float4* __restrict__ rec, const uint32_t* __restrict__ idx, float* __restrict__ sink, size_t n_per_wave)
{
    typedef const __attribute__((address_space(4))) char* cbytes;
    const cbytes recb = (cbytes)(uintptr_t)rec;
    const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6;
    const uint32_t* __restrict__ my = idx + wave * n_per_wave;
    float acc = 0.f;
    for (size_t j = 0; j < n_per_wave; ++j) {
        const uint32_t r = __builtin_amdgcn_readfirstlane((int)my[j]);
        const uint32_t off = r * 48u;
        const f32x8 a = *(const __attribute__((address_space(4))) f32x8*)(recb + off);
        const float b = *(const __attribute__((address_space(4))) float*)(recb + off + 32);
        acc += a[0] + a[7] + b;
    }
    if (acc == 123.456f) sink[0] = acc;
}
__global__ void cal_write16(float4* __restrict__ dst, size_t n4)                                     // 16 B per lane, coalesced
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = make_float4((float)i, 1.f, 2.f, 3.f);
}
__global__ void cal_write4_scatter(uint32_t* __restrict__ dst, const uint32_t* __restrict__ idx, size_t n)   // 4 B to a random slot (quadrant streams)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[idx[i]] = (uint32_t)i;
}
// the blend backward's burst: lanes 0..8 of a wave add to 9 consecutive floats of a random 48-byte row
__global__ void cal_atomic9(float* __restrict__ acc, const uint32_t* __restrict__ idx, size_t n_per_wave)
{
    const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    const uint32_t* __restrict__ my = idx + wave * n_per_wave;
    for (size_t j = 0; j < n_per_wave; ++j) {
        const uint32_t r = __builtin_amdgcn_readfirstlane((int)my[j]);
        if (lane < 9) unsafeAtomicAdd(acc + (size_t)12 * r + lane, 1.0f);
    }
}

int main(int argc, char** argv)
{
    const size_t GiB = (size_t)1 << 30;
    const size_t bytes = argc > 1 ? (size_t)atoll(argv[1]) * GiB : 2 * GiB;          // streamed footprint
    const size_t nrec = bytes / 48, ngather = (size_t)1 << 24;                      // 16 M gathers / scalar fetches / scattered writes
    float4* buf; float* sink; uint32_t* idx; uint32_t* idx_lo;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&sink, 256)); CK(hipMalloc(&idx, ngather * 4)); CK(hipMalloc(&idx_lo, ngather * 4));
    CK(hipMemset(buf, 0, bytes));
    std::vector<uint32_t> h(ngather), h2(ngather);
    uint64_t s = 0x9E3779B97F4A7C15ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    for (size_t i = 0; i < ngather; ++i) { h[i] = (uint32_t)(rnd() % nrec); h2[i] = (uint32_t)(rnd() % (bytes / 4)); }
    CK(hipMemcpy(idx, h.data(), ngather * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(idx_lo, h2.data(), ngather * 4, hipMemcpyHostToDevice));
    // line footprints of the gathers (distinct 64-byte and 128-byte lines touched by [48 r, 48 r + 32) resp. + 36)
    auto lines = [&](size_t width, size_t line) {
        std::vector<uint64_t> v; v.reserve(2 * ngather);
        for (size_t i = 0; i < ngather; ++i) { const uint64_t a = (uint64_t)h[i] * 48, b = a + width - 1; for (uint64_t l = a / line; l <= b / line; ++l) v.push_back(l); }
        std::sort(v.begin(), v.end()); return (size_t)(std::unique(v.begin(), v.end()) - v.begin()) * line;
    };
    const int blocks = 256 * 8, thr = 256;
    const size_t waves = (size_t)blocks * thr / 64, per_wave = ngather / waves;
    printf("{\"footprint_bytes\": %zu, \"gathers\": %zu,\n", bytes, ngather);
    printf(" \"requested\": {\"cal_stream16\": %zu, \"cal_stream4\": %zu, \"cal_stream8\": %zu, \"cal_gather32\": %zu, \"cal_sload36\": %zu, "
           "\"cal_write16\": %zu, \"cal_write4_scatter\": %zu, \"cal_atomic9\": %zu},\n",
           bytes, bytes, bytes, ngather * 32 + ngather * 4, waves * per_wave * 36 + ngather * 4, bytes, ngather * 4 + ngather * 4, waves * per_wave * 36);
    printf(" \"line_footprint\": {\"cal_gather32_64B\": %zu, \"cal_gather32_128B\": %zu, \"cal_sload36_64B\": %zu, \"cal_sload36_128B\": %zu}}\n",
           lines(32, 64), lines(32, 128), lines(36, 64), lines(36, 128));
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(cal_stream16, dim3(blocks), dim3(thr), 0, 0, (const float4*)buf, sink, bytes / 16);
        hipLaunchKernelGGL(cal_stream4, dim3(blocks), dim3(thr), 0, 0, (const float*)buf, sink, bytes / 4);
        hipLaunchKernelGGL(cal_stream8, dim3(blocks), dim3(thr), 0, 0, (const double*)buf, sink, bytes / 8);
        hipLaunchKernelGGL(cal_gather32, dim3(blocks), dim3(thr), 0, 0, (const float4*)buf, (const uint32_t*)idx, sink, ngather);
        hipLaunchKernelGGL(cal_sload36, dim3(blocks), dim3(thr), 0, 0, (const float4*)buf, (const uint32_t*)idx, sink, per_wave);
        hipLaunchKernelGGL(cal_write16, dim3(blocks), dim3(thr), 0, 0, buf, bytes / 16);
        hipLaunchKernelGGL(cal_write4_scatter, dim3(blocks), dim3(thr), 0, 0, (uint32_t*)buf, (const uint32_t*)idx_lo, ngather);
        hipLaunchKernelGGL(cal_atomic9, dim3(blocks), dim3(thr), 0, 0, (float*)buf, (const uint32_t*)idx, per_wave);
        CK(hipDeviceSynchronize());
    }
    return 0;
}
