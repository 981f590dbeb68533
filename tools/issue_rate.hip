// Issue rate of ONE wave64 on a SIMD (what bounds the tail of the blend kernels), dependent chain vs independent streams, and the
// same with k waves per SIMD:  hipcc --offload-arch=gfx950 -O2 tools/issue_rate.hip -o build/issue_rate && build/issue_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int STREAMS, bool WITH_EXP>
__global__ void k_chain(float* out, long long* cycles, int iters)
{
    float a[STREAMS];
#pragma unroll
    for (int s = 0; s < STREAMS; ++s) a[s] = 1.0f + 0.001f * (float)(threadIdx.x + s);
    const float m = 0.999f, c = 0.0005f;
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 16 / STREAMS; ++r)
#pragma unroll
            for (int s = 0; s < STREAMS; ++s) {
                a[s] = __builtin_fmaf(a[s], m, c);
                if (WITH_EXP && r == 0) a[s] = __builtin_amdgcn_exp2f(a[s] * 0.01f);
            }
    }
    const long long t1 = clock64();
    float sum = 0.f;
#pragma unroll
    for (int s = 0; s < STREAMS; ++s) sum += a[s];
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

template <int STREAMS, bool WITH_EXP>
int run(const char* name, int waves_per_simd, float* d, long long* dc)
{
    const int iters = 20000;
    // one workgroup of 64 * 4 * waves_per_simd threads on one CU: waves_per_simd waves on every SIMD (1 -> a lone wave per SIMD)
    const int threads = 64 * (waves_per_simd == 0 ? 1 : 4 * waves_per_simd);
    hipLaunchKernelGGL((k_chain<STREAMS, WITH_EXP>), dim3(1), dim3(threads), 0, 0, d, dc, iters);
    CK(hipDeviceSynchronize());
    long long cyc = 0;
    CK(hipMemcpy(&cyc, dc, 8, hipMemcpyDeviceToHost));
    const double instr = (double)iters * (16 + (WITH_EXP ? 2 * STREAMS : 0));   // fma (+ mul + exp per stream)
    printf("%-34s waves/SIMD %d: %.2f clock64 ticks per instruction of one wave\n", name, waves_per_simd == 0 ? 1 : waves_per_simd, (double)cyc / instr);
    return 0;
}

int main()
{
    float* d; long long* dc;
    CK(hipMalloc(&d, 1 << 20)); CK(hipMalloc(&dc, 8));
    for (int w : {0, 1, 2, 4}) {
        run<1, false>("fma, 1 dependent chain", w, d, dc);
        run<2, false>("fma, 2 independent chains", w, d, dc);
        run<4, false>("fma, 4 independent chains", w, d, dc);
        run<8, false>("fma, 8 independent chains", w, d, dc);
        run<1, true>("fma + exp2, 1 chain", w, d, dc);
        run<4, true>("fma + exp2, 4 chains", w, d, dc);
    }
    int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    int wclk = 0; hipDeviceGetAttribute(&wclk, hipDeviceAttributeWallClockRate, 0);
    printf("shader clock %d kHz, wall clock %d kHz (clock64 counts shader cycles on gfx9)\n", clk, wclk);
    return 0;
}
