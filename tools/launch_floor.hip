// Per-kernel floor of dependent tiny kernels on one stream, plain launches vs one hipGraph launch:  hipcc --offload-arch=gfx950 -O2 tools/launch_floor.hip -o build/launch_floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_tiny(float* p, int n) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1.f; }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main()
{
    const int N = 200;
    float* d; CK(hipMalloc(&d, 1 << 24)); CK(hipMemset(d, 0, 1 << 24));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int sz : {64, 65536, 1 << 20}) {
        const int blocks = (sz + 255) / 256;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_tiny, dim3(blocks), dim3(256), 0, s, d, sz);
            CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep == 2) printf("n=%8d plain : %.2f us per kernel\n", sz, 1e3 * ms / N);
        }
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_tiny, dim3(blocks), dim3(256), 0, s, d, sz);
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0, s));
            CK(hipGraphLaunch(ge, s));
            CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep == 2) printf("n=%8d graph : %.2f us per kernel\n", sz, 1e3 * ms / N);
        }
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
