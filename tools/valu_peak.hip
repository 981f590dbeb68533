// Aggregate VALU issue rate of a SIMD (gfx950) by instruction kind and waves per SIMD: what "instruction issue bound" means for the blend kernels.
//   hipcc --offload-arch=gfx950 -O2 tools/valu_peak.hip -o build/valu_peak && build/valu_peak
// Every kernel runs ITER x 64 instructions of one kind per wave from inline asm (8 independent destination registers, so no instruction
// waits for the previous one), on 256 CUs x 4 SIMDs x W waves; time from HIP events -> wave-instructions per SIMD per ns.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

#define R8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
#define R64(OP) R8(OP) R8(OP) R8(OP) R8(OP) R8(OP) R8(OP) R8(OP) R8(OP)

#define FMA(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n\t"
#define MUL(i) "v_mul_f32 %" #i ", %" #i ", %8\n\t"
#define EXP(i) "v_exp_f32 %" #i ", %" #i "\n\t"
#define RCP(i) "v_rcp_f32 %" #i ", %" #i "\n\t"
#define CND(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n\t"
#define CMP(i) "v_cmp_le_f32 vcc, %" #i ", %8\n\t"
#define MIN(i) "v_min_f32 %" #i ", %" #i ", %8\n\t"
#define DPP(i) "v_add_f32_dpp %" #i ", %" #i ", %" #i " row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define SALU(i) "s_add_u32 s" #i ", s" #i ", 1\n\t"
#define CNDS(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, s[20:21]\n\t"
#define CNDI(i) "v_cndmask_b32_e64 %" #i ", 0, %" #i ", s[20:21]\n\t"
#define CMPS(i) "v_cmp_le_f32_e64 s[22:23], %" #i ", %8\n\t"
#define MAX(i) "v_max_f32 %" #i ", %" #i ", %8\n\t"
#define ADD(i) "v_add_f32 %" #i ", %" #i ", %8\n\t"
#define SUB(i) "v_sub_f32 %" #i ", %8, %" #i "\n\t"
#define AND(i) "v_and_b32 %" #i ", %" #i ", %8\n\t"
#define MED(i) "v_med3_f32 %" #i ", %" #i ", %8, %9\n\t"
#define MOV(i) "v_mov_b32 %" #i ", %8\n\t"
#define FMAS(i) "v_fma_f32 %" #i ", %" #i ", s24, %9\n\t"
#define FMAK(i) "v_fmac_f32 %" #i ", %8, %9\n\t"
#define MULL(i) "v_mul_legacy_f32 %" #i ", %" #i ", %8\n\t"
#define PAIRV(i) "v_cmp_le_f32 vcc, %" #i ", %8\n\tv_cndmask_b32 %" #i ", %" #i ", %9, vcc\n\t"
#define PAIRS(i) "v_cmp_le_f32_e64 s[22:23], %" #i ", %8\n\tv_cndmask_b32_e64 %" #i ", %" #i ", %9, s[22:23]\n\t"
#define PAIRV2(i) "v_cmp_le_f32 vcc, %" #i ", %8\n\tv_add_f32 %" #i ", %" #i ", %8\n\tv_cndmask_b32 %" #i ", %" #i ", %9, vcc\n\t"
#define MULS(i) "v_mul_f32 %" #i ", s24, %" #i "\n\t"
#define SUBS(i) "v_sub_f32 %" #i ", s24, %" #i "\n\t"
#define MOVS(i) "v_mov_b32 %" #i ", s24\n\t"
#define MULK(i) "v_mul_f32 %" #i ", 0x3f7d70a4, %" #i "\n\t"
#define MULI(i) "v_mul_f32 %" #i ", 0.5, %" #i "\n\t"
#define MIX(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n\ts_add_u32 s20, s20, 1\n\t"

template <int KIND>
__global__ __launch_bounds__(512) void k(float* out, int iters)
{
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float m = 0.999f, c = 0.0005f;
    asm volatile("s_mov_b64 s[20:21], 0x5555\n\ts_mov_b32 s24, 0x3f7fbe77" ::: "s20", "s21", "s24");
    for (int i = 0; i < iters; ++i) {
#define BODY(OPS, CLOB) asm volatile(R64(OPS) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c) : CLOB)
        if (KIND == 0) BODY(FMA, "memory");
        if (KIND == 1) BODY(MUL, "memory");
        if (KIND == 2) BODY(EXP, "memory");
        if (KIND == 3) BODY(RCP, "memory");
        if (KIND == 4) BODY(CND, "vcc");
        if (KIND == 5) BODY(CMP, "vcc");
        if (KIND == 6) BODY(MIN, "memory");
        if (KIND == 7) BODY(DPP, "memory");
        if (KIND == 8) asm volatile(R64(SALU) ::: "s0", "s1", "s2", "s3", "s4", "s5", "s6", "s7", "scc");
        if (KIND == 10) BODY(CNDS, "s20");
        if (KIND == 11) BODY(CNDI, "s20");
        if (KIND == 12) BODY(CMPS, "s22");
        if (KIND == 13) BODY(MAX, "memory");
        if (KIND == 14) BODY(ADD, "memory");
        if (KIND == 15) BODY(SUB, "memory");
        if (KIND == 16) BODY(AND, "memory");
        if (KIND == 17) BODY(MED, "memory");
        if (KIND == 18) BODY(MOV, "memory");
        if (KIND == 19) BODY(FMAS, "s24");
        if (KIND == 20) BODY(FMAK, "memory");
        if (KIND == 21) BODY(MULL, "memory");
        if (KIND == 22) BODY(PAIRV, "vcc");
        if (KIND == 23) BODY(PAIRS, "s22");
        if (KIND == 24) BODY(PAIRV2, "vcc");
        if (KIND == 25) BODY(MULS, "memory");
        if (KIND == 26) BODY(SUBS, "memory");
        if (KIND == 27) BODY(MOVS, "memory");
        if (KIND == 28) BODY(MULK, "memory");
        if (KIND == 29) BODY(MULI, "memory");
        if (KIND == 9) asm volatile(R64(MIX) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c) : "s20", "scc");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

// packed: v_pk_fma_f32 on register pairs
__global__ __launch_bounds__(512) void k_pk(float* out, int iters)
{
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 a0 = {threadIdx.x * 1e-3f, 1.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    const f2 m = {0.999f, 0.998f}, c = {0.0005f, 0.0004f};
#define PK(i) "v_pk_fma_f32 %" #i ", %" #i ", %8, %9\n\t"
    for (int i = 0; i < iters; ++i)
        asm volatile(R64(PK) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
    const f2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}

template <typename K>
int run(const char* name, K kern, int waves_per_simd, float* d, double instr_per_iter)
{
    const int iters = 2000, cus = 256;
    const int threads = 64 * (waves_per_simd >= 2 ? 8 : 4);           // 512-thread WGs: 2 waves per SIMD each; W = 1: 256 threads
    const int wgs_per_cu = waves_per_simd >= 2 ? waves_per_simd / 2 : 1;
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(kern, dim3(cus * wgs_per_cu), dim3(threads), 0, 0, d, 10);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(kern, dim3(cus * wgs_per_cu), dim3(threads), 0, 0, d, iters);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    const double per_simd = (double)iters * instr_per_iter * waves_per_simd;     // wave-instructions one SIMD issued
    printf("%-28s W=%d  %.3f ns per wave-instruction and SIMD  (%.2f cycles at 2.4 GHz)\n", name, waves_per_simd, ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4);
    return 0;
}

int main()
{
    float* d;
    CK(hipMalloc(&d, 64 << 20));
    for (int w : {1, 4}) {
        run("cmp vcc + cndmask vcc (pair)", k<22>, w, d, 128);
        run("cmp sgpr + cndmask sgpr (pair)", k<23>, w, d, 128);
        run("cmp vcc, add, cndmask vcc", k<24>, w, d, 192);
        run("v_mul_f32 sgpr src0", k<25>, w, d, 64);
        run("v_sub_f32 sgpr src0", k<26>, w, d, 64);
        run("v_mov_b32 v, sgpr", k<27>, w, d, 64);
        run("v_mul_f32 literal", k<28>, w, d, 64);
        run("v_mul_f32 inline const", k<29>, w, d, 64);
    }
    for (int w : {4}) {
        run("v_cndmask_e64 sgpr mask", k<10>, w, d, 64);
        run("v_cndmask_e64 0, v, sgpr", k<11>, w, d, 64);
        run("v_cmp_e64 -> sgpr", k<12>, w, d, 64);
        run("v_max_f32", k<13>, w, d, 64);
        run("v_add_f32", k<14>, w, d, 64);
        run("v_sub_f32", k<15>, w, d, 64);
        run("v_and_b32", k<16>, w, d, 64);
        run("v_med3_f32", k<17>, w, d, 64);
        run("v_mov_b32", k<18>, w, d, 64);
        run("v_fma_f32 sgpr operand", k<19>, w, d, 64);
        run("v_fmac_f32", k<20>, w, d, 64);
        run("v_mul_legacy_f32", k<21>, w, d, 64);
    }
    for (int w : {1, 2, 4, 8}) {
        run("v_fma_f32", k<0>, w, d, 64);
        run("v_mul_f32", k<1>, w, d, 64);
        run("v_pk_fma_f32", k_pk, w, d, 64);
        run("v_exp_f32", k<2>, w, d, 64);
        run("v_rcp_f32", k<3>, w, d, 64);
        run("v_cndmask_b32", k<4>, w, d, 64);
        run("v_cmp_le_f32", k<5>, w, d, 64);
        run("v_min_f32", k<6>, w, d, 64);
        run("v_add_f32_dpp", k<7>, w, d, 64);
        run("s_add_u32", k<8>, w, d, 64);
        run("v_fma + s_add alternating", k<9>, w, d, 128);
    }
    return 0;
}
