// gsr_forward_pc.hip -- the fast blend's forward with every quadrant's walk split over TWO waves (round 6; gfx950, wave64).
//
// k_render<true, 0> (gsr_forward.hip) gives a quadrant's 64 pixels to one wave that walks the quadrant's record stream alone: per record
// ~12 instructions of alpha evaluation (conic, exponential, acceptance) and ~10 of the serial T / C recurrence.  The kernel ends when its
// deepest walk ends, and a wave issues one instruction per ~2.5 ns however empty its SIMD is: the last third of the
// kernel is a few hundred deep walks at ~0.1 us per record with the chip idle.  Handing the REST of a deep stream to other workgroups costs
// more than that tail (state through HBM, ~15 us before a continuation's chain runs).  Here the split is by ROLE, inside
// the workgroup, and needs no state to change hands:
//
//   wave q     (q < 4)  CONSUMER of quadrant q: the recurrence only.  Per-pixel alphas come out of an LDS ring, a record's colour out of
//                       the chunk stage (one broadcast ds_read_b96); T, C, last contributor, checkpoints, tail mode, outputs as before.
//   wave 4 + p          PRODUCER of quadrant (p + 3) & 3 (on another SIMD than that quadrant's consumer when waves are dealt round-robin):
//                       gathers the stream's chunks of 60 records into the stage (as the lone walk did), evaluates alpha of every pixel
//                       of the quadrant for every record, BL records ahead of the consumer at most GSR_PC_RING blocks, and knows nothing
//                       about the pixels' state.
//
// Same expressions, same order per pixel as k_render<true, 0>: the image, the checkpoints and the lists are the SAME BITS (tests/
// test_render_pc_gpu.py).  Total instruction work is the lone walk's plus what a producer evaluates beyond the record its consumer stops
// at (<= GSR_PC_RING blocks); what changes is the length of the dependent chain: a record costs its two waves ~14 issue slots each.
//
// Hand-shake (LDS, per quadrant): s_prod = blocks written, s_cons = blocks taken (bit 31: the consumer is done, stop).  A wave's DS
// operations execute in order, so a block's alphas are in LDS before the count that announces them; the producer may overwrite ring
// slot b % RING once the consumer has TAKEN block b - RING into registers, and stage buffer (c + 1) & 1 once it has taken the first
// block of chunk c (it reads colours of chunk c - 1 until then) -- the ring condition implies it as long as RING < blocks per chunk.
// LDS reads that feed the two loops are issued from inline assembly a batch ahead and waited for with lgkmcnt(0) only (gsr_forward.hip
// explains why the compiler cannot be left to place them); every wait names the registers that landed.
#include "gsr_device.h"
#include <type_traits>

namespace gsr {

#ifndef GSR_PC_BLOCK
#define GSR_PC_BLOCK 12   // records per ring block (6 or 12): what a producer may be ahead is GSR_PC_RING of these
#endif
#ifndef GSR_PC_RING
#define GSR_PC_RING 2
#endif
#ifndef GSR_PC_WAVES_PER_EU
#define GSR_PC_WAVES_PER_EU 6
#endif
#ifndef GSR_PC_SPIN_LIMIT
#define GSR_PC_SPIN_LIMIT (1 << 22)
#endif
#ifndef GSR_FAST_TAIL_LANES
#define GSR_FAST_TAIL_LANES 12
#endif

typedef float pc4 __attribute__((ext_vector_type(4)));
typedef float pc3 __attribute__((ext_vector_type(3)));
typedef float pc2 __attribute__((ext_vector_type(2)));

#ifdef GSR_EXPERIMENT_TIMELINE
__device__ unsigned long long gsr_dbg_pc[4 * 16384];
__device__ unsigned long long gsr_dbg_pc2[4 * 16384];   // per quadrant: consumer (slow blocks << 32 | spins), producer (blocks << 32 | spins), producer start, producer end
extern "C" int gsr_debug_read_pc2(unsigned long long* host, int n) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(gsr_dbg_pc2), (size_t)n * 8); }
#define PC_DBG(x) x
#else
#define PC_DBG(x)
#endif
#ifdef GSR_EXPERIMENT_TIMELINE
extern "C" int gsr_debug_read_pc(unsigned long long* host, int n) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(gsr_dbg_pc), (size_t)n * 8); }
#endif

__global__ __launch_bounds__(512, GSR_PC_WAVES_PER_EU) void k_render_pc(Settings s, const uint32_t* __restrict__ tile_order, const uint32_t* __restrict__ qstart,
                                                                         const uint32_t* __restrict__ qcount, const float4* __restrict__ grec,
                                                                         const uint32_t* __restrict__ qpos, const uint32_t* __restrict__ qlist,
                                                                         float* __restrict__ final_T, uint32_t* __restrict__ n_contrib,
                                                                         uint32_t* __restrict__ n_contrib_q, float* __restrict__ c_final,
                                                                         float4* __restrict__ ck, float* __restrict__ out_color, unsigned long long capacity,
                                                                         const unsigned long long* __restrict__ total_dev, uint32_t* __restrict__ units, int tiles)
{
    if (*total_dev > capacity) return;
    constexpr int CH = GSR_BWD_SEGMENT;            // records per chunk of the stage = the backward's segment
    constexpr int BL = GSR_PC_BLOCK, R = GSR_PC_RING;
    constexpr int TAIL_LANES = GSR_FAST_TAIL_LANES;
    static_assert(BL == 6 || BL == 12, "a block is one or two groups of six records");
    static_assert(CH % BL == 0 && R >= 2 && R * BL < CH, "blocks do not straddle chunks; the ring condition covers the stage");
    // the chunk stage: record j of chunk c at [c & 1][3 j]: (x, y, conic a, conic b) | (conic c, log2 opacity, alpha's upper bound, -) | (r, g, b, -)
    __shared__ float4 stage_all[4][2][CH * 3];
    __shared__ float ring_all[4][R][64 * BL];       // alpha[lane][BL] per block
    __shared__ uint32_t s_prod[4], s_cons[4];
    const int pwave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const bool producer = pwave >= 4;
#ifndef GSR_PC_PROD_SHIFT
#define GSR_PC_PROD_SHIFT 3
#endif
    const int wave = producer ? ((pwave + GSR_PC_PROD_SHIFT) & 3) : pwave;   // the QUADRANT
    if (threadIdx.x < 4) { s_prod[threadIdx.x] = 0u; s_cons[threadIdx.x] = 0u; }
    __syncthreads();
#ifdef GSR_EXPERIMENT_TIMELINE
    const unsigned long long t_start = wall_clock64();
#endif
    const int W = s.W, H = s.H;
    const int gx = (W + GSR_BLOCK_X - 1) / GSR_BLOCK_X;
    const uint32_t lpos = blockIdx.x;
    const int tile = (int)tile_order[blockIdx.x];
    const int tile_x = tile % gx, tile_y = tile / gx;
    const int pxi = tile_x * GSR_BLOCK_X + (wave & 1) * 8 + (lane & 7);
    const int pyi = tile_y * GSR_BLOCK_Y + (wave >> 1) * 8 + (lane >> 3);
    const bool inside = pxi < W && pyi < H;
    const int n = (int)qcount[4 * tile + wave];
    const uint32_t qs = qstart[4 * tile + wave];
    const float4* __restrict__ rec = grec;
    const uint32_t* __restrict__ qp = qpos + qs;
    if (n > 2048) __builtin_amdgcn_s_setprio(3);
    else if (n > 1024) __builtin_amdgcn_s_setprio(2);
    else if (n > 512) __builtin_amdgcn_s_setprio(1);

    float4(*const stage)[CH * 3] = stage_all[wave];
    const uint32_t st0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float4*)&stage[0][0];
    const uint32_t st1 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float4*)&stage[1][0];
    const uint32_t ring0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)&ring_all[wave][0][0];
    constexpr uint32_t SLOT_BYTES = 64u * BL * 4u;
    const uint32_t ring_lane = ring0 + (uint32_t)lane * (BL * 4u);
    const uint32_t prod_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)&s_prod[wave];
    const uint32_t cons_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)&s_cons[wave];

#define PC_WAIT0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define PC_ISSUE_U32(ADDR, V) asm volatile("ds_read_b32 %0, %1" : "=&v"(V) : "v"(ADDR) : "memory")
#define PC_TIE_U32(V) asm volatile("" : "+v"(V))

    if (producer) {
        // =========================================== PRODUCER: alpha of every pixel for every record ===========================================
        if (n == 0) return;
        const float pixx = (float)pxi, pixy = (float)pyi;
        struct PRec { pc4 q0[3]; pc3 q1[3]; };
#define PC_ISSUE_REC(ADDR, O, V)                                                                                                             \
    asm volatile("ds_read_b128 %0, %6 offset:%7\n\tds_read_b96 %1, %6 offset:%7+16\n\tds_read_b128 %2, %6 offset:%7+48\n\t"                \
                 "ds_read_b96 %3, %6 offset:%7+64\n\tds_read_b128 %4, %6 offset:%7+96\n\tds_read_b96 %5, %6 offset:%7+112"                  \
                 : "=&v"(V.q0[0]), "=&v"(V.q1[0]), "=&v"(V.q0[1]), "=&v"(V.q1[1]), "=&v"(V.q0[2]), "=&v"(V.q1[2])                          \
                 : "v"(ADDR), "n"(O)                                                                                                        \
                 : "memory")
// (X: the three alphas of the batch before -- as pass-through operands they pin that batch's arithmetic in front of this wait and of the
//  issue behind it, which reuses the batch's registers; left free, the compiler sinks all four batches of a block below the last wait)
#define PC_READY_REC(V, X)                                                                                                                   \
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(V.q0[0]), "+v"(V.q1[0]), "+v"(V.q0[1]), "+v"(V.q1[1]), "+v"(V.q0[2]), "+v"(V.q1[2]),       \
                 "+v"((X)[0]), "+v"((X)[1]), "+v"((X)[2])::"memory")
        float4 g0, g1, g2;   // the chunk in flight: this lane's record
        auto gather = [&](int c) {   // entries past the end re-read the last one (their alphas are masked)
            const size_t idx = (size_t)qp[min(c * CH + lane, n - 1)];
            g0 = rec[3 * idx + 0];
            g1 = rec[3 * idx + 1];
            g2 = rec[3 * idx + 2];
        };
        auto park = [&](int c) {
            if (lane < CH) {
                float4* d = &stage[c & 1][3 * lane];
                d[0] = g0;
                d[1] = make_float4(g1.x, g1.y, g2.y, 0.0f);
                d[2] = make_float4(g1.z, g1.w, g2.x, 0.0f);
            }
        };
        auto alpha3 = [&](const PRec& V, float* al) {
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const float dx = V.q0[u].x - pixx;
                const float dy = V.q0[u].y - pixy;
                // (k_render<true, 0>'s expressions: the record holds the conic scaled by -log2(e)/2 and L = log2(opacity))
                const float power = __builtin_fmaf(__builtin_fmaf(V.q0[u].w, dy, V.q0[u].z * dx), dx, __builtin_fmaf(V.q1[u].x * dy, dy, V.q1[u].y));
                const float raw = __builtin_amdgcn_exp2f(power);
                const bool ok = __builtin_amdgcn_fmed3f(raw, 1.0f / 255.0f, V.q1[u].z) == raw;
                const float a = __builtin_amdgcn_fmed3f(raw, 0.0f, 0.99f);
                al[u] = ok ? a : 0.0f;
            }
        };
        gather(0);
        park(0);
        if (n > CH) gather(1);
        uint32_t blk = 0;     // blocks written so far
        uint32_t seen = 0;    // the consumer's count as last read
        PRec VA, VB;
        float pin[3] = {0.f, 0.f, 0.f};
        PC_DBG(uint32_t dbg_spins = 0;)
        for (int c = 0;; ++c) {
            const int m = min(CH, n - c * CH);
            const int nblk = (m + BL - 1) / BL;
            uint32_t st = (c & 1) ? st1 : st0;
            PC_ISSUE_REC(st, 0, VA);
            for (int bi = 0; bi < nblk; ++bi) {
                float al[BL];
                uint32_t polled;
                PC_READY_REC(VA, pin);
                PC_ISSUE_REC(st, 144, VB);
                PC_ISSUE_U32(cons_addr, polled);
                alpha3(VA, &al[0]);
                PC_READY_REC(VB, &al[0]);
                PC_TIE_U32(polled);
                if constexpr (BL == 12) {
                    PC_ISSUE_REC(st, 288, VA);
                    alpha3(VB, &al[3]);
                    PC_READY_REC(VA, &al[3]);
                    PC_ISSUE_REC(st, 432, VB);
                    alpha3(VA, &al[6]);
                    PC_READY_REC(VB, &al[6]);
                    if (bi + 1 < nblk) PC_ISSUE_REC(st, 576, VA);
                    alpha3(VB, &al[9]);
                } else {
                    if (bi + 1 < nblk) PC_ISSUE_REC(st, 288, VA);
                    alpha3(VB, &al[3]);
                }
                st += 48u * BL;
                const int jb = c * CH + bi * BL;
                if (jb + BL > n) {   // the stream's last block: nothing behind its end
#pragma unroll
                    for (int u = 0; u < BL; ++u) al[u] = jb + u < n ? al[u] : 0.0f;
                }
                // ring slot blk % R is free once the consumer has taken block blk - R
                seen = (uint32_t)__builtin_amdgcn_readfirstlane((int)polled);
#ifdef GSR_PC_EXP_FREE   // timing experiment: neither side waits for the other (results are garbage)
                seen = 0x7fffffffu;
#endif
                for (int spin = 0; !(seen >> 31) && (int)(seen & 0x7fffffffu) < (int)blk + 1 - R; ++spin) {
                    PC_DBG(++dbg_spins;)
                    if (spin > GSR_PC_SPIN_LIMIT) return;   // (never: the partner wave is resident; a bound keeps a logic error from hanging the GPU)
                    __builtin_amdgcn_s_sleep(2);
                    seen = __hip_atomic_load(&s_cons[wave], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                if (seen >> 31) {   // the consumer stopped (pixels closed, or few enough left for its tail mode)
                    PC_DBG(if (lane == 0 && (size_t)blockIdx.x * 4 + wave < 16384) { unsigned long long* d = gsr_dbg_pc2 + 4 * ((size_t)blockIdx.x * 4 + wave); d[1] = (unsigned long long)blk << 32 | dbg_spins; d[2] = t_start; d[3] = wall_clock64(); })
                    return;
                }
                {
                    float* dst = &ring_all[wave][blk % (uint32_t)R][lane * BL];
#pragma unroll
                    for (int u = 0; u < BL; u += 2) *reinterpret_cast<pc2*>(dst + u) = pc2{al[u], al[u + 1]};
                }
                ++blk;
#ifdef GSR_PC_FLAG_WAIT
                PC_WAIT0();   // (not needed: DS operations of a wave execute in order.  A RELEASE store would also wait for the gather in flight: vmcnt)
#else
                asm volatile("" ::: "memory");
#endif
                if (lane == 0) __hip_atomic_store(&s_prod[wave], blk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            if ((c + 1) * CH >= n) break;
            park(c + 1);
            if ((c + 2) * CH < n) gather(c + 2);
        }
        PC_DBG(if (lane == 0 && (size_t)blockIdx.x * 4 + wave < 16384) { unsigned long long* d = gsr_dbg_pc2 + 4 * ((size_t)blockIdx.x * 4 + wave); d[1] = (unsigned long long)blk << 32 | dbg_spins; d[2] = t_start; d[3] = wall_clock64(); })
        return;
#undef PC_ISSUE_REC
#undef PC_READY_REC
    }

    // =============================================== CONSUMER: the recurrence, the quadrant's outputs ===============================================
    // ONE transmittance register per pixel: Tw is the transmittance while the pixel is open and MINUS the transmittance it ended with once it is
    // closed (outside the image: 0) -- gsr_forward.hip
    float Tw = inside ? 1.0f : 0.0f;
    float C0 = 0.f, C1 = 0.f, C2 = 0.f;
    uint32_t last_q = 0;
    int lq = 0;   // last contributor, relative to the walk position
    int j0 = 0;
    const float pixx = (float)pxi, pixy = (float)pyi;
    const uint32_t ucap = (uint32_t)unit_list_cap((size_t)tiles);

    struct Col { pc3 c[3]; };
    struct Alp { pc2 a[BL / 2]; };
#define PC_ISSUE_COL(ADDR, O, C)                                                                                                             \
    asm volatile("ds_read_b96 %0, %8 offset:%9+32\n\tds_read_b96 %1, %8 offset:%9+80\n\tds_read_b96 %2, %8 offset:%9+128"                    \
                 : "=&v"(C.c[0]), "=&v"(C.c[1]), "=&v"(C.c[2]), /* issued BEHIND the blend before it (pass-through state) */                  \
                   "+v"(Tw), "+v"(C0), "+v"(C1), "+v"(C2), "+v"(lq)                                                                         \
                 : "v"(ADDR), "n"(O)                                                                                                        \
                 : "memory")
#define PC_READY_COL(C) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(C.c[0]), "+v"(C.c[1]), "+v"(C.c[2])::"memory")
#define PC_ISSUE_ALP3(ADDR, O, A, K)                                                                                                         \
    asm volatile("ds_read_b64 %0, %3 offset:%4\n\tds_read_b64 %1, %3 offset:%4+8\n\tds_read_b64 %2, %3 offset:%4+16"                         \
                 : "=&v"(A.a[K]), "=&v"(A.a[K + 1]), "=&v"(A.a[K + 2])                                                                      \
                 : "v"(ADDR), "n"(O)                                                                                                        \
                 : "memory")
#define PC_TIE_ALP3(A, K) asm volatile("" : "+v"(A.a[K]), "+v"(A.a[K + 1]), "+v"(A.a[K + 2]))
    auto issue_alphas = [&](uint32_t block, Alp& A) {
        const uint32_t addr = ring_lane + (block % (uint32_t)R) * SLOT_BYTES;
        PC_ISSUE_ALP3(addr, 0, A, 0);
        if constexpr (BL == 12) PC_ISSUE_ALP3(addr, 24, A, 3);
    };
    auto tie_alphas = [&](Alp& A) {
        PC_TIE_ALP3(A, 0);
        if constexpr (BL == 12) PC_TIE_ALP3(A, 3);
    };

    auto blend3 = [&](const float a0, const float a1, const float a2, const Col& C, auto off) {
        constexpr int OFF = decltype(off)::value;
        const float al[3] = {a0, a1, a2};
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const float test_T = __builtin_fmaf(-al[u], Tw, Tw);   // alpha == 0 (skipped record): exactly Tw, accepted, nothing changes
            const bool keep = test_T >= 0.0001f;                    // false at the record that saturates the pixel, and forever after
            const float ae = keep ? al[u] : 0.0f;
            const float wgt = ae * Tw;
            C0 = __builtin_fmaf(C.c[u].x, wgt, C0);
            C1 = __builtin_fmaf(C.c[u].y, wgt, C1);
            C2 = __builtin_fmaf(C.c[u].z, wgt, C2);
            Tw = keep ? test_T : -__builtin_fabsf(Tw);
            lq = ae > 0.0f ? OFF + u + 1 : lq;                       // (an accepted alpha is >= 1/255)
        }
    };
    using Off0 = std::integral_constant<int, 0>;
    using Off1 = std::integral_constant<int, 3>;

    auto keep_going = [&](int jb) {
        const unsigned long long open_mask = __ballot(Tw > 0.0f);
        int open;
        asm("s_bcnt1_i32_b64 %0, %1" : "=s"(open) : "s"(open_mask) : "scc");
        if (open > TAIL_LANES) return true;
        return open != 0 && n - jb <= 2 * GSR_WAVE;   // nothing open: stop; few open pixels and a long way to go: tail mode
    };
    const size_t HWs = (size_t)H * W;
    float4* ck_ptr = ck + (inside ? (size_t)(W * pyi + pxi) : (size_t)(GSR_BWD_SEGMENTS - 1) * HWs);
    const size_t ck_step = inside ? HWs : 0;
    int next_ck = s.forward_only ? 0x7fffffff : GSR_BWD_SEGMENT;
    auto checkpoint = [&](int jtop) {
        if (jtop == next_ck) {
            if (next_ck <= (GSR_BWD_SEGMENTS - 1) * GSR_BWD_SEGMENT) {
                *ck_ptr = make_float4(__builtin_fabsf(Tw), C0, C1, C2);
                ck_ptr += ck_step;
            }
            next_ck += GSR_BWD_SEGMENT;
        }
    };

    PC_DBG(uint32_t dbg_slow = 0; uint32_t dbg_cspins = 0;)
    if (n > 0) {
        // Per block of BL records: [flag, the block's alphas, its first three colours] were requested during the block before, in that order (a
        // wave's DS operations execute in order: alphas read behind a flag that announces them are the announced ones); if the flag says
        // the producer had not got that far, wait for it and ask again.  Within a block every address is the block's base plus an immediate.
        uint32_t blk = 0;         // the block being consumed
        uint32_t stg = st0;       // LDS address of the block's first record
        int bic = 0;              // the block's index in its chunk
        uint32_t odd = 0;         // the chunk's buffer
        uint32_t flag;
        PC_DBG(dbg_slow = 0; dbg_cspins = 0;)
        Alp AA, AB;
        Col CA, CB;
        auto request = [&](uint32_t block, uint32_t at, Alp& A) {   // flag first
            PC_ISSUE_U32(prod_addr, flag);
            issue_alphas(block, A);
            PC_ISSUE_COL(at, 0, CA);
        };
        // one group of six records (two batches of three) at stg + 288 g; true: the walk ends here
        auto group = [&](const float a0, const float a1, const float a2, const float a3, const float a4, const float a5, auto g_t, auto last_t, uint32_t nstg,
                         Alp& nxt) -> bool {
            constexpr int G = decltype(g_t)::value;
            constexpr bool last = decltype(last_t)::value;
            if (j0 >= n) return true;
            if (!keep_going(j0)) return true;
            PC_ISSUE_COL(stg, 288 * G + 144, CB);
            blend3(a0, a1, a2, CA, Off0{});
            PC_READY_COL(CB);
            if constexpr (last) request(blk + 1u, nstg, nxt);   // (past the stream's end: harmless, unused)
            else PC_ISSUE_COL(stg, 288 * G + 288, CA);
            blend3(a3, a4, a5, CB, Off1{});
            j0 += 6;
            lq -= 6;
            return false;
        };
        auto block = [&](Alp& cur, Alp& nxt) -> bool {
            PC_READY_COL(CA);
            tie_alphas(cur);
            PC_TIE_U32(flag);
#ifdef GSR_PC_EXP_FREE
            flag = 0x7fffffffu;
#endif
            if ((uint32_t)__builtin_amdgcn_readfirstlane((int)flag) <= blk) {   // not announced when the flag was read: wait for it, ask again
                PC_DBG(++dbg_slow;)
                for (int spin = 0; __hip_atomic_load(&s_prod[wave], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) <= blk; ++spin) {
                    PC_DBG(++dbg_cspins;)
                    if (spin > GSR_PC_SPIN_LIMIT) return true;
                    __builtin_amdgcn_s_sleep(1);
                }
                request(blk, stg, cur);
                PC_READY_COL(CA);
                tie_alphas(cur);
                PC_TIE_U32(flag);
            }
            if (lane == 0) __hip_atomic_store(&s_cons[wave], blk + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // taken: the slot is the producer's again
            checkpoint(j0);   // (chunks begin on blocks)
            uint32_t nstg = stg + 48u * BL;
            if (++bic == CH / BL) {
                bic = 0;
                odd ^= 1u;
                nstg = odd ? st1 : st0;
            }
            if constexpr (BL == 12) {
                if (group(cur.a[0].x, cur.a[0].y, cur.a[1].x, cur.a[1].y, cur.a[2].x, cur.a[2].y, std::integral_constant<int, 0>{}, std::false_type{}, nstg, nxt)) return true;
                PC_READY_COL(CA);
                if (group(cur.a[3].x, cur.a[3].y, cur.a[4].x, cur.a[4].y, cur.a[5].x, cur.a[5].y, std::integral_constant<int, 1>{}, std::true_type{}, nstg, nxt)) return true;
            } else {
                if (group(cur.a[0].x, cur.a[0].y, cur.a[1].x, cur.a[1].y, cur.a[2].x, cur.a[2].y, std::integral_constant<int, 0>{}, std::true_type{}, nstg, nxt)) return true;
            }
            stg = nstg;
            ++blk;
            return j0 >= n;
        };
        request(0u, stg, AA);
        for (;;) {
            if (block(AA, AB)) break;
            if (block(AB, AA)) break;
        }
        PC_WAIT0();   // (whatever was requested ahead has landed before its registers are anybody else's)
        if (lane == 0) __hip_atomic_store(&s_cons[wave], 0x80000000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // the producer may stop
    }
    last_q = (uint32_t)(lq + j0);   // (lq + j0 >= 0: a pixel without a hit kept lq = -j0)
    if (j0 < n) checkpoint(j0);
#ifdef GSR_EXPERIMENT_TIMELINE
    const unsigned long long t_main = wall_clock64();
    const int j_main = j0;
#endif

    // ---- tail mode (k_render<true, 0>'s): chunks of the stream up to the next checkpoint boundary, lanes = records, the open pixels two at a time
    if (j0 < n) {
        unsigned long long open_mask = __ballot(Tw > 0.0f);
        int c0 = j0;
        while (c0 < n && open_mask) {
            const int c1 = min(n, (c0 / GSR_BWD_SEGMENT + 1) * GSR_BWD_SEGMENT);
            const int j = c0 + lane;
            const bool valid = j < c1;
            const size_t jc = (size_t)qp[valid ? j : c1 - 1];
            const float4 r0 = rec[3 * jc + 0];
            const float4 r1 = rec[3 * jc + 1];
            const float4 r2 = rec[3 * jc + 2];
            unsigned long long todo = open_mask, closed_mask = 0ull;
            while (todo) {
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
                    ok[e] = valid && __builtin_amdgcn_fmed3f(raw, 1.0f / 255.0f, r2.y) == raw;
                    const float a = __builtin_amdgcn_fmed3f(raw, 0.0f, 0.99f);
                    am[e] = ok[e] ? a : 0.0f;
                    prod[e] = 1.0f - am[e];
                    okf[e] = am[e] * __builtin_amdgcn_rcpf(prod[e]);   // alpha / (1 - alpha)
                }
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
                    Tfull[e] = Tp[e] * prod[e];
                    keepl[e] = Tfull[e] >= 0.0001f;
                    K[e] = __ballot(keepl[e]);
                    hits[e] = __ballot(ok[e] && keepl[e]);
                    const float wgt = keepl[e] ? Tfull[e] * okf[e] : 0.0f;
                    s3[e][0] = r1.z * wgt; s3[e][1] = r1.w * wgt; s3[e][2] = r2.x * wgt;
                }
                float h3[3];
#pragma unroll
                for (int cc = 0; cc < 3; ++cc) {
                    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(s3[0][cc]), __float_as_uint(s3[1][cc]), false, false);
                    float v = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
                    v += dpp_f<0xB1, 0xf>(v);
                    v += dpp_f<0x4E, 0xf>(v);
                    v += dpp_f<0x141, 0xf>(v);
                    v += dpp_f<0x140, 0xf>(v);
                    v += dpp_f<0x142, 0xa>(v);
                    h3[cc] = v;
                }
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    if (e == 1 && !two) break;
                    const int src = e ? 63 : 31;
                    const float A0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(h3[0]), src));
                    const float A1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(h3[1]), src));
                    const float A2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(h3[2]), src));
                    const bool closed = K[e] != ~0ull;
                    const int f = closed ? __builtin_ctzll(~K[e]) : 64;
                    const float Tlast = f == 0 ? Tp[e] : __int_as_float(__builtin_amdgcn_readlane(__float_as_int(Tfull[e]), f - 1));
                    if (lane == pp[e]) {
                        Tw = closed ? -Tlast : Tlast;
                        C0 += A0; C1 += A1; C2 += A2;
                        if (hits[e]) last_q = (uint32_t)(c0 + 64 - __builtin_clzll(hits[e]));
                    }
                    if (closed) closed_mask |= 1ull << pp[e];
                }
            }
            open_mask &= ~closed_mask;
            c0 = c1;
            if (c0 < n) checkpoint(c0);
        }
    }
#ifdef GSR_EXPERIMENT_TIMELINE
    if (lane == 0 && (size_t)blockIdx.x * 4 + wave < 16384) {
        unsigned long long* d = gsr_dbg_pc + 4 * ((size_t)blockIdx.x * 4 + wave);
        d[0] = t_start;
        d[1] = wall_clock64();
        d[2] = ((unsigned long long)(uint32_t)n << 32) | (uint32_t)j_main;
        d[3] = t_main;
        gsr_dbg_pc2[4 * ((size_t)blockIdx.x * 4 + wave)] = (unsigned long long)dbg_slow << 32 | dbg_cspins;
    }
#endif

    // ---- the end of a quadrant: the backward's work units, the per-pixel state the backward reads, the image (k_render's finish()) ----
    if (!s.forward_only) {
        const uint32_t qmax = wave_max_u32(inside ? last_q : 0u);
        const uint32_t nseg = min((qmax + GSR_BWD_SEGMENT - 1u) / GSR_BWD_SEGMENT, (uint32_t)GSR_BWD_SEGMENTS);
        if (nseg) {
            const uint32_t w = lpos * 4u + (uint32_t)wave, list = w % (uint32_t)GSR_UNIT_LISTS;
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(units + 32u * list, nseg);
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            if ((uint32_t)lane < nseg)
                units[32u * GSR_UNIT_LISTS + list * ucap + base + (uint32_t)lane] = (uint32_t)tile << 6 | (uint32_t)wave << 4 | (uint32_t)lane;
        }
    }
    if (inside) {
        const int pix_id = W * pyi + pxi;
        const size_t HW = (size_t)H * W;
        if (!s.forward_only) {
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
    (void)pixx; (void)pixy;
}

}  // namespace gsr
