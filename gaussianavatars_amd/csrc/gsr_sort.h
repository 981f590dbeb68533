// gsr_sort.h -- register-resident bitonic block sort of 64-bit keys (shared by the per-tile sort of the parity modes and the
// per-bucket depth sort of the production binning).
#pragma once
#include "gsr_device.h"

namespace gsr {

// ------------------------------------------------------------------------------------------
// Register-resident block sort for the in-LDS size classes: 8 keys per thread (index i = 8*tid + k).
// Normalised bitonic network (every compare-exchange puts the smaller key at the lower index, first step
// of a merge pairs i with i ^ (size-1), the rest with i ^ j), but a compare-exchange whose partner index
// i ^ M differs only in the low 3 bits is done in registers, one that differs in lane bits goes through
// the cross-lane network (ds_bpermute), and only masks reaching across waves (M >= 512) touch LDS with
// a barrier: 10 barrier steps instead of 91 for 8192 keys.  Every mask is a compile-time constant, so
// the key array stays in VGPRs.  Slots >= n hold the +inf pattern (GSR_SORT_PAD) and sink to the end.
// ------------------------------------------------------------------------------------------
typedef unsigned long long u64;
__device__ __forceinline__ u64 shfl64(u64 v, int src_lane)
{
    const int lo = __builtin_amdgcn_ds_bpermute(src_lane << 2, (int)(uint32_t)v);
    const int hi = __builtin_amdgcn_ds_bpermute(src_lane << 2, (int)(uint32_t)(v >> 32));
    return ((u64)(uint32_t)hi << 32) | (u64)(uint32_t)lo;
}
constexpr int top_bit(int m) { int b = 1; while ((b << 1) <= m) b <<= 1; return b; }

// Keys are compared as DOUBLES: a key is (depth bits << 32 | splat) with the depth a positive finite float, so its high word
// is below 0x7F800000 and the 64-bit pattern is a positive finite (possibly denormal: FP64 denormals are never flushed)
// double whose order is the order of the unsigned integers; padding slots hold +inf.  A compare-exchange is then
// v_min_f64 + v_max_f64 (full-rate on CDNA4) instead of a 64-bit integer compare and four selects.
__device__ __forceinline__ u64 kmin(u64 a, u64 b)
{
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(__builtin_bit_cast(double, a)), "v"(__builtin_bit_cast(double, b)));
    return __builtin_bit_cast(u64, r);
}
__device__ __forceinline__ u64 kmax(u64 a, u64 b)
{
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(__builtin_bit_cast(double, a)), "v"(__builtin_bit_cast(double, b)));
    return __builtin_bit_cast(u64, r);
}
#define GSR_SORT_PAD 0x7FF0000000000000ull   // +inf: sinks to the end

template <int M, int THREADS, int EPT>
__device__ __forceinline__ void cx_step(u64 (&key)[EPT], u64* __restrict__ sk, int tid)
{
    constexpr int LE = EPT == 16 ? 4 : 3;                 // log2(keys per thread)
    constexpr int KM = M & (EPT - 1);
    constexpr int LM = (M >> LE) & 63;
    constexpr int WM = M >> (LE + 6);
    if constexpr (LM == 0 && WM == 0) {
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
            if ((k ^ KM) > k) {
                const u64 a = key[k], b = key[k ^ KM];
                key[k] = kmin(a, b);
                key[k ^ KM] = kmax(a, b);
            }
        }
    } else {
        constexpr int TOP = top_bit(M);                  // >= EPT here: decided by the thread id alone
        // lanes that keep the larger key negate both operands (sign bit of the double), take the minimum and negate back
        const u64 flip = (tid & (TOP >> LE)) ? 0x8000000000000000ull : 0ull;
        u64 other[EPT];
        if constexpr (WM == 0) {
            const int pl = (tid & 63) ^ LM;
#pragma unroll
            for (int k = 0; k < EPT; ++k) other[k] = shfl64(key[k ^ KM], pl);
        } else {
            // staging layout sk[k * THREADS + tid]: consecutive lanes hit consecutive banks
#pragma unroll
            for (int k = 0; k < EPT; ++k) sk[k * THREADS + tid] = key[k];
            __syncthreads();
            const int pt = tid ^ (M >> LE);
#pragma unroll
            for (int k = 0; k < EPT; ++k) other[k] = sk[(k ^ KM) * THREADS + pt];
            __syncthreads();
        }
#pragma unroll
        for (int k = 0; k < EPT; ++k) key[k] = kmin(key[k] ^ flip, other[k] ^ flip) ^ flip;
    }
}

template <int J, int THREADS, int EPT>
__device__ __forceinline__ void cx_tail(u64 (&key)[EPT], u64* __restrict__ sk, int tid)
{
    if constexpr (J > 0) {
        cx_step<J, THREADS, EPT>(key, sk, tid);
        cx_tail<(J >> 1), THREADS, EPT>(key, sk, tid);
    }
}
template <int SIZE, int N, int THREADS, int EPT>
__device__ __forceinline__ void cx_stage(u64 (&key)[EPT], u64* __restrict__ sk, int tid, uint32_t n)
{
    if constexpr (SIZE <= N) {
        // a merge of blocks of SIZE/2 has nothing to do once the first block holds every real key (the rest is +inf padding):
        // the network stops at the first power of two >= n instead of at the class size (workgroup-uniform test)
        if ((uint32_t)(SIZE / 2) >= n) return;
        cx_step<SIZE - 1, THREADS, EPT>(key, sk, tid);            // first step of a merge: partner = i ^ (size - 1)
        cx_tail<(SIZE >> 2), THREADS, EPT>(key, sk, tid);         // then i ^ j for j = size/4 ... 1
        cx_stage<(SIZE << 1), N, THREADS, EPT>(key, sk, tid, n);
    }
}

// sorts seg[0..n) (n <= EPT*THREADS); thread t ends up holding sorted positions EPT*t .. EPT*t+EPT-1 in key[]
template <int THREADS, int EPT>
__device__ __forceinline__ void block_sort_regs(u64 (&key)[EPT], u64* __restrict__ sk, const u64* __restrict__ seg, uint32_t n, int tid)
{
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
        const uint32_t i = (uint32_t)tid * (uint32_t)EPT + (uint32_t)k;
        key[k] = i < n ? seg[i] : GSR_SORT_PAD;
    }
    cx_stage<2, EPT * THREADS, THREADS, EPT>(key, sk, tid, n);
}

// ------------------------------------------------------------------------------------------
// Oversize tiles (more entries than the LDS class holds): sort KEYS-sized chunks with the register sort,
// then merge the runs pairwise in global memory (merge path: every thread binary-searches its diagonal
// and merges a private output slice).  `tmp` is scratch of at least n keys (the tile's still-unused
// quadrant-record region); the sorted result always ends in seg[0..n).  Keys are unique per tile.
// ------------------------------------------------------------------------------------------
template <int THREADS, int EPT>
__device__ __forceinline__ void oversize_sort(u64* __restrict__ seg, u64* __restrict__ tmp, u64* __restrict__ sk, uint32_t n, int tid)
{
    constexpr uint32_t CH = (uint32_t)(THREADS * EPT);
    for (uint32_t c0 = 0; c0 < n; c0 += CH) {
        const uint32_t m = min(CH, n - c0);
        u64 key[EPT];
        block_sort_regs<THREADS, EPT>(key, sk, seg + c0, m, tid);
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
            const uint32_t i = (uint32_t)tid * (uint32_t)EPT + (uint32_t)k;
            if (i < m) tmp[c0 + i] = key[k];
        }
        __syncthreads();
    }
    u64* src = tmp;
    u64* dst = seg;
    for (uint32_t w = CH; w < n; w <<= 1) {
        for (uint32_t p0 = 0; p0 < n; p0 += 2 * w) {
            const u64* A = src + p0;
            const uint32_t na = min(w, n - p0);
            const u64* B = A + na;
            const uint32_t nb = (p0 + na < n) ? min(w, n - p0 - na) : 0u;
            const uint32_t total = na + nb;
            const uint32_t S = (total + THREADS - 1) / THREADS;
            const uint32_t d0 = min((uint32_t)tid * S, total), d1 = min(d0 + S, total);
            // merge path: a = number of A elements among the first d0 outputs
            uint32_t lo = d0 > nb ? d0 - nb : 0u, hi = min(d0, na);
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (A[mid] < B[d0 - 1 - mid]) lo = mid + 1; else hi = mid;
            }
            uint32_t a = lo, b = d0 - lo;
            for (uint32_t o = d0; o < d1; ++o) {
                const bool takeA = b >= nb || (a < na && A[a] < B[b]);
                dst[p0 + o] = takeA ? A[a] : B[b];
                a += takeA ? 1u : 0u;
                b += takeA ? 0u : 1u;
            }
        }
        __syncthreads();
        u64* t = src; src = dst; dst = t;
    }
    if (src != seg) {   // odd number of passes (or none): bring the result home
        for (uint32_t i = tid; i < n; i += THREADS) seg[i] = src[i];
        __syncthreads();
    }
}

}  // namespace gsr
