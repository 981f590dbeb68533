// gsr_binning.hip -- production binning of the MI355X-native splat rasterizer (gfx950, wave64): depth-ordered scatter
// straight into the 8x8-quadrant streams the blend kernels walk.
//
// The reference sorts one 64-bit key per (splat, tile) INSTANCE (A.2); round 1 did the same per tile in LDS.  At 100 k
// splats that is 1.5-2.2 M keys, and 80 % of the resulting stream entries are never read (they sit behind the surface
// that saturates the pixels).  The order those lists need is only the splats' depth order, so here the SPLATS are sorted
// (17x fewer keys) and every splat is then appended, in that order, to the streams of the quadrants its {alpha >= 1/255}
// ellipse can reach -- a stable counting sort by quadrant of a sequence emitted in depth order:
//
//   k_dbucket    per splat : depth bucket histogram (monotone linear map of this frame's depth range)
//   k_dscan      1 WG      : bucket offsets, bucket launch order
//   k_dscatter   per splat : (depth bits << 32 | splat) into its bucket
//   k_dsort      per bucket: register bitonic sort (gsr_sort.h) -> order[] = splat indices by (depth, index)
//   k_qcount     per chunk : one wave walks its S consecutive splats of order[], 64 lanes = the quadrants of ONE splat's
//                            snug rect; exact reach test per lane; per-quadrant counts in LDS (bytes, 4 per word)
//   k_qscan      per 64 q  : exclusive prefix over the chunks of every quadrant -> qprefix[chunk][q], totals
//   k_qscan_glob 1 WG      : stream starts, per-tile launch order (heaviest first), instance count posted to the host
//   k_qscatter   per chunk : same walk; slot = qstart[q] + qprefix[chunk][q] + (LDS cursor++): a stable append, no atomics
//                            between waves, no sort, no per-instance record gather
//
// Streams come out exactly as the parity path's (k_tile_sort epilogue): the entries of quadrant q are the splats whose tile-level
// snug rect holds q's tile and whose ellipse reaches q, in (depth, index) order -- tests compare them entry for entry.
#include "gsr_device.h"
#include "gsr_sort.h"

namespace gsr {

// ---- depth buckets ------------------------------------------------------------------------------------------------
// bucket = clamp((depth - lo) * scale): monotone in depth, so sorting every bucket on its own sorts everything.
__device__ __forceinline__ uint32_t depth_bucket(float depth, float lo, float scale, uint32_t nb)
{
    const float f = (depth - lo) * scale;
    const int b = f2i_sat(f);
    return (uint32_t)min(max(b, 0), (int)nb - 1);
}
__device__ __forceinline__ void bucket_map(uint32_t dmin_inv, uint32_t dmax_bits, uint32_t nb, float& lo, float& scale)
{
    lo = __uint_as_float(~dmin_inv);
    const float hi = __uint_as_float(dmax_bits);
    const float span = hi - lo;
    scale = span > 0.f ? (float)nb * 0.99999f / span : 0.f;   // every depth of the frame lands in [0, nb)
}

// <= GSR_BIN_BLOCKS workgroups, each a contiguous chunk of splats: bucket histogram in LDS (depth is concentrated on the visible
// surface: a few buckets are hot), non-empty bins to the global counters, the per-workgroup histogram kept for k_dscatter.
// The prologue folds k_preprocess's statistics slots (every workgroup needs the depth range; workgroup 0 publishes the header).
__global__ __launch_bounds__(256) void k_dbucket(int P, const uint32_t* __restrict__ brec_rect, const float* __restrict__ depths,
                                                  BinHeader* __restrict__ hdr, uint32_t nb, uint32_t* __restrict__ bcount,
                                                  uint32_t* __restrict__ bhist)
{
    extern __shared__ uint32_t hist[];
    __shared__ uint32_t s_rng[2];
    const int tid = threadIdx.x;
    if (tid < 64) {
        const BinStatSlot* sl = hdr->slot + (tid & (GSR_STAT_SLOTS - 1));
        const bool mine = tid < GSR_STAT_SLOTS;
        const uint32_t mn = wave_max_u32(mine ? sl->dmin_inv : 0u), mx = wave_max_u32(mine ? sl->dmax_bits : 0u);
        unsigned long long nv = mine ? sl->nvis : 0u, bt = mine ? sl->binned_tiles : 0ull, rt = mine ? sl->rect_total : 0ull;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { nv += __shfl_xor(nv, d, 64); bt += __shfl_xor(bt, d, 64); rt += __shfl_xor(rt, d, 64); }
        if (tid == 0) {
            s_rng[0] = mn; s_rng[1] = mx;
            if (blockIdx.x == 0) {
                hdr->nvis = (uint32_t)nv; hdr->dmin_inv = mn; hdr->dmax_bits = mx; hdr->dmin_bits = ~mn;
                hdr->binned_tiles = bt; hdr->rect_total = rt;
            }
        }
    }
    for (uint32_t t = tid; t < nb; t += 256) hist[t] = 0u;
    __syncthreads();
    float lo, scale;
    bucket_map(s_rng[0], s_rng[1], nb, lo, scale);
    const int chunk = ((P + (int)gridDim.x - 1) / (int)gridDim.x + 255) / 256 * 256;
    const int begin = blockIdx.x * chunk, end = min(P, begin + chunk);
    for (int i = begin + tid; i < end; i += 256) {
        if (brec_rect[12 * (size_t)i + 8] == brec_rect[12 * (size_t)i + 9]) continue;   // empty quadrant rect: not binned
        atomicAdd(&hist[depth_bucket(depths[i], lo, scale, nb)], 1u);
    }
    __syncthreads();
    uint32_t* __restrict__ mine = bhist + (size_t)blockIdx.x * nb;
    for (uint32_t t = tid; t < nb; t += 256) {
        const uint32_t v = hist[t];
        mine[t] = v;
        if (v) atomicAdd(&bcount[t], v);
    }
}

// one workgroup: exclusive scan of the bucket counts; buckets listed heaviest-first (the dispatcher hands out workgroups in order)
__global__ __launch_bounds__(1024) void k_dscan(uint32_t nb, const uint32_t* __restrict__ bcount, uint32_t* __restrict__ bstart,
                                                 uint32_t* __restrict__ bcursor, uint32_t* __restrict__ border, BinHeader* __restrict__ hdr)
{
    __shared__ uint32_t wave_tot[16];
    __shared__ uint32_t carry_s;
    __shared__ uint32_t bucket[34];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (tid == 0) carry_s = 0;
    (void)hdr;
    __syncthreads();
    for (uint32_t base = 0; base < nb; base += 1024) {
        const uint32_t t = base + tid;
        const uint32_t v = t < nb ? bcount[t] : 0u;
        uint32_t incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = __shfl_up(incl, d, 64);
            if (lane >= d) incl += up;
        }
        if (lane == 63) wave_tot[wid] = incl;
        __syncthreads();
        uint32_t wave_off = 0;
        for (int w = 0; w < wid; ++w) wave_off += wave_tot[w];
        const uint32_t carry = carry_s;
        if (t < nb) {
            bstart[t] = carry + wave_off + incl - v;
            bcursor[t] = 0u;
        }
        __syncthreads();
        if (tid == 1023) carry_s = carry + wave_off + incl;
        __syncthreads();
    }
    if (tid < 34) bucket[tid] = 0u;
    __syncthreads();
    auto bucket_of = [](uint32_t c) { return c ? 32u - (uint32_t)(31 - __builtin_clz(c)) - 1u : 32u; };
    for (uint32_t t = tid; t < nb; t += 1024) atomicAdd(&bucket[bucket_of(bcount[t])], 1u);
    __syncthreads();
    if (tid == 0) {
        uint32_t run = 0;
        for (int b = 0; b < 33; ++b) { const uint32_t c = bucket[b]; bucket[b] = run; run += c; }
    }
    __syncthreads();
    for (uint32_t t = tid; t < nb; t += 1024) border[atomicAdd(&bucket[bucket_of(bcount[t])], 1u)] = t;
}

// same chunking as k_dbucket: one returning L2 atomic per (workgroup, non-empty bucket) reserves a sub-range, slots from LDS cursors
__global__ __launch_bounds__(256) void k_dscatter(int P, const uint32_t* __restrict__ brec_rect, const float* __restrict__ depths,
                                                   const BinHeader* __restrict__ hdr, uint32_t nb, const uint32_t* __restrict__ bstart,
                                                   uint32_t* __restrict__ bcursor, unsigned long long* __restrict__ dkeys,
                                                   const uint32_t* __restrict__ bhist)
{
    extern __shared__ uint32_t hist[];
    const int tid = threadIdx.x;
    const uint32_t* __restrict__ mine = bhist + (size_t)blockIdx.x * nb;
    for (uint32_t t = tid; t < nb; t += 256) {
        const uint32_t v = mine[t];
        hist[t] = v ? bstart[t] + atomicAdd(&bcursor[t], v) : 0u;
    }
    __syncthreads();
    float lo, scale;
    bucket_map(hdr->dmin_inv, hdr->dmax_bits, nb, lo, scale);
    const int chunk = ((P + (int)gridDim.x - 1) / (int)gridDim.x + 255) / 256 * 256;
    const int begin = blockIdx.x * chunk, end = min(P, begin + chunk);
    for (int i = begin + tid; i < end; i += 256) {
        if (brec_rect[12 * (size_t)i + 8] == brec_rect[12 * (size_t)i + 9]) continue;
        const float d = depths[i];
        const uint32_t slot = atomicAdd(&hist[depth_bucket(d, lo, scale, nb)], 1u);
        dkeys[slot] = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned long long)(uint32_t)i;
    }
}

// per bucket: sort the keys by (depth, splat), write the splat indices to order[] at the bucket's offset.  Two size classes
// (<= 2048 keys: 256 threads; larger: 1024 threads x 16 keys with chunked sorts + merges beyond 16384; scratch for those = tmp)
template <int KEYS, int THREADS>
__global__ __launch_bounds__(THREADS) void k_dsort(uint32_t n_lo, uint32_t n_hi, const uint32_t* __restrict__ border,
                                                    const uint32_t* __restrict__ bcount, const uint32_t* __restrict__ bstart,
                                                    unsigned long long* __restrict__ dkeys, unsigned long long* __restrict__ tmp,
                                                    uint32_t* __restrict__ order)
{
    __shared__ unsigned long long skeys[KEYS];
    constexpr int EPT = KEYS / THREADS;
    const uint32_t b = border[blockIdx.x];
    const uint32_t n = bcount[b];
    if (n <= n_lo || n > n_hi) return;
    const uint32_t start = bstart[b];
    const int tid = threadIdx.x;
    unsigned long long* seg = dkeys + start;
    if (n <= (uint32_t)KEYS) {
        u64 key[EPT];
        block_sort_regs<THREADS, EPT>(key, skeys, seg, n, tid);
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
            const uint32_t i = (uint32_t)tid * (uint32_t)EPT + (uint32_t)k;
            if (i < n) order[start + i] = (uint32_t)key[k];
        }
    } else {
        oversize_sort<THREADS, EPT>(seg, tmp + start, skeys, n, tid);
        __syncthreads();
        for (uint32_t i = tid; i < n; i += THREADS) order[start + i] = (uint32_t)seg[i];
    }
}
template __global__ void k_dsort<GSR_SORT_SMALL_KEYS, 256>(uint32_t, uint32_t, const uint32_t*, const uint32_t*, const uint32_t*,
                                                            unsigned long long*, unsigned long long*, uint32_t*);

// ---- the walk over a chunk of order[] shared by the counting and the scatter pass --------------------------------------
// Per splat the 64 lanes take the quadrants of its snug rect (row-major; rects beyond 64 quadrants take more rounds) and every
// lane runs the exact reach test for its own quadrant.  Splat records are fetched 64 at a time (lane l: splat j + l, one memory
// round trip per 64 splats) and broadcast with v_readlane, so no load sits inside the serial loop.
//
//   counting pass  : workgroup = chunk, its 4 waves take a quarter of the chunk each (counting needs no order) and share ONE
//                    LDS row of byte counters (S <= 255), four to a word; the hit mask of every round is kept.
//   scatter pass   : ONE wave per chunk walks its splats strictly one after the other -- quadrants of one splat are distinct, so
//                    the per-quadrant LDS words need no conflict handling inside a splat and the append is stable.  The LDS row
//                    holds ABSOLUTE stream cursors for the whole grid (4 bytes per quadrant), preloaded with coalesced reads of
//                    qstart + qprefix[chunk]: no dependent global load in the serial loop, and the saved masks replace the tests.
__device__ __forceinline__ int quadrant_of(int qx, int qy, int gx) { return 4 * ((qy >> 1) * gx + (qx >> 1)) + (((qy & 1) << 1) | (qx & 1)); }

__global__ __launch_bounds__(256) void k_qcount(QBinArgs a)
{
    extern __shared__ uint32_t cnt[];                              // ceil(Q/4) words of 4 byte counters, shared by the 4 waves
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t c = blockIdx.x;
    const uint32_t nvis = a.hdr->nvis;
    const uint32_t S = (nvis + a.chunks - 1) / a.chunks;           // <= 255 by construction (gsr_api.hip)
    const uint32_t qw = ((uint32_t)a.Q + 3) >> 2;
    for (uint32_t w = threadIdx.x; w < qw; w += 256) cnt[w] = 0u;
    __syncthreads();
    const uint32_t j0 = min(c * S, nvis), j1 = min(j0 + S, nvis);
    // wave w takes the splats j0 + 64 * (4 r + w) ... of every batch round r: batches of 64 splats, dealt round-robin
    for (uint32_t jb = j0 + 64u * (uint32_t)wave; jb < j1; jb += 256u) {
        const uint32_t mine_j = jb + (uint32_t)lane;
        const uint32_t my_sid = mine_j < j1 ? a.order[mine_j] : 0u;
        float4 m0 = make_float4(0.f, 0.f, 0.f, 0.f), m1 = m0, m2 = m0;
        if (mine_j < j1) { m0 = a.brec[3 * (size_t)my_sid]; m1 = a.brec[3 * (size_t)my_sid + 1]; m2 = a.brec[3 * (size_t)my_sid + 2]; }
        const int cntb = (int)min(64u, j1 - jb);
        unsigned long long* __restrict__ mask_row = a.qmask + (size_t)jb * GSR_WALK_MASKS;   // GSR_WALK_MASKS words per splat
        for (int l = 0; l < cntb; ++l) {
            auto bc = [&](float v) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); };
            Span r;
            r.px = bc(m0.x); r.py = bc(m0.y); r.B = bc(m0.z); r.det = bc(m0.w);
            r.twoTA = bc(m1.x); r.A = bc(m1.y); r.dyr = bc(m1.z); r.mode = __float_as_int(bc(m1.w));
            const uint32_t r0 = __float_as_uint(bc(m2.x)), r1 = __float_as_uint(bc(m2.y));
            const int qx0 = (int)(r0 & 0xFFFFu), qy0 = (int)(r0 >> 16), qx1 = (int)(r1 & 0xFFFFu), qy1 = (int)(r1 >> 16);
            const int w = qx1 - qx0;
            const int nq = w * (qy1 - qy0);
            const float rw = __builtin_amdgcn_rcpf((float)w);      // k / w = floor((k + 0.5) * (1 / w)): exact for the k < 2^15 of any rect
            int round = 0;
            for (int base = 0; base < nq; base += GSR_WAVE, ++round) {
                const int k = base + lane;
                const int row = (int)(((float)k + 0.5f) * rw);
                const int qx = qx0 + k - row * w, qy = qy0 + row;
                bool hit = k < nq;
                if (hit) hit = band_hit(band_of(r, (float)(qy * 8)), r, (float)(qx * 8));
                const unsigned long long m = __ballot(hit);
                if (round < GSR_WALK_MASKS && lane == 0) mask_row[(size_t)l * GSR_WALK_MASKS + round] = m;
                if (hit) {
                    const uint32_t q = (uint32_t)quadrant_of(qx, qy, a.gx);
                    atomicAdd(&cnt[q >> 2], 1u << ((q & 3u) * 8u));                        // ds_add_u32
                }
            }
        }
    }
    __syncthreads();
    uint32_t* __restrict__ row = a.qhist + (size_t)c * qw;
    for (uint32_t w = threadIdx.x; w < qw; w += 256) row[w] = cnt[w];
}

__global__ __launch_bounds__(64) void k_qscatter(QBinArgs a)
{
    extern __shared__ uint32_t cur[];                              // Q words: the absolute cursor of every quadrant stream for this chunk
    if (a.hdr->total > a.capacity) return;                         // the host grows the buffer and replays the frame
    const int lane = lane_id();
    const uint32_t c = blockIdx.x;
    const uint32_t nvis = a.hdr->nvis;
    const uint32_t S = (nvis + a.chunks - 1) / a.chunks;
    const uint32_t* __restrict__ prefix_row = a.qprefix + (size_t)c * (size_t)a.Q;
    {   // 16 loads in flight per lane: the preload is 2 x 4 Q bytes per wave and must not pay a memory round trip per element
        constexpr int U = 8;
        int k = lane;
        for (; k + (U - 1) * GSR_WAVE < a.Q; k += U * GSR_WAVE) {
            uint32_t x[U], y[U];
#pragma unroll
            for (int u = 0; u < U; ++u) { x[u] = a.qstart[k + u * GSR_WAVE]; y[u] = prefix_row[k + u * GSR_WAVE]; }
#pragma unroll
            for (int u = 0; u < U; ++u) cur[k + u * GSR_WAVE] = x[u] + y[u];
        }
        for (; k < a.Q; k += GSR_WAVE) cur[k] = a.qstart[k] + prefix_row[k];
    }
    const uint32_t j0 = min(c * S, nvis), j1 = min(j0 + S, nvis);
    for (uint32_t jb = j0; jb < j1; jb += GSR_WAVE) {
        const uint32_t mine_j = jb + (uint32_t)lane;
        const uint32_t my_sid = mine_j < j1 ? a.order[mine_j] : 0u;
        float4 m2 = make_float4(0.f, 0.f, 0.f, 0.f);
        unsigned long long mk[GSR_WALK_MASKS];
#pragma unroll
        for (int t = 0; t < GSR_WALK_MASKS; ++t) mk[t] = 0ull;
        if (mine_j < j1) {
            m2 = a.brec[3 * (size_t)my_sid + 2];
#pragma unroll
            for (int t = 0; t < GSR_WALK_MASKS; ++t) mk[t] = a.qmask[(size_t)mine_j * GSR_WALK_MASKS + t];
        }
        const int cntb = (int)min((uint32_t)GSR_WAVE, j1 - jb);
        for (int l = 0; l < cntb; ++l) {
            const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(m2.x), l);
            const uint32_t r1 = (uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(m2.y), l);
            const uint32_t sid = (uint32_t)__builtin_amdgcn_readlane((int)my_sid, l);
            const int qx0 = (int)(r0 & 0xFFFFu), qy0 = (int)(r0 >> 16), qx1 = (int)(r1 & 0xFFFFu), qy1 = (int)(r1 >> 16);
            const int w = qx1 - qx0;
            const int nq = w * (qy1 - qy0);
            const float rw = __builtin_amdgcn_rcpf((float)w);
            Span r;
            r.mode = 3;                                            // operands not fetched yet (needed beyond the saved masks only)
            int round = 0;
            for (int base = 0; base < nq; base += GSR_WAVE, ++round) {
                const int k = base + lane;
                const int row = (int)(((float)k + 0.5f) * rw);
                const int qx = qx0 + k - row * w, qy = qy0 + row;
                unsigned long long m = 0ull;
                bool saved = false;
#pragma unroll
                for (int t = 0; t < GSR_WALK_MASKS; ++t)
                    if (round == t) {
                        m = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(mk[t] >> 32), l) << 32) |
                            (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)mk[t], l);
                        saved = true;
                    }
                if (!saved) {   // a rect of more than 64 * GSR_WALK_MASKS quadrants: test again
                    if (r.mode == 3) {
                        const float4 g0 = a.brec[3 * (size_t)sid], g1 = a.brec[3 * (size_t)sid + 1];
                        r.px = g0.x; r.py = g0.y; r.B = g0.z; r.det = g0.w; r.twoTA = g1.x; r.A = g1.y; r.dyr = g1.z;
                        r.mode = __float_as_int(g1.w);
                    }
                    bool hit = k < nq;
                    if (hit) hit = band_hit(band_of(r, (float)(qy * 8)), r, (float)(qx * 8));
                    m = __ballot(hit);
                }
                if ((m >> lane) & 1ull) {
                    const uint32_t slot = atomicAdd(&cur[quadrant_of(qx, qy, a.gx)], 1u);   // ds_add_rtn_u32: the absolute stream position
                    a.qpos[slot] = sid;
                }
            }
        }
    }
}

// ---- prefix over the chunks of every quadrant ------------------------------------------------------------------------
// Block = 32 packed words (128 quadrants) x 32 segments of the chunk axis: a thread sums its segment of one word column (four
// byte counters per load), the segment bases come from LDS, then the thread walks its segment again writing the exclusive
// prefixes of its four quadrants as one 16-byte store per row.
__global__ __launch_bounds__(1024) void k_qscan(int Q, uint32_t chunks, const uint8_t* __restrict__ qhist8 /* [chunks][qw] words */,
                                                 uint32_t* __restrict__ qprefix /* [chunks][Q] */, uint32_t* __restrict__ qcount /* [Q] */)
{
    __shared__ uint4 part[32][32];
    const uint32_t* __restrict__ qhist = reinterpret_cast<const uint32_t*>(qhist8);
    const int wi = threadIdx.x & 31, seg = threadIdx.x >> 5;
    const uint32_t qw = ((uint32_t)Q + 3u) >> 2;
    const uint32_t word = blockIdx.x * 32u + (uint32_t)wi;
    const uint32_t per = (chunks + 31u) / 32u;
    const uint32_t c0 = min((uint32_t)seg * per, chunks), c1 = min(c0 + per, chunks);
    const bool live = word < qw;
    uint4 sum = make_uint4(0u, 0u, 0u, 0u);
    if (live) {
#pragma unroll 8
        for (uint32_t c = c0; c < c1; ++c) {
            const uint32_t v = qhist[(size_t)c * qw + word];
            sum.x += v & 0xFFu; sum.y += (v >> 8) & 0xFFu; sum.z += (v >> 16) & 0xFFu; sum.w += v >> 24;
        }
    }
    part[seg][wi] = sum;
    __syncthreads();
    uint4 run = make_uint4(0u, 0u, 0u, 0u);
    for (int s2 = 0; s2 < seg; ++s2) { const uint4 p = part[s2][wi]; run.x += p.x; run.y += p.y; run.z += p.z; run.w += p.w; }
    if (!live) return;
    const uint32_t q = word * 4u;
    const bool whole = ((uint32_t)Q & 3u) == 0u;   // rows are then 16-byte aligned: one store per row
    for (uint32_t c = c0; c < c1; ++c) {
        uint32_t* __restrict__ dst = qprefix + (size_t)c * (size_t)Q + q;
        if (whole) {
            *reinterpret_cast<uint4*>(dst) = run;
        } else {
            dst[0] = run.x;
            if (q + 1 < (uint32_t)Q) dst[1] = run.y;
            if (q + 2 < (uint32_t)Q) dst[2] = run.z;
            if (q + 3 < (uint32_t)Q) dst[3] = run.w;
        }
        const uint32_t v = qhist[(size_t)c * qw + word];
        run.x += v & 0xFFu; run.y += (v >> 8) & 0xFFu; run.z += (v >> 16) & 0xFFu; run.w += v >> 24;
    }
    if (seg == 31) {
        qcount[q] = run.x;
        if (q + 1 < (uint32_t)Q) qcount[q + 1] = run.y;
        if (q + 2 < (uint32_t)Q) qcount[q + 2] = run.z;
        if (q + 3 < (uint32_t)Q) qcount[q + 3] = run.w;
    }
}

// one workgroup: exclusive scan of the quadrant totals -> stream starts; the count goes to the device header and the host
// mailbox; tiles listed heaviest-first (by the sum of their four quadrants) as the launch order of the blend kernels
__global__ __launch_bounds__(1024) void k_qscan_glob(int tiles, const uint32_t* __restrict__ qcount, uint32_t* __restrict__ qstart,
                                                      uint32_t* __restrict__ tile_order, BinHeader* __restrict__ hdr,
                                                      unsigned long long* mailbox, unsigned long long seq, unsigned long long post_capacity)
{
    __shared__ uint32_t wave_tot[16];
    __shared__ unsigned long long carry_s;
    __shared__ uint32_t bucket[34];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int Q = 4 * tiles;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < Q; base += 1024) {
        const int t = base + tid;
        const uint32_t v = t < Q ? qcount[t] : 0u;
        uint32_t incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = __shfl_up(incl, d, 64);
            if (lane >= d) incl += up;
        }
        if (lane == 63) wave_tot[wid] = incl;
        __syncthreads();
        uint32_t wave_off = 0;
        for (int w = 0; w < wid; ++w) wave_off += wave_tot[w];
        const unsigned long long carry = carry_s;
        if (t < Q) qstart[t] = (uint32_t)carry + wave_off + incl - v;
        __syncthreads();
        if (tid == 1023) carry_s = carry + wave_off + incl;
        __syncthreads();
    }
    if (tid == 0) {
        const unsigned long long grand = carry_s;
        hdr->total = grand;
        __hip_atomic_store(mailbox, (seq << 40) | (grand & 0xFFFFFFFFFFull), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        // deferred count (post_capacity = the binning capacity, else ~0): a frame that does not fit leaves a STICKY mark in the slot's second
        // word -- later frames overwrite the count above, nothing but the host clears this one (gsr_count_slot_overflow)
        if (grand > post_capacity) __hip_atomic_store(mailbox + GSR_COUNT_SLOTS, grand, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (tid < 34) bucket[tid] = 0u;
    __syncthreads();
    auto tile_sum = [&](int t) { return qcount[4 * t] + qcount[4 * t + 1] + qcount[4 * t + 2] + qcount[4 * t + 3]; };
    auto bucket_of = [](uint32_t c) { return c ? 32u - (uint32_t)(31 - __builtin_clz(c)) - 1u : 32u; };
    for (int t = tid; t < tiles; t += 1024) atomicAdd(&bucket[bucket_of(tile_sum(t))], 1u);
    __syncthreads();
    if (tid == 0) {
        uint32_t run = 0;
        for (int b = 0; b < 33; ++b) { const uint32_t c = bucket[b]; bucket[b] = run; run += c; }
    }
    __syncthreads();
    for (int t = tid; t < tiles; t += 1024) tile_order[atomicAdd(&bucket[bucket_of(tile_sum(t))], 1u)] = (uint32_t)t;
}

}  // namespace gsr
