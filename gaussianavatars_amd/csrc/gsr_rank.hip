// gsr_rank.hip -- per-tile ordering by global depth RANK (gfx950, wave64): the per-tile sort path without a per-tile sort.
//
// The reference orders every (splat, tile) instance by (tile, depth) with one radix sort over all instances (A.2); round 1
// sorted each tile's 64-bit (depth, splat) keys in LDS with a bitonic network.  The order inside a tile is only the SPLATS'
// (depth, index) order restricted to the tile, so here the splats are ranked once (P keys, 17x fewer than instances) and a
// tile orders its instances by setting bit `rank` in an LDS bitmap and reading the bitmap back in order -- no comparisons:
//
//   k_rcount    per chunk  : tile-instance histogram (as round 1's k_count) + depth-bucket histogram of the binned splats,
//                            snug tile rect of every splat kept for the two scatters
//   k_rdscatter per chunk  : bucket offsets (every workgroup scans the nb counters itself), (depth bits << 32 | splat) into buckets
//               + 1 WG     : exclusive scan of the tile counters, heavy-first tile order, instance count posted to the host
//   k_rsort_rscatter (one band; round 4), ONE launch with two kinds of workgroup that do not depend on each other:
//               per chunk  : (splat | quadrant mask << 28) into every tile segment the splat's snug rect covers (4 bytes per instance)
//               per bucket : register bitonic sort (gsr_sort.h) -> rank[splat]
//   (frames beyond GSR_RANK_MAX_SPLATS splats: k_rdsort per bucket, k_band_* = the rank of a splat inside every band of tile rows it
//    touches, then k_rscatter with (rank in the tile's band, splat | mask) entries of 8 bytes)
//   k_tile_rank per tile   : bitmap of the tile's ranks in LDS (ds_or; one band: rank[splat] gathered per entry), word popcounts scanned; an
//                            entry's position in the sorted list is the number of set bits below its own; then the four 8x8-quadrant
//                            streams exactly as round 1's sort epilogue wrote them (stable compaction of the entries whose ellipse reaches
//                            the quadrant)
//
// Lists come out identical to a (depth, index) sort per tile: ranks are unique, and rank order IS (depth, index) order.
#include "gsr_device.h"
#include "gsr_sort.h"

namespace gsr {

__device__ __forceinline__ uint32_t rank_bucket(float depth, float lo, float scale, uint32_t nb)
{
    const int b = f2i_sat((depth - lo) * scale);
    return (uint32_t)min(max(b, 0), (int)nb - 1);
}
__device__ __forceinline__ void rank_bucket_map(uint32_t dmin_bits, uint32_t dmax_bits, uint32_t nb, float& lo, float& scale)
{
    lo = __uint_as_float(dmin_bits);
    const float span = __uint_as_float(dmax_bits) - lo;
    scale = span > 0.f ? (float)nb * 0.99999f / span : 0.f;   // every depth of the frame lands in [0, nb); monotone in depth
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) { return ~wave_max_u32(~v); }

// Expansion of a splat's tile rect shared by the counting and the scatter pass: GSR_RANK_GROUP lanes take one splat together
// (lanes striding its tiles), i.e. 4 splats per wave at a time -- a splat per lane leaves 100 k splats at one wave per SIMD and the
// pass latency-bound.  Rects beyond BIG tiles are expanded by the whole wave, one splat at a time, so that a screen-filling splat
// does not serialise one group for thousands of iterations.  Must be called by all 64 lanes; every lane of a group passes the same
// rect (n == 0 for idle groups).  f(x, y, src_lane): src_lane = the lane whose splat this tile belongs to (lane of its group).
template <int G = GSR_RANK_GROUP, typename F>
__device__ __forceinline__ void for_each_tile_grouped(int minx, int miny, int maxx, int maxy, uint32_t n, F f)
{
    constexpr uint32_t BIG = 256;
    const int lane = lane_id(), sub = lane & (G - 1);
    if (n > 0 && n <= BIG) {
        const uint32_t w = (uint32_t)(maxx - minx);
        const float rw = __builtin_amdgcn_rcpf((float)w);      // k / w = floor((k + 0.5) * (1 / w)): exact for k < 2^15
        for (uint32_t k = (uint32_t)sub; k < n; k += G) {
            const uint32_t row = (uint32_t)(((float)k + 0.5f) * rw);
            f((uint32_t)minx + k - row * w, (uint32_t)miny + row, lane);
        }
    }
    uint64_t big = __ballot(n > BIG && sub == 0);
    while (big) {
        const int src = __builtin_ctzll(big);
        big &= big - 1;
        const uint32_t bminx = (uint32_t)__builtin_amdgcn_readlane(minx, src), bminy = (uint32_t)__builtin_amdgcn_readlane(miny, src);
        const uint32_t w = (uint32_t)__builtin_amdgcn_readlane(maxx, src) - bminx, bn = (uint32_t)__builtin_amdgcn_readlane((int)n, src);
        for (uint32_t k = (uint32_t)lane; k < bn; k += GSR_WAVE) f(bminx + k % w, bminy + k / w, src);
    }
}

// ------------------------------------------------------------------------------------------
// k_rcount: <= GSR_BIN_BLOCKS workgroups, each a contiguous chunk of splats.  Tile histogram and depth-bucket histogram in
// LDS (ds_add, no return); only non-empty bins go to the global counters; both per-workgroup histograms are kept for the
// scatters (same chunking).  The rect of every splat comes from k_preprocess (srect: snug in the culling modes).
// The frame's depth range comes from k_preprocess's per-workgroup (min, max) pairs: no atomics, nothing to zero.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(GSR_RANK_BIN_THREADS) void k_rcount(int ilv, int P, int gx, int tiles, int pblocks, uint32_t nb, const ushort4* __restrict__ srect,
                                                                  const uint32_t* __restrict__ tiles_touched, const float* __restrict__ depths,
                                                                  const uint4* __restrict__ pstat, uint32_t* __restrict__ tile_count,
                                                                  unsigned long long* __restrict__ rect_total, uint32_t* __restrict__ block_hist,
                                                                  uint32_t* __restrict__ bcount, uint32_t* __restrict__ bhist,
                                                                  BinHeader* __restrict__ hdr, uint32_t* __restrict__ cb)
{
    extern __shared__ uint32_t lds[];
    uint32_t* const dh = lds;            // [nb] depth buckets
    uint32_t* const hist = lds + nb;     // [tiles] (absent beyond GSR_RANK_HIST_TILES)
    constexpr int NT = GSR_RANK_BIN_THREADS, NWV = NT / 64, G = GSR_RANK_GROUP;
    __shared__ unsigned long long rect_sum[NWV];
    __shared__ uint32_t s_mn[NWV], s_mx[NWV];
    const int tid = threadIdx.x;
    const bool direct = rank_direct(gx, tiles);
    uint32_t mn = 0xFFFFFFFFu, mx = 0u, imax = 0u, isum = 0u;   // (isum: 2^32 instances is also the limit of the binning state's offsets)
    for (int j = tid; j < pblocks; j += NT) {
        const uint4 v = pstat[j];
        mn = min(mn, v.x);
        mx = max(mx, v.y);
        if (blockIdx.x == 0) { imax = max(imax, v.z); isum += v.z; }
    }
    mn = wave_min_u32(mn);
    mx = wave_max_u32(mx);
    __shared__ uint32_t s_imax[NWV], s_isum[NWV];
    if (blockIdx.x == 0) {   // how unevenly the tile instances are spread along the splat order (for the NEXT frame's chunking: gsr_api.hip)
        imax = wave_max_u32(imax);
        isum = (uint32_t)__builtin_amdgcn_readlane((int)wave_scan_incl_u32(isum), 63);
        if ((tid & 63) == 0) { s_imax[tid >> 6] = imax; s_isum[tid >> 6] = isum; }
    }
    if ((tid & 63) == 0) { s_mn[tid >> 6] = mn; s_mx[tid >> 6] = mx; }
    for (uint32_t t = tid; t < nb; t += NT) dh[t] = 0u;
    if (!direct)
        for (int t = tid; t < (gx + 1) * (tiles / gx + 1); t += NT) hist[t] = 0u;
    __syncthreads();
    mn = s_mn[0]; mx = s_mx[0];
#pragma unroll
    for (int w = 1; w < NWV; ++w) { mn = min(mn, s_mn[w]); mx = max(mx, s_mx[w]); }
    if (mn > mx) { mn = 0u; mx = 0u; }   // nothing visible
    if (blockIdx.x == 0 && tid == 0) {
        hdr->dmin_bits = mn; hdr->dmax_bits = mx;
        uint32_t bmax = 0u;
        unsigned long long bsum = 0ull;
#pragma unroll
        for (int w = 0; w < NWV; ++w) { bmax = max(bmax, s_imax[w]); bsum += s_isum[w]; }
        hdr->chunk_imbalance = (pblocks >= 64 && (unsigned long long)bmax * (unsigned long long)pblocks > (unsigned long long)GSR_RANK_IMBALANCE * bsum) ? 1u : 0u;
    }
    float lo, scale;
    rank_bucket_map(mn, mx, nb, lo, scale);

    const int nblk = (int)gridDim.x, blk = (int)blockIdx.x;
    RankMap rmap = RankMap{rank_chunk(P, nblk, ilv), -1};   // local indices 0 .. chunk - 1 of this workgroup's splats: rank_splat_of
    if (ilv == -2) {
        // balanced chunks (gsr_device.h): thread t holds the sixteen 16-splat group sums of k_preprocess's workgroup t (pblocks <= NT on this path);
        // weight = instances + GSR_RANK_SPLAT_WEIGHT per splat; this workgroup's chunk = the groups between the weight targets total * blk / nblk and
        // total * (blk + 1) / nblk -- the number of groups whose inclusive prefix is <= the target, the same function in every workgroup
        __shared__ uint32_t s_wv[NWV], s_c0[NWV], s_c1[NWV];
        uint32_t g[16];
        uint32_t local = 0;
        const int ngroups = (P + 15) >> 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint4 v = tid < pblocks ? pstat[(size_t)pblocks * (1 + q) + tid] : make_uint4(0u, 0u, 0u, 0u);
            g[4 * q] = v.x; g[4 * q + 1] = v.y; g[4 * q + 2] = v.z; g[4 * q + 3] = v.w;
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            g[k] = 16 * tid + k < ngroups ? g[k] + 16u * GSR_RANK_SPLAT_WEIGHT : 0u;
            local += g[k];
        }
        const uint32_t incl = wave_scan_incl_u32(local);
        if ((tid & 63) == 63) s_wv[tid >> 6] = incl;
        __syncthreads();
        uint32_t before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < NWV; ++w) { const uint32_t v = s_wv[w]; if (w < (tid >> 6)) before += v; total += v; }
        const unsigned long long t0 = (unsigned long long)total * (unsigned long long)blk / (unsigned long long)nblk;
        const unsigned long long t1 = (unsigned long long)total * (unsigned long long)(blk + 1) / (unsigned long long)nblk;
        uint32_t run = before + incl - local, c0 = 0, c1 = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            run += g[k];
            const bool exists = 16 * tid + k < ngroups;
            c0 += (exists && (unsigned long long)run <= t0) ? 1u : 0u;
            c1 += (exists && (unsigned long long)run <= t1) ? 1u : 0u;
        }
        c0 = (uint32_t)__builtin_amdgcn_readlane((int)wave_scan_incl_u32(c0), 63);
        c1 = (uint32_t)__builtin_amdgcn_readlane((int)wave_scan_incl_u32(c1), 63);
        if ((tid & 63) == 0) { s_c0[tid >> 6] = c0; s_c1[tid >> 6] = c1; }
        __syncthreads();
        uint32_t g0 = 0, g1 = 0;
#pragma unroll
        for (int w = 0; w < NWV; ++w) { g0 += s_c0[w]; g1 += s_c1[w]; }
        if (blk == 0) g0 = 0u;
        if (blk == nblk - 1) g1 = (uint32_t)ngroups;
        const int s0 = min(P, (int)(g0 << 4)), s1 = min(P, (int)(g1 << 4));
        rmap = RankMap{s1 - s0, s0};
        if (tid == 0) { cb[blk] = (uint32_t)s0; if (blk == nblk - 1) cb[nblk] = (uint32_t)s1; }
    }
    const int chunk = rmap.chunk;
    unsigned long long touched = 0;
    const int gy = tiles / gx, sx = gx + 1;   // the LDS grid has one more column and row: rect corners lie on tile CORNERS
    if (direct) {
        for (int base = 0; base < chunk; base += NT / G) {
            const int j = base + tid / G, i = j < chunk ? rank_splat_of(j, blk, nblk, rmap, ilv) : P;
            uint32_t n = 0;
            int minx = 0, miny = 0, maxx = 0, maxy = 0;
            if (i < P) {
                const ushort4 r = srect[i];
                minx = r.x; miny = r.y; maxx = r.z; maxy = r.w;
                n = (uint32_t)((maxx - minx) * (maxy - miny));
                if ((tid & (G - 1)) == 0) {
                    touched += tiles_touched[i];
                    if (n) atomicAdd(&dh[rank_bucket(depths[i], lo, scale, nb)], 1u);
                }
            }
            for_each_tile_grouped(minx, miny, maxx, maxy, n, [=](uint32_t x, uint32_t y, int) { atomicAdd(&tile_count[y * (uint32_t)gx + x], 1u); });
        }
    } else {
        // A rect adds 1 to every tile it covers = +1 / -1 / -1 / +1 at its four corners followed by a 2-D prefix sum over the grid:
        // four LDS atomics per splat whatever its size (18 tiles on average), one splat per lane
        for (int j = tid; j < chunk; j += NT) {
            const int i = rank_splat_of(j, blk, nblk, rmap, ilv);
            if (i >= P) continue;
            const ushort4 r = srect[i];
            touched += tiles_touched[i];
            if (r.z != r.x) {
                atomicAdd(&dh[rank_bucket(depths[i], lo, scale, nb)], 1u);
                atomicAdd(&hist[r.y * sx + r.x], 1u);
                atomicSub(&hist[r.y * sx + r.z], 1u);
                atomicSub(&hist[r.w * sx + r.x], 1u);
                atomicAdd(&hist[r.w * sx + r.z], 1u);
            }
        }
        __syncthreads();
        const int lane = tid & 63, wv = tid >> 6;
        for (int y = wv; y <= gy; y += NWV) {            // along x: a wave per row, 64 columns at a time
            uint32_t carry = 0;
            for (int x0 = 0; x0 < sx; x0 += 64) {
                const int x = x0 + lane;
                const uint32_t v = x < sx ? hist[y * sx + x] : 0u;
                const uint32_t incl = wave_scan_incl_u32(v) + carry;
                if (x < sx) hist[y * sx + x] = incl;
                carry = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            }
        }
        __syncthreads();
        for (int x = wv; x < gx; x += NWV) {             // along y: a wave per column, 64 rows at a time
            uint32_t carry = 0;
            for (int y0 = 0; y0 <= gy; y0 += 64) {
                const int y = y0 + lane;
                const uint32_t v = y <= gy ? hist[y * sx + x] : 0u;
                const uint32_t incl = wave_scan_incl_u32(v) + carry;
                if (y <= gy) hist[y * sx + x] = incl;
                carry = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            }
        }
        __syncthreads();   // hist[y * sx + x] = this workgroup's instances in tile (x, y)
    }
    // the rect-based instance count (the reference's num_rendered) is kept beside the culled one: sum of tiles_touched
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) touched += __shfl_xor(touched, d, 64);
    if ((tid & 63) == 0) rect_sum[tid >> 6] = touched;
    __syncthreads();
    if (tid == 0) {
        unsigned long long tot = 0;
#pragma unroll
        for (int w = 0; w < NWV; ++w) tot += rect_sum[w];
        atomicAdd(rect_total, tot);
    }
    // One RETURNING L2 atomic per (workgroup, non-empty bin) both counts the bin and reserves this workgroup's sub-range inside it
    // (in arrival order): the offset is what the rows keep for the two scatters, which then need no atomics of their own.
    uint32_t* __restrict__ bmine = bhist + (size_t)blockIdx.x * nb;
    for (uint32_t t = tid; t < nb; t += NT) {
        const uint32_t v = dh[t];
        bmine[t] = v ? atomicAdd(&bcount[t], v) : 0u;
    }
    if (direct) return;
    uint32_t* __restrict__ mine = block_hist + (size_t)blockIdx.x * tiles;
    for (int t = tid; t < tiles; t += NT) {
        const int y = t / gx;
        const uint32_t v = hist[t + y];   // y * sx + x
        mine[t] = v ? atomicAdd(&tile_count[t], v) : 0u;
    }
}

// the tile counters -> tile offsets, heavy-first tile order, the instance count posted to the host: 256 threads of ONE workgroup
__device__ __forceinline__ void tile_scan_256(int tiles, const uint32_t* __restrict__ tile_count, uint32_t* __restrict__ tile_start,
                                              uint32_t* __restrict__ tile_cursor, uint2* __restrict__ ranges,
                                              uint32_t* __restrict__ tile_order, uint4* __restrict__ tdesc,
                                              unsigned long long* __restrict__ total_dev, unsigned long long* mailbox, unsigned long long seq, unsigned long long post_capacity,
                                              const BinHeader* __restrict__ hdr = nullptr)
{
    __shared__ uint32_t wave_tot[4];
    __shared__ unsigned long long carry_s;
    __shared__ uint32_t bucket[34];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < tiles; base += 1024) {
        const int t0 = base + 4 * tid;   // four consecutive tiles per thread
        uint32_t v[4], sum = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            v[k] = (t0 + k < tiles) ? tile_count[t0 + k] : 0u;
            sum += v[k];
        }
        uint32_t incl = sum;
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
        uint32_t excl = (uint32_t)carry + wave_off + incl - sum;   // offsets are 32-bit like upstream's
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (t0 + k < tiles) {
                tile_start[t0 + k] = excl;
                tile_cursor[t0 + k] = 0u;
                if (ranges) ranges[t0 + k] = v[k] ? make_uint2(excl, excl + v[k]) : make_uint2(0u, 0u);
            }
            excl += v[k];
        }
        __syncthreads();
        if (tid == 255) carry_s = carry + wave_off + incl;   // 64-bit running total: a frame past 2^32 instances is reported, not wrapped
        __syncthreads();
    }
    if (tid == 0) {
        const unsigned long long grand = carry_s;
        *total_dev = grand;
        // post (seq, I) to the host: one 8-byte system-scope store into mapped pinned memory
        // (bit 39 of the count: this frame's splat order is uneven in tile instances -- BinHeader::chunk_imbalance, written by k_rcount in the launch before)
        __hip_atomic_store(mailbox, (seq << 40) | (grand & 0x7FFFFFFFFFull) | ((unsigned long long)(hdr ? hdr->chunk_imbalance & 1u : 0u) << 39), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        // deferred count (post_capacity = the binning capacity, else ~0): a frame that does not fit leaves a STICKY mark in the slot's second
        // word -- later frames overwrite the count above, nothing but the host clears this one (gsr_count_slot_overflow)
        if (grand > post_capacity) __hip_atomic_store(mailbox + GSR_COUNT_SLOTS, grand, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // heavy-first launch order for the per-tile kernels (pure scheduling, see round 1's k_tile_scan)
    if (tid < 34) bucket[tid] = 0u;
    __syncthreads();
    auto bucket_of = [](uint32_t c) { return c ? 32u - (uint32_t)(31 - __builtin_clz(c)) - 1u : 32u; };   // big counts first, empty last
    for (int t = tid; t < tiles; t += 256) atomicAdd(&bucket[bucket_of(tile_count[t])], 1u);
    __syncthreads();
    if (tid == 0) {
        uint32_t run = 0;
        for (int b = 0; b < 33; ++b) { const uint32_t c = bucket[b]; bucket[b] = run; run += c; }
    }
    __syncthreads();
    // (tile_start was written by other threads of this workgroup above: visible after the barriers in between)
    for (int t = tid; t < tiles; t += 256) {
        const uint32_t c = tile_count[t];
        const uint32_t pos = atomicAdd(&bucket[bucket_of(c)], 1u);
        tile_order[pos] = (uint32_t)t;
        tdesc[pos] = make_uint4((uint32_t)t, c, tile_start[t], 0u);   // what the per-tile kernel needs, in one load
    }
}

// The same by all GSR_RANK_BIN_THREADS threads of a workgroup, for grids of up to TS_PER tiles per thread: every counter is loaded ONCE and stays in
// registers through the scan, the heavy-first bucket count and the placement -- one level of global loads instead of four dependent ones (the
// 256-thread version above runs 8 - 13 us on its own; beside the depth sort that was hidden, in front of k_rsort_rscatter it would not be).
constexpr int TS_PER = 8;
#ifndef GSR_TS_SUB
#define GSR_TS_SUB 0   // heavy-first order: 2^GSR_TS_SUB classes per octave of the tile's entry count
#endif
constexpr int TS_BUCKETS = 32 * (1 << GSR_TS_SUB) + 1;
__device__ __forceinline__ void tile_scan_wide(const TileScanArgs& ts)
{
    constexpr int NT = GSR_RANK_BIN_THREADS, NWV = NT / 64;
    __shared__ uint32_t wave_tot[NWV];
    __shared__ uint32_t bucket[TS_BUCKETS + 1];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    // wave w owns the tiles [64 per w, 64 per (w + 1)), lane l the tiles 64 per w + 64 k + l: coalesced loads, and the placement below hands out
    // positions in tile order inside a wave -- neighbouring tiles stay neighbours in the launch order of the per-tile kernels, which is worth
    // 2 us of k_render (their records are shared in L2); `per` <= TS_PER
    const int tiles = ts.tiles, per = (tiles + NT - 1) / NT;
    const int w0 = wid * 64 * per;
    uint32_t v[TS_PER], excl[TS_PER], carry = 0;
#pragma unroll
    for (int k = 0; k < TS_PER; ++k) {
        const int t = w0 + 64 * k + lane;
        v[k] = (k < per && t < tiles) ? ts.tile_count[t] : 0u;
    }
    for (int k = tid; k <= TS_BUCKETS; k += NT) bucket[k] = 0u;
#pragma unroll
    for (int k = 0; k < TS_PER; ++k) {
        const uint32_t incl = wave_scan_incl_u32(v[k]);
        excl[k] = carry + incl - v[k];
        carry += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    }
    if (lane == 63) wave_tot[wid] = carry;
    __syncthreads();
    // big counts first, empty last: the octave of the count and its next GSR_TS_SUB bits (0: whole octaves; finer classes were measured and are
    // not better: 2 us of k_render lost at 8 / 32 classes per octave)
    auto bucket_of = [](uint32_t c) {
        if (!c) return (uint32_t)TS_BUCKETS - 1u;
        const uint32_t o = 31u - (uint32_t)__builtin_clz(c);
        const uint32_t sub = (o >= GSR_TS_SUB ? c >> (o - GSR_TS_SUB) : c << (GSR_TS_SUB - o)) & ((1u << GSR_TS_SUB) - 1u);
        return (uint32_t)TS_BUCKETS - 2u - (o << GSR_TS_SUB | sub);
    };
    unsigned long long wave_off = 0, grand = 0;
#pragma unroll
    for (int w = 0; w < NWV; ++w) {
        const uint32_t t = wave_tot[w];
        if (w < wid) wave_off += t;
        grand += t;   // 64-bit total: a frame past 2^32 instances is reported, not wrapped
    }
#pragma unroll
    for (int k = 0; k < TS_PER; ++k) {
        const int t = w0 + 64 * k + lane;
        excl[k] += (uint32_t)wave_off;   // offsets are 32-bit like upstream's
        if (k < per && t < tiles) {
            ts.tile_start[t] = excl[k];
            ts.tile_cursor[t] = 0u;
            if (ts.ranges) ts.ranges[t] = v[k] ? make_uint2(excl[k], excl[k] + v[k]) : make_uint2(0u, 0u);
            atomicAdd(&bucket[bucket_of(v[k])], 1u);
        }
    }
    if (tid == 0) {
        *ts.total_dev = grand;
        // post (seq, I) to the host: one 8-byte system-scope store into mapped pinned memory
        __hip_atomic_store(ts.mailbox, (ts.seq << 40) | (grand & 0x7FFFFFFFFFull) | ((unsigned long long)(ts.hdr ? ts.hdr->chunk_imbalance & 1u : 0u) << 39), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        // deferred count (post_capacity = the binning capacity, else ~0): a frame that does not fit leaves a STICKY mark in the slot's second
        // word -- later frames overwrite the count above, nothing but the host clears this one (gsr_count_slot_overflow)
        if (grand > ts.post_capacity) __hip_atomic_store(ts.mailbox + GSR_COUNT_SLOTS, grand, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __syncthreads();
    if (wid == 0) {   // exclusive scan of the bucket counts (consecutive buckets per lane)
        constexpr int BPL = (TS_BUCKETS + 63) / 64;
        uint32_t c[BPL], tot = 0;
#pragma unroll
        for (int j = 0; j < BPL; ++j) {
            const int bi = lane * BPL + j;
            c[j] = bi < TS_BUCKETS ? bucket[bi] : 0u;
            tot += c[j];
        }
        uint32_t acc = wave_scan_incl_u32(tot) - tot;
#pragma unroll
        for (int j = 0; j < BPL; ++j) {
            const int bi = lane * BPL + j;
            if (bi < TS_BUCKETS) bucket[bi] = acc;
            acc += c[j];
        }
    }
    __syncthreads();
    // heavy-first launch order for the per-tile kernels (pure scheduling, see round 1's k_tile_scan)
#pragma unroll
    for (int k = 0; k < TS_PER; ++k) {
        const int t = w0 + 64 * k + lane;
        if (k < per && t < tiles) {
            const uint32_t pos = atomicAdd(&bucket[bucket_of(v[k])], 1u);
            ts.tile_order[pos] = (uint32_t)t;
            ts.tdesc[pos] = make_uint4((uint32_t)t, v[k], excl[k], 0u);   // what the per-tile kernel needs, in one load
        }
    }
}

// ------------------------------------------------------------------------------------------
// k_rdscatter: same chunking.  Every workgroup scans the nb bucket counters itself (nb <= GSR_RANK_MAX_BUCKETS: a few
// microseconds, no separate one-workgroup launch) and hands out slots from LDS cursors that start at the sub-range k_rcount
// reserved for it in every bucket.  Workgroup 0 publishes the bucket offsets and the number
// of ranked splats.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(GSR_RANK_BIN_THREADS) void k_rdscatter(int ilv, int P, uint32_t nb, const ushort4* __restrict__ srect, const float* __restrict__ depths,
                                                    BinHeader* __restrict__ hdr, const uint32_t* __restrict__ bcount,
                                                    uint32_t* __restrict__ bstart, uint32_t* __restrict__ bcursor,
                                                    unsigned long long* __restrict__ dkeys, const uint32_t* __restrict__ bhist, TileScanArgs ts,
                                                    const uint32_t* __restrict__ cb)
{
    extern __shared__ uint32_t base[];   // [nb]
    constexpr int NT = GSR_RANK_BIN_THREADS, NWV = NT / 64;
    const uint32_t nblk = gridDim.x - 1u;   // the chunk workgroups (k_rcount's chunking); the LAST workgroup scans the tile counters instead:
    if (blockIdx.x == nblk) {               // independent of the depth keys, it rides along here so that k_rscatter can start beside the depth sort
        if (ts.tiles <= TS_PER * NT) { tile_scan_wide(ts); return; }
        if (threadIdx.x >= 256) return;
        tile_scan_256(ts.tiles, ts.tile_count, ts.tile_start, ts.tile_cursor, ts.ranges, ts.tile_order, ts.tdesc, ts.total_dev, ts.mailbox, ts.seq,
                      ts.post_capacity, ts.hdr);
        return;
    }
    __shared__ uint32_t wave_tot[NWV];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const uint32_t per = (nb + (uint32_t)NT - 1u) / (uint32_t)NT;   // consecutive counters per thread
    uint32_t sum = 0;
    for (uint32_t k = 0; k < per; ++k) {
        const uint32_t t = (uint32_t)tid * per + k;
        sum += t < nb ? bcount[t] : 0u;
    }
    const uint32_t incl = wave_scan_incl_u32(sum);
    if (lane == 63) wave_tot[wid] = incl;
    __syncthreads();
    uint32_t run = incl - sum;
    for (int w = 0; w < wid; ++w) run += wave_tot[w];
    const uint32_t* __restrict__ mine = bhist + (size_t)blockIdx.x * nb;
    for (uint32_t k = 0; k < per; ++k) {
        const uint32_t t = (uint32_t)tid * per + k;
        if (t < nb) {
            const uint32_t c = bcount[t];
            if (blockIdx.x == 0) bstart[t] = run;
            base[t] = run + mine[t];   // this workgroup's first slot in bucket t (k_rcount reserved it)
            run += c;
        }
    }
    if (blockIdx.x == 0 && tid == 0) {
        uint32_t tot = 0;
#pragma unroll
        for (int w = 0; w < NWV; ++w) tot += wave_tot[w];
        hdr->nvis = tot;
    }
    __syncthreads();
    float lo, scale;
    rank_bucket_map(hdr->dmin_bits, hdr->dmax_bits, nb, lo, scale);
    const RankMap rmap = rank_map(P, (int)nblk, (int)blockIdx.x, ilv, cb);
    const int chunk = rmap.chunk;
    for (int j = tid; j < chunk; j += NT) {
        const int i = rank_splat_of(j, (int)blockIdx.x, (int)nblk, rmap, ilv);
        if (i >= P) continue;
        const ushort4 r = srect[i];
        if (r.z == r.x) continue;   // not binned
        const float d = depths[i];
        const uint32_t slot = atomicAdd(&base[rank_bucket(d, lo, scale, nb)], 1u);
        dkeys[slot] = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned long long)(uint32_t)i;
    }
}

// ------------------------------------------------------------------------------------------
// The depth sort: a workgroup per bucket, by (depth, splat) -> order[] / rank[].
// ------------------------------------------------------------------------------------------
// one depth bucket, by 256 threads: sorted by (depth, splat); one band: rank[splat] (obs == nullptr); bands: (splat, first band | last band << 8)
// by rank, for k_band_count / k_band_rank
__device__ __forceinline__ void depth_sort_bucket(uint32_t b, const uint32_t* __restrict__ bcount, const uint32_t* __restrict__ bstart,
                                                  unsigned long long* __restrict__ dkeys, unsigned long long* __restrict__ tmp,
                                                  uint32_t* __restrict__ rank, uint2* __restrict__ obs, const ushort4* __restrict__ srect, int band_rows,
                                                  unsigned long long* __restrict__ skeys /* GSR_SORT_SMALL_KEYS keys of LDS */)
{
    constexpr int KEYS = GSR_SORT_SMALL_KEYS, THREADS = 256, EPT = KEYS / THREADS;
    const uint32_t n = bcount[b];
    if (n == 0) return;
    const uint32_t start = bstart[b];
    const int tid = threadIdx.x;
    unsigned long long* seg = dkeys + start;
    auto put = [&](uint32_t splat, uint32_t r) {
        if (obs) {
            const ushort4 q = srect[splat];
            obs[r] = make_uint2(splat, (uint32_t)q.y / (uint32_t)band_rows | ((uint32_t)(q.w - 1) / (uint32_t)band_rows) << 8);
        } else {
            rank[splat] = r;
        }
    };
    if (n <= (uint32_t)KEYS) {
        u64 key[EPT];
        block_sort_regs<THREADS, EPT>(key, skeys, seg, n, tid);
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
            const uint32_t i = (uint32_t)tid * (uint32_t)EPT + (uint32_t)k;
            if (i < n) put((uint32_t)key[k], start + i);
        }
    } else {
        oversize_sort<THREADS, EPT>(seg, tmp + start, skeys, n, tid);
        __syncthreads();
        for (uint32_t i = tid; i < n; i += THREADS) put((uint32_t)seg[i], start + i);
    }
}

// k_rdsort (band mode only: one band sorts beside k_rscatter, see k_rsort_rscatter): workgroup b sorts bucket b
__global__ __launch_bounds__(256) void k_rdsort(const uint32_t* __restrict__ bcount, const uint32_t* __restrict__ bstart,
                                                 unsigned long long* __restrict__ dkeys, unsigned long long* __restrict__ tmp,
                                                 uint32_t* __restrict__ rank, uint2* __restrict__ obs,
                                                 const ushort4* __restrict__ srect, int band_rows)
{
    __shared__ unsigned long long skeys[GSR_SORT_SMALL_KEYS];
    depth_sort_bucket(blockIdx.x, bcount, bstart, dkeys, tmp, rank, obs, srect, band_rows, skeys);
}

// ------------------------------------------------------------------------------------------
// Bands (frames beyond GSR_RANK_MAX_SPLATS splats).  A tile only needs the ORDER of its own instances, and a tile bitmap over
// the whole frame's ranks is almost empty there (2 M ranks, ~2500 instances).  So the tile rows are cut into <= GSR_RANK_BANDS
// bands and a splat gets, for every band its rect touches, its rank among the splats of THAT band: the walk below goes over the
// splats in global depth order (k_rdsort's obs[]), a wave per GSR_RANK_BAND_CHUNK consecutive ranks, one ballot per band.
//   k_band_count: members of every band per chunk;  k_band_scan: exclusive prefix along the chunks, the band totals;
//   k_band_rank : the same walk again, rank of the splat in band b = chunk base + members before the lane: the first four bands
//                 of a rect go to rank4[splat] (what k_rscatter reads, coalesced), further ones (very tall rects) to over[splat][b].
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_band_count(const BinHeader* __restrict__ hdr, const uint2* __restrict__ obs, uint32_t nbands,
                                                     uint32_t nwc, uint32_t* __restrict__ bandcnt)
{
    const uint32_t nvis = hdr->nvis;
    const int lane = lane_id();
    const uint32_t wc = blockIdx.x * 4u + (threadIdx.x >> 6), r0 = wc * (uint32_t)GSR_RANK_BAND_CHUNK;
    if (r0 >= nvis) return;
    uint32_t cnt = 0;   // lane b: members of band b
#pragma unroll
    for (int k = 0; k < GSR_RANK_BAND_CHUNK / 64; ++k) {
        const uint32_t r = r0 + (uint32_t)(k * 64 + lane);
        const bool valid = r < nvis;
        const uint32_t bb = valid ? obs[r].y : 0u, b0 = bb & 255u, b1 = bb >> 8;
        for (uint32_t b = 0; b < nbands; ++b) {
            const uint64_t m = __ballot(valid && b0 <= b && b <= b1);
            if ((uint32_t)lane == b) cnt += (uint32_t)__builtin_popcountll(m);
        }
    }
    if ((uint32_t)lane < nbands) bandcnt[(size_t)lane * nwc + wc] = cnt;
}

__global__ __launch_bounds__(1024) void k_band_scan(BinHeader* __restrict__ hdr, uint32_t nwc, uint32_t* __restrict__ bandcnt)
{
    __shared__ uint32_t wave_tot[16];
    __shared__ uint32_t carry_s;
    const uint32_t n = (hdr->nvis + (uint32_t)GSR_RANK_BAND_CHUNK - 1u) / (uint32_t)GSR_RANK_BAND_CHUNK;   // chunks the count pass filled
    uint32_t* const c = bandcnt + (size_t)blockIdx.x * nwc;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (tid == 0) carry_s = 0u;
    __syncthreads();
    for (uint32_t base = 0; base < n; base += 4096u) {
        const uint32_t i0 = base + 4u * (uint32_t)tid;
        uint32_t v[4], sum = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { v[k] = i0 + k < n ? c[i0 + k] : 0u; sum += v[k]; }
        const uint32_t incl = wave_scan_incl_u32(sum);
        if (lane == 63) wave_tot[wid] = incl;
        __syncthreads();
        uint32_t off = carry_s + incl - sum, all = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) { const uint32_t t = wave_tot[w]; if (w < wid) off += t; all += t; }
#pragma unroll
        for (int k = 0; k < 4; ++k) { if (i0 + k < n) c[i0 + k] = off; off += v[k]; }
        __syncthreads();
        if (tid == 0) carry_s += all;
        __syncthreads();
    }
    if (tid == 0) hdr->band_total[blockIdx.x] = carry_s;
}

__global__ __launch_bounds__(256) void k_band_rank(const BinHeader* __restrict__ hdr, const uint2* __restrict__ obs, uint32_t nbands,
                                                    uint32_t nwc, const uint32_t* __restrict__ bandcnt, uint4* __restrict__ rank4,
                                                    uint32_t* __restrict__ over)
{
    const uint32_t nvis = hdr->nvis;
    const int lane = lane_id();
    const uint32_t wc = blockIdx.x * 4u + (threadIdx.x >> 6), r0 = wc * (uint32_t)GSR_RANK_BAND_CHUNK;
    if (r0 >= nvis) return;
    uint32_t run = (uint32_t)lane < nbands ? bandcnt[(size_t)lane * nwc + wc] : 0u;   // lane b: the next rank of band b
#pragma unroll
    for (int k = 0; k < GSR_RANK_BAND_CHUNK / 64; ++k) {
        const uint32_t r = r0 + (uint32_t)(k * 64 + lane);
        const bool valid = r < nvis;
        const uint2 e = valid ? obs[r] : make_uint2(0u, 0u);
        const uint32_t b0 = e.y & 255u, b1 = e.y >> 8;
        uint4 four = make_uint4(0u, 0u, 0u, 0u);
        for (uint32_t b = 0; b < nbands; ++b) {
            const bool in = valid && b0 <= b && b <= b1;
            const uint64_t m = __ballot(in);
            if (m == 0) continue;
            const uint32_t mine = (uint32_t)__builtin_amdgcn_readlane((int)run, (int)b) + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull));
            if (in) {
                const uint32_t k = b - b0;
                if (k == 0u) four.x = mine;
                else if (k == 1u) four.y = mine;
                else if (k == 2u) four.z = mine;
                else if (k == 3u) four.w = mine;
                else over[(size_t)e.x * nbands + b] = mine;
            }
            if ((uint32_t)lane == b) run += (uint32_t)__builtin_popcountll(m);
        }
        if (valid) rank4[e.x] = four;
    }
}

// ------------------------------------------------------------------------------------------
// k_rscatter: one entry (rank, splat | quadrant mask << 28) into every tile segment of the splat's rect.  Same chunking as
// k_rcount, which reserved a contiguous sub-range per (workgroup, touched tile): slots come from LDS cursors started there.  Order inside a tile segment is arbitrary; the bitmap in k_tile_rank does
// not care.  The mask says which of the tile's four 8x8 quadrants the splat's {alpha >= 1/255} ellipse can reach: evaluated
// here from the splat's Span (k_preprocess), so that the per-tile kernel never gathers a per-splat record.
// Where the time goes (round 3, same-box experiment builds): without its store the kernel runs 186 us of 394 at 2 M splats -- there the 17 M
// scattered 8-byte stores (one 32-byte sector each: 468 MB written for 136 MB) cost as much as everything else -- but at 100 k / 200 k
// splats the store is hidden and the per-instance instruction stream bounds it (quadrant_mask_of without branches: 30.5 -> 26 us).  A
// variant that built the workgroup's entries in LDS grouped by tile and wrote them as runs (count pass, scan, placement, entry-parallel
// copy) was SLOWER at those sizes (26 -> 34 us, 45 -> 58 us: the runs of one (workgroup, tile) pair are only ~4 entries long) and is not
// kept; for the band mode of large frames the runs would have to come from splats grouped by band first.
// ------------------------------------------------------------------------------------------
// G lanes expand one splat's rect together: 16 where a rect holds ~18 tiles (100 k - 200 k splats at 802 x 550), 8 on the large frames of the
// band mode, whose rects are smaller (2 M splats at 1600 x 1100: ~12 tiles; k_rscatter 303 -> 265 us; at 100 k splats 8 lanes cost 28.6 -> 31 us)
// LEAN (one band): the entry is the 4-byte (splat | quadrant mask << 28) alone -- nothing here depends on the depth sort, so the workgroups run
// BESIDE it in one launch (k_rsort_rscatter) and k_tile_rank gathers rank[splat] itself.
template <int G, bool LEAN>
__device__ __forceinline__ void rscatter_body(int P, int gx, int tiles, BandTables bt, const ushort4* __restrict__ srect,
                                              const uint32_t* __restrict__ rank, const float4* __restrict__ sspan,
                                              const uint32_t* __restrict__ tile_start, uint32_t* __restrict__ tile_cursor,
                                              uint2* __restrict__ ranks, unsigned long long capacity,
                                              const unsigned long long* __restrict__ total_dev,
                                              const uint32_t* __restrict__ block_hist, int nblk, int ilv, const uint32_t* __restrict__ cb)
{
    extern __shared__ uint32_t hist[];
    uint32_t* __restrict__ const lean = reinterpret_cast<uint32_t*>(ranks);
    constexpr int NT = GSR_RANK_BIN_THREADS;
    if (*total_dev > capacity) return;  // the host will grow the buffer and replay the frame
    const int tid = threadIdx.x;
    const bool direct = rank_direct(gx, tiles);
    const bool bands = !LEAN && bt.nbands > 1;
    const uint4* __restrict__ rank4 = reinterpret_cast<const uint4*>(rank);   // bands: the ranks inside the first four bands of the rect
    const int blk = (int)blockIdx.x;
    const RankMap rmap = rank_map(P, nblk, blk, ilv, cb);
    const int chunk = rmap.chunk;
    auto splat_of = [&](int j) { return j < chunk ? rank_splat_of(j, blk, nblk, rmap, ilv) : P; };   // local index -> splat (P: none)
    // a splat's inputs, fetched one round ahead of their use (all four loads are independent: rank / operands of a splat that is
    // not binned are never looked at)
    struct In { ushort4 q; uint4 rk; float4 s0, s1; };
    auto fetch = [&](int i) {
        In v;
        v.q = make_ushort4(0, 0, 0, 0); v.rk = make_uint4(0u, 0u, 0u, 0u); v.s0 = v.s1 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < P) {
            v.q = srect[i]; v.s0 = sspan[2 * (size_t)i]; v.s1 = sspan[2 * (size_t)i + 1];
            if (bands) v.rk = rank4[i]; else if (!LEAN) v.rk.x = rank[i];
        }
        return v;
    };
    In nxt = fetch(splat_of(tid / G));
    if (!direct) {
        const uint32_t* __restrict__ mine = block_hist + (size_t)blockIdx.x * tiles;
        for (int t = tid; t < tiles; t += NT) hist[t] = tile_start[t] + mine[t];   // first slot of this workgroup in tile t (k_rcount reserved it)
        __syncthreads();
    }
    for (int base = 0; base < chunk; base += NT / G) {
        const int i = splat_of(base + tid / G);
        const In cur = nxt;
        nxt = fetch(splat_of(base + tid / G + NT / G));
        const int minx = cur.q.x, miny = cur.q.y, maxx = cur.q.z, maxy = cur.q.w;
        const uint32_t n = (uint32_t)((maxx - minx) * (maxy - miny));
        const uint4 rk = cur.rk;
        // bands: y / band_rows = floor((y + 0.5) * (1 / band_rows)), exact at these sizes
        const uint32_t fb = bands ? (uint32_t)(((float)miny + 0.5f) * bt.inv_band_rows) : 0u;
        Span sp;
        sp.px = cur.s0.x; sp.py = cur.s0.y; sp.B = cur.s0.z; sp.det = cur.s0.w;
        sp.twoTA = cur.s1.x; sp.A = cur.s1.y; sp.dyr = cur.s1.z; sp.mode = __float_as_int(cur.s1.w);
        for_each_tile_grouped<G>(minx, miny, maxx, maxy, n, [&](uint32_t x, uint32_t y, int src) {
            const int me = lane_id();
            Span b = sp;
            uint4 brk4 = rk;
            uint32_t bfb = fb, bidx = (uint32_t)i;
            if (src != me) {   // whole-wave expansion of a large rect: the owner's operands (src is wave-uniform there)
                auto bf = [&](float v) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src)); };
                auto bu = [&](uint32_t v) { return (uint32_t)__builtin_amdgcn_readlane((int)v, src); };
                b.px = bf(sp.px); b.py = bf(sp.py); b.B = bf(sp.B); b.det = bf(sp.det); b.twoTA = bf(sp.twoTA); b.A = bf(sp.A); b.dyr = bf(sp.dyr);
                b.mode = __builtin_amdgcn_readlane(sp.mode, src);
                brk4 = make_uint4(bu(rk.x), bu(rk.y), bu(rk.z), bu(rk.w)); bfb = bu(fb);
                bidx = (uint32_t)__builtin_amdgcn_readlane(i, src);
            }
            const uint32_t tile = y * (uint32_t)gx + x;
            uint32_t brk = brk4.x;
            if (bands) {   // the splat's rank inside the tile's band
                const uint32_t band = (uint32_t)(((float)y + 0.5f) * bt.inv_band_rows), k = band - bfb;
                brk = k == 0u ? brk4.x : k == 1u ? brk4.y : k == 2u ? brk4.z : brk4.w;
                if (k > 3u) brk = bt.over[(size_t)bidx * bt.nbands + band];   // fifth band onwards of a very tall rect
            }
            const uint32_t m = quadrant_mask_of(b, (float)(x * GSR_BLOCK_X), (float)(y * GSR_BLOCK_Y));
            // tile grids beyond the LDS histogram: one returning L2 atomic per instance
            const uint32_t slot = direct ? tile_start[tile] + atomicAdd(&tile_cursor[tile], 1u) : atomicAdd(&hist[tile], 1u);
#ifdef GSR_EXP_RSCATTER_NOSTORE   // timing experiment: everything but the store (the condition keeps the rank, the mask and the slot live; no frame meets it)
            if (brk == 0xFFFFFFF0u && m == 15u && bidx == 0x0FFFFFFFu) ranks[slot] = make_uint2(brk, bidx);
#else
            if (LEAN) lean[slot] = bidx | (m << GSR_RANK_IDX_BITS);
            else ranks[slot] = make_uint2(brk, bidx | (m << GSR_RANK_IDX_BITS));
#endif
        });
    }
}

// ------------------------------------------------------------------------------------------
// The same pass with the tile instances dealt EVENLY to the lanes (round 4).  Above, G lanes expand one splat's rect in lockstep with the
// wave's other groups: a rect of n tiles costs ceil(n / G) trips whether its last trip fills the group or not, and the wave makes as many
// trips as its LARGEST rect needs -- about half of the lane-trips carry an instance (rects of 12 - 23 tiles, G = 8 / 16), and every trip is
// the ~150 instructions of the quadrant mask, the slot and the store.  Here a wave takes 64 splats (a splat per lane: coalesced loads),
// scans their tile counts, parks the operands of the 64 in its own corner of LDS, and walks the instances 64 at a time: instance k belongs
// to the splat whose range [excl, excl + n) holds it -- found without a search: every splat drops its lane number at the window position
// its range starts at, an inclusive max-scan (DPP) spreads it over the positions behind, the previous window's last owner carries over --
// and its operands come back as four ds_read_b128 at the owner's row.  Every lane of every trip but the round's last carries an instance.
// Rects beyond 256 tiles go through the whole-wave expansion as before.  No barriers: the staging rows are the wave's own, and LDS
// instructions of one wave execute in order.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t wave_scan_incl_max_u32(uint32_t v)
{
    auto mx = [](uint32_t a, uint32_t b) { return a > b ? a : b; };
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false));   // row_shr:1 (lanes without a source see 0)
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false));   // row_shr:2
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false));   // row_shr:4
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false));   // row_shr:8
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false));   // row_bcast:15 -> rows 1, 3
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false));   // row_bcast:31 -> rows 2, 3
    return v;
}
// What one lane wrote to the wave's LDS rows, another lane reads: to the compiler that is a data race between threads (it forwarded a lane's
// own store of 0 to its later load of the same word and dropped the load: the owners the other lanes had written never arrived).  A
// wavefront-scope release / acquire pair around a wave barrier makes the hand-over visible to it; the hardware needs nothing (the LDS
// instructions of one wave execute in order), so no instruction is emitted.
__device__ __forceinline__ void wave_lds_handover()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
template <bool LEAN>
struct RscatterStage {
    static constexpr int ROW = LEAN ? 3 : 4;                       // uint4 per splat: span (2), [band ranks], (rect, width, excl, splat)
    uint4 row[GSR_RANK_BIN_THREADS / 64][64][ROW];
    uint32_t own[GSR_RANK_BIN_THREADS / 64][64];
};
template <bool LEAN>
__device__ __forceinline__ void rscatter_balanced(RscatterStage<LEAN>& stage, int P, int gx, int tiles, BandTables bt, const ushort4* __restrict__ srect,
                                                  const uint32_t* __restrict__ rank, const float4* __restrict__ sspan,
                                                  const uint32_t* __restrict__ tile_start, uint32_t* __restrict__ tile_cursor,
                                                  uint2* __restrict__ ranks, unsigned long long capacity,
                                                  const unsigned long long* __restrict__ total_dev,
                                                  const uint32_t* __restrict__ block_hist, int nblk, int ilv, const uint32_t* __restrict__ cb)
{
    extern __shared__ uint32_t hist[];
    uint32_t* __restrict__ const lean = reinterpret_cast<uint32_t*>(ranks);
    constexpr int NT = GSR_RANK_BIN_THREADS, ROW = RscatterStage<LEAN>::ROW;
    constexpr uint32_t BIG = 256;
    if (*total_dev > capacity) return;  // the host will grow the buffer and replay the frame
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool direct = rank_direct(gx, tiles);
    const bool bands = !LEAN && bt.nbands > 1;
    const uint4* __restrict__ rank4 = reinterpret_cast<const uint4*>(rank);
    const int blk = (int)blockIdx.x;
    const RankMap rmap = rank_map(P, nblk, blk, ilv, cb);
    const int chunk = rmap.chunk;
    uint4(*const rows)[ROW] = stage.row[wv];
    uint32_t* const own = stage.own[wv];
    struct In { ushort4 q; uint4 rk; float4 s0, s1; };
    auto fetch = [&](int i) {
        In v;
        v.q = make_ushort4(0, 0, 0, 0); v.rk = make_uint4(0u, 0u, 0u, 0u); v.s0 = v.s1 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < P) {
            v.q = srect[i]; v.s0 = sspan[2 * (size_t)i]; v.s1 = sspan[2 * (size_t)i + 1];
            if (bands) v.rk = rank4[i]; else if (!LEAN) v.rk.x = rank[i];
        }
        return v;
    };
    // a wave takes SB splats per round (lanes SB.. carry none into the scan, but every lane takes instances): 64 on large frames; on small
    // ones the chunk is dealt over all sixteen waves -- at 100 k splats a workgroup's ~400 would otherwise keep seven waves busy and nine idle
    const int SB = min(64, max(8, (chunk + NT / 64 - 1) / (NT / 64)));
    const int stride = SB * (NT / 64);
    auto mine_of = [&](int base) {   // this lane's splat of the round that starts at local index `base` (P: none)
        const int j = base + wv * SB + lane;
        return lane < SB && j < chunk ? rank_splat_of(j, blk, nblk, rmap, ilv) : P;
    };
    In nxt = fetch(mine_of(0));
    if (!direct) {
        const uint32_t* __restrict__ mine = block_hist + (size_t)blockIdx.x * tiles;
        for (int t = tid; t < tiles; t += NT) hist[t] = tile_start[t] + mine[t];   // first slot of this workgroup in tile t (k_rcount reserved it)
        __syncthreads();
    }
    auto f4u = [](float4 v) { return make_uint4(__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)); };
    // one instance: tile (x, y) of the splat whose operands are (sp, rk4, first band fb, index idx)
    auto emit = [&](uint32_t x, uint32_t y, const Span& b, const uint4& brk4, uint32_t bfb, uint32_t bidx) {
        const uint32_t tile = y * (uint32_t)gx + x;
        uint32_t brk = brk4.x;
        if (bands) {   // the splat's rank inside the tile's band
            const uint32_t band = (uint32_t)(((float)y + 0.5f) * bt.inv_band_rows), k = band - bfb;
            brk = k == 0u ? brk4.x : k == 1u ? brk4.y : k == 2u ? brk4.z : brk4.w;
            if (k > 3u) brk = bt.over[(size_t)bidx * bt.nbands + band];   // fifth band onwards of a very tall rect
        }
        const uint32_t m = quadrant_mask_of(b, (float)(x * GSR_BLOCK_X), (float)(y * GSR_BLOCK_Y));
        // tile grids beyond the LDS histogram: one returning L2 atomic per instance
        const uint32_t slot = direct ? tile_start[tile] + atomicAdd(&tile_cursor[tile], 1u) : atomicAdd(&hist[tile], 1u);
        if (LEAN) lean[slot] = bidx | (m << GSR_RANK_IDX_BITS);
        else ranks[slot] = make_uint2(brk, bidx | (m << GSR_RANK_IDX_BITS));
    };
    for (int base = 0; base < chunk; base += stride) {
        const int i = mine_of(base);
        const In cur = nxt;
        nxt = fetch(mine_of(base + stride));
        const int minx = cur.q.x, miny = cur.q.y, maxx = cur.q.z, maxy = cur.q.w;
        const uint32_t n = (uint32_t)((maxx - minx) * (maxy - miny)), w = (uint32_t)(maxx - minx);
        const bool big = n > BIG;
        const uint32_t nbal = big ? 0u : n;
        const uint32_t incl = wave_scan_incl_u32(nbal), excl = incl - nbal;
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        wave_lds_handover();   // (the previous round's reads of these rows are behind us)
        rows[lane][0] = f4u(cur.s0);
        rows[lane][1] = f4u(cur.s1);
        if (!LEAN) rows[lane][2] = cur.rk;
        rows[lane][ROW - 1] = make_uint4((uint32_t)minx | (uint32_t)miny << 16, w, excl, (uint32_t)i);
        wave_lds_handover();
        uint32_t carry = 0u;   // (owner lane + 1) of the position in front of the window
        for (uint32_t k0 = 0; k0 < total; k0 += 64u) {
            own[lane] = 0u;
            const uint32_t rel = excl - k0;                     // (unsigned: ranges that began in an earlier window wrap far above 64)
            if (nbal != 0u && rel < 64u) own[rel] = (uint32_t)lane + 1u;
            wave_lds_handover();
            uint32_t o = wave_scan_incl_max_u32(own[lane]);
            o = o > carry ? o : carry;
            carry = (uint32_t)__builtin_amdgcn_readlane((int)o, 63);
            const uint32_t k = k0 + (uint32_t)lane;
            if (k < total) {
                const uint4(&r)[ROW] = rows[o - 1u];
                const uint4 a0 = r[0], a1 = r[1], a3 = r[ROW - 1];
                uint4 brk4 = make_uint4(0u, 0u, 0u, 0u);
                if (!LEAN) brk4 = r[2];
                Span sp;
                sp.px = __uint_as_float(a0.x); sp.py = __uint_as_float(a0.y); sp.B = __uint_as_float(a0.z); sp.det = __uint_as_float(a0.w);
                sp.twoTA = __uint_as_float(a1.x); sp.A = __uint_as_float(a1.y); sp.dyr = __uint_as_float(a1.z); sp.mode = (int)a1.w;
                const uint32_t ominx = a3.x & 0xFFFFu, ominy = a3.x >> 16, ow = a3.y, t = k - a3.z;
                const uint32_t row = (uint32_t)(((float)t + 0.5f) * __builtin_amdgcn_rcpf((float)ow));   // t / ow: exact for t < 2^15
                const uint32_t fb = bands ? (uint32_t)(((float)ominy + 0.5f) * bt.inv_band_rows) : 0u;
                emit(ominx + t - row * ow, ominy + row, sp, brk4, fb, a3.w);
            }
        }
        // rects beyond BIG tiles: the whole wave expands one splat at a time (a screen-filling splat must not serialise one lane)
        uint64_t bigm = __ballot(big);
        if (bigm) {
            Span sp;
            sp.px = cur.s0.x; sp.py = cur.s0.y; sp.B = cur.s0.z; sp.det = cur.s0.w;
            sp.twoTA = cur.s1.x; sp.A = cur.s1.y; sp.dyr = cur.s1.z; sp.mode = __float_as_int(cur.s1.w);
            const uint32_t fb = bands ? (uint32_t)(((float)miny + 0.5f) * bt.inv_band_rows) : 0u;
            while (bigm) {
                const int src = __builtin_ctzll(bigm);
                bigm &= bigm - 1;
                auto bf = [&](float v) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src)); };
                auto bu = [&](uint32_t v) { return (uint32_t)__builtin_amdgcn_readlane((int)v, src); };
                Span b;
                b.px = bf(sp.px); b.py = bf(sp.py); b.B = bf(sp.B); b.det = bf(sp.det); b.twoTA = bf(sp.twoTA); b.A = bf(sp.A); b.dyr = bf(sp.dyr);
                b.mode = __builtin_amdgcn_readlane(sp.mode, src);
                const uint4 brk4 = make_uint4(bu(cur.rk.x), bu(cur.rk.y), bu(cur.rk.z), bu(cur.rk.w));
                const uint32_t bfb = bu(fb), bidx = bu((uint32_t)i), bminx = bu((uint32_t)minx), bminy = bu((uint32_t)miny), bw = bu(w), bn = bu(n);
                for (uint32_t k = (uint32_t)lane; k < bn; k += GSR_WAVE) emit(bminx + k % bw, bminy + k / bw, b, brk4, bfb, bidx);
            }
        }
    }
}

// stage_off >= 0: the balanced expansion, its staging rows at that byte offset of the dynamic LDS (behind the tile histogram); < 0: the
// lockstep form (the API takes it when the rows do not fit beside the histogram: tile grids of ~23 k tiles and more)
template <int G>
__global__ __launch_bounds__(GSR_RANK_BIN_THREADS) void k_rscatter(int ilv, int P, int gx, int tiles, BandTables bt, const ushort4* __restrict__ srect,
                                                                    const uint32_t* __restrict__ rank, const float4* __restrict__ sspan,
                                                                    const uint32_t* __restrict__ tile_start, uint32_t* __restrict__ tile_cursor,
                                                                    uint2* __restrict__ ranks, unsigned long long capacity,
                                                                    const unsigned long long* __restrict__ total_dev,
                                                                    const uint32_t* __restrict__ block_hist, int stage_off, const uint32_t* __restrict__ cb)
{
    extern __shared__ uint32_t dyn_lds[];
    if (stage_off >= 0)
        rscatter_balanced<false>(*reinterpret_cast<RscatterStage<false>*>(reinterpret_cast<unsigned char*>(dyn_lds) + stage_off), P, gx, tiles, bt, srect, rank, sspan,
                                 tile_start, tile_cursor, ranks, capacity, total_dev, block_hist, (int)gridDim.x, ilv, cb);
    else
        rscatter_body<G, false>(P, gx, tiles, bt, srect, rank, sspan, tile_start, tile_cursor, ranks, capacity, total_dev, block_hist, (int)gridDim.x, ilv, cb);
}

// ------------------------------------------------------------------------------------------
// k_rsort_rscatter (one band): the first `scatter_blocks` workgroups are k_rscatter's (LEAN entries), workgroup scatter_blocks + b sorts
// depth bucket b with its first four waves.  The two halves have no dependence on each other -- both only need the launch before
// (k_rdscatter: bucket contents, tile offsets) -- and the scatter, one 16-wave workgroup per CU at 67 VGPRs, leaves three wave slots per SIMD
// and 130 KB of LDS per CU for the sorting workgroups: the depth sort (10.6 - 15 us as a launch of its own) disappears behind the scatter.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(GSR_RANK_BIN_THREADS) void k_rsort_rscatter(int ilv, int scatter_blocks, int P, int gx, int tiles, const ushort4* __restrict__ srect,
                                                                          const float4* __restrict__ sspan, const uint32_t* __restrict__ tile_start,
                                                                          uint32_t* __restrict__ tile_cursor, uint32_t* __restrict__ entries,
                                                                          unsigned long long capacity, const unsigned long long* __restrict__ total_dev,
                                                                          const uint32_t* __restrict__ block_hist, const uint32_t* __restrict__ bcount,
                                                                          const uint32_t* __restrict__ bstart, unsigned long long* __restrict__ dkeys,
                                                                          unsigned long long* __restrict__ dtmp, uint32_t* __restrict__ rank, int stage_off,
                                                                          const uint32_t* __restrict__ cb)
{
    extern __shared__ uint32_t dyn_lds[];
    if ((int)blockIdx.x < scatter_blocks) {
        BandTables bt;
        bt.nbands = 1u; bt.inv_band_rows = 1.f; bt.over = nullptr;
        if (stage_off >= 0)   // (52 KB of rows + the histogram: two workgroups per CU at 802 x 550 -- a scatter workgroup and a sorting one beside it)
            rscatter_balanced<true>(*reinterpret_cast<RscatterStage<true>*>(reinterpret_cast<unsigned char*>(dyn_lds) + stage_off), P, gx, tiles, bt, srect, nullptr,
                                    sspan, tile_start, tile_cursor, reinterpret_cast<uint2*>(entries), capacity, total_dev, block_hist, scatter_blocks, ilv, cb);
        else
            rscatter_body<GSR_RANK_GROUP, true>(P, gx, tiles, bt, srect, nullptr, sspan, tile_start, tile_cursor, reinterpret_cast<uint2*>(entries), capacity,
                                                total_dev, block_hist, scatter_blocks, ilv, cb);
        return;
    }
    if (threadIdx.x >= 256) return;   // (ended waves do not take part in the barriers of the sort)
    __shared__ unsigned long long skeys[GSR_SORT_SMALL_KEYS];
    depth_sort_bucket(blockIdx.x - (uint32_t)scatter_blocks, bcount, bstart, dkeys, dtmp, rank, nullptr, nullptr, 1, skeys);
}

template __global__ void k_rscatter<8>(int, int, int, int, BandTables, const ushort4*, const uint32_t*, const float4*, const uint32_t*, uint32_t*, uint2*, unsigned long long,
                                       const unsigned long long*, const uint32_t*, int, const uint32_t*);

// ------------------------------------------------------------------------------------------
// The tile's sorted list -> its four quadrant streams, GSR_RANK_WINDOW entries at a time (striped like round 1's epilogue): every
// entry carries its quadrant mask (k_rscatter), so no per-splat record is gathered; stable compaction with ballots + one scan of
// the (chunk, wave) counters; the parity modes also write the reference-format key list.  `lsorted`: the list in LDS (tiles whose
// entries stayed in registers), else `sorted` in global memory (written by this workgroup before the preceding barrier).
// ------------------------------------------------------------------------------------------
template <int THREADS>
__device__ __forceinline__ void tile_streams(uint32_t total, uint32_t n, uint32_t tile, uint32_t start, uint32_t* __restrict__ sorted,
                                             const uint32_t* __restrict__ lsorted, uint32_t (*__restrict__ cntw)[GSR_RANK_WINDOW / 64 + 1],
                                             const float* __restrict__ depths, unsigned long long* __restrict__ keys,
                                             uint32_t* __restrict__ qpbase, uint32_t* __restrict__ qlbase, uint32_t* __restrict__ qcount)
{
    constexpr int NW = THREADS / 64, EPT = GSR_RANK_WINDOW / THREADS, NE = EPT * NW;
    static_assert(NE == GSR_RANK_WINDOW / 64 && NE <= 64, "one (chunk, wave) counter per lane in the scan");
    static_assert(EPT % 4 == 0, "four chunks at a time");
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const bool fast = lsorted != nullptr;
    const unsigned long long tile_hi = (unsigned long long)tile << 32;
    uint32_t run[4] = {0u, 0u, 0u, 0u};
    for (uint32_t win_lo = 0; win_lo < total; win_lo += (uint32_t)GSR_RANK_WINDOW) {
        const uint32_t m = min((uint32_t)GSR_RANK_WINDOW, total - win_lo);
        // ---- 4. epilogue over entries i0 .. i0 + m of the tile's list ----
        const uint32_t i0 = win_lo;
        uint32_t msk[EPT], rnk[EPT], sid[EPT];
#pragma unroll
        for (int h = 0; h < EPT / 4; ++h) {
            if ((uint32_t)(4 * h * THREADS) >= m) {   // workgroup-uniform: nothing left in this window
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    msk[4 * h + k] = 0u; rnk[4 * h + k] = 0u; sid[4 * h + k] = 0u;
                    if (lane < 4) cntw[lane][(4 * h + k) * NW + wid] = 0u;
                }
                continue;
            }
            uint32_t ent[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t j = (uint32_t)(4 * h + k) * THREADS + (uint32_t)tid;
                ent[k] = j < m ? (fast ? lsorted[j] : sorted[i0 + j]) : 0u;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int c = 4 * h + k;
                const uint32_t j = (uint32_t)c * THREADS + (uint32_t)tid;
                const uint32_t idx = ent[k] & ((1u << GSR_RANK_IDX_BITS) - 1u);
                if (j < m && keys) {   // the reference-format lists are a parity/debug artefact: nothing downstream reads them
                    keys[start + i0 + j] = tile_hi | (unsigned long long)__float_as_uint(depths[idx]);
                    sorted[i0 + j] = idx;   // point_list without the mask bits
                }
                const uint32_t mq = j < m ? ent[k] >> GSR_RANK_IDX_BITS : 0u;
                msk[c] = mq;
                sid[c] = idx;
                uint32_t r = 0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const unsigned long long bal = __ballot((mq >> q) & 1u);
                    r |= (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u)) << (8 * q);
                    if (lane == 0) cntw[q][c * NW + wid] = (uint32_t)__builtin_popcountll(bal);
                }
                rnk[c] = r;
            }
        }
        __syncthreads();
        if (wid < 4) {   // wave q scans quadrant q's NE (chunk, wave) counters
            const uint32_t a = lane < NE ? cntw[wid][lane] : 0u;
            const uint32_t incl = wave_scan_incl_u32(a);
            if (lane < NE) cntw[wid][lane] = incl - a;
            if (lane == 63) cntw[wid][NE] = incl;   // total
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < EPT; ++c) {
            const uint32_t mq = msk[c];
            if (mq) {
                const uint32_t j = (uint32_t)c * THREADS + (uint32_t)tid;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if ((mq >> q) & 1u) {
                        const uint32_t pos = run[q] + cntw[q][c * NW + wid] + ((rnk[c] >> (8 * q)) & 0xFFu);
                        qpbase[(size_t)q * n + pos] = sid[c];
                        if (qlbase) qlbase[(size_t)q * n + pos] = i0 + j;
                    }
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) run[q] += cntw[q][NE];
        __syncthreads();   // cntw[] is rewritten by the next window
    }
    if (tid < 4) qcount[4 * tile + tid] = tid == 0 ? run[0] : tid == 1 ? run[1] : tid == 2 ? run[2] : run[3];
}

// ------------------------------------------------------------------------------------------
// k_tile_rank: one workgroup per tile (heaviest first).  The tile's rank space is the frame's binned splats, or -- frames beyond
// GSR_RANK_MAX_SPLATS splats -- those of the tile's band of tile rows; it fits the bitmap (`words` * 32 ranks) in one pass unless
// a single band holds more than GSR_RANK_MAX_SPLATS splats, in which case the four steps below repeat per slice of the space:
//   1. bitmap[r >> 5] |= 1 << (r & 31) for the tile's (rank, splat) entries (ds_or)
//   2. rows of 64 words go round-robin to the waves: a wave scan of the popcounts gives every word its offset inside the row
//      (wprefix, 16 bit), the row totals are scanned by wave 0 -> rowoff[]
//   3. every entry computes its own position in the sorted list -- rowoff[row] + wprefix[word] + popcount(bits below its own) --
//      and drops its splat index there: sorted[position]; GSR_RANK_WINDOW positions at a time
//   4. epilogue, GSR_RANK_WINDOW entries at a time (striped like round 1's): the entry carries the splat's quadrant mask
//      (k_rscatter), so no per-splat record is gathered; stable compaction into the tile's four n-slot streams with
//      ballots + one scan of the (chunk, wave) counters; the parity modes also write the reference-format key list
// No comparison, no data-dependent loop, every step entry-parallel.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(GSR_RANK_TILE_THREADS) void k_tile_rank(uint32_t words, int gx, int nbands, float inv_band_rows,
                                                    const uint4* __restrict__ tdesc, const uint2* __restrict__ ranks, const uint32_t* __restrict__ rank_of,
                                                    const float* __restrict__ depths, const BinHeader* __restrict__ hdr,
                                                    unsigned long long* __restrict__ keys, uint32_t* __restrict__ point_list,
                                                    uint32_t* __restrict__ qlist, uint32_t* __restrict__ qpos, uint32_t* __restrict__ qcount,
                                                    uint32_t* __restrict__ qstart, unsigned long long capacity,
                                                    const unsigned long long* __restrict__ total_dev)
{
    constexpr int THREADS = GSR_RANK_TILE_THREADS, NW = THREADS / 64, EPT = GSR_RANK_WINDOW / THREADS, NE = EPT * NW;
    constexpr int ROWS = GSR_RANK_MAX_SPLATS / 2048, RPL = (ROWS + 63) / 64;   // rows of 64 words = 2048 ranks
#ifndef GSR_RANK_UB
#define GSR_RANK_UB 8
#endif
    constexpr int UB = GSR_RANK_UB;   // entries per thread whose loads are in flight together (and that stay in registers between the two passes)
    static_assert(NE <= 64, "one (chunk, wave) counter per lane in the epilogue's scan");
    static_assert(EPT % 4 == 0, "the epilogue gathers four chunks at a time");
    extern __shared__ uint32_t bitmap[];                                   // [words] (a multiple of 64, at most 64 * ROWS), then
    uint16_t* const wprefix = reinterpret_cast<uint16_t*>(bitmap + words);  // [words] set bits before the word inside its row
    __shared__ uint32_t cntw[4][NE + 1];
    __shared__ uint32_t rowoff[ROWS + 1];
    __shared__ uint32_t lsorted[UB * THREADS];   // the sorted list of a tile whose entries fit the registers (the common case)
    if (*total_dev > capacity) return;
    const uint4 td = tdesc[blockIdx.x];      // (tile, entries, first entry): launch order = heaviest tiles first
    const uint32_t tile = td.x, n = td.y;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (n == 0) {
        if (tid < 4) { qcount[4 * tile + tid] = 0u; qstart[4 * tile + tid] = 0u; }
        return;
    }
    const uint32_t start = td.z;
    if (tid < 4) qstart[4 * tile + tid] = 4u * start + (uint32_t)tid * n;   // the tile's four n-slot streams
    // the tile's entries: (rank, splat | mask) pairs, or -- one band (rank_of) -- the 4-byte (splat | mask) with the rank gathered here
    const uint2* __restrict__ rk2 = ranks + start;
    const uint32_t* __restrict__ rk1 = reinterpret_cast<const uint32_t*>(ranks) + start;
    auto entry = [&](uint32_t i) {
        if (!rank_of) return rk2[i];
        const uint32_t e = rk1[i];
        return make_uint2(rank_of[e & ((1u << GSR_RANK_IDX_BITS) - 1u)], e);
    };
    uint32_t* const qpbase = qpos + (size_t)4 * start;
    uint32_t* const qlbase = qlist ? qlist + (size_t)4 * start : nullptr;
    // the rank space of this tile's entries: the frame's binned splats, or those of the tile's band
    const uint32_t space = nbands > 1 ? hdr->band_total[(uint32_t)(((float)(tile / (uint32_t)gx) + 0.5f) * inv_band_rows)] : hdr->nvis;
    const uint32_t per_pass = words * 32u;
    const bool single = space <= per_pass;   // (always, unless one band holds more than GSR_RANK_MAX_SPLATS splats)
    const bool fast = single && n <= (uint32_t)(UB * THREADS);   // every entry stays in a register between the two passes over them
    uint32_t* const sorted = point_list + start;   // the tile's sorted list in global memory (the reference's point_list)

    uint2 e0[UB];
#pragma unroll
    for (int k = 0; k < UB; ++k) {   // the first UB * THREADS entries: issued before the bitmap is cleared
        const uint32_t i = (uint32_t)(k * THREADS + tid);
        e0[k] = i < n ? entry(i) : make_uint2(0xFFFFFFFFu, 0u);
    }
    // one pass over the ranks [lo, lo + span): the tile's entries in there go to the list from position `placed`; returns their number
    auto pass = [&](const uint32_t lo, const uint32_t span, const bool in_lds, const uint32_t placed) {
        const uint32_t rows = (((span + 31u) >> 5) + 63u) >> 6;   // (0xFFFFFFFF - lo, the filler of missing entries, stays above any span)
        for (uint32_t w = tid; w < rows * 64u; w += THREADS) bitmap[w] = 0u;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < UB; ++k) {
            const uint32_t r = e0[k].x - lo;
            if (r < span) atomicOr(&bitmap[r >> 5], 1u << (r & 31u));
        }
        for (uint32_t base = UB * THREADS; base < n; base += UB * THREADS) {
            uint32_t r[UB];
#pragma unroll
            for (int k = 0; k < UB; ++k) {
                const uint32_t i = base + (uint32_t)(k * THREADS + tid);
                r[k] = i < n ? entry(i).x - lo : 0xFFFFFFFFu;
            }
#pragma unroll
            for (int k = 0; k < UB; ++k)
                if (r[k] < span) atomicOr(&bitmap[r[k] >> 5], 1u << (r[k] & 31u));
        }
        __syncthreads();
        for (uint32_t row = wid; row < rows; row += NW) {
            const uint32_t w = row * 64u + (uint32_t)lane;
            const uint32_t v = (uint32_t)__builtin_popcount(bitmap[w]);
            const uint32_t incl = wave_scan_incl_u32(v);
            wprefix[w] = (uint16_t)(incl - v);
            if (lane == 63) rowoff[row] = incl;   // the row's total, scanned in place below
        }
        __syncthreads();
        if (wid == 0) {   // exclusive scan of the row totals (RPL consecutive rows per lane)
            uint32_t a[RPL], sum = 0;
#pragma unroll
            for (int j = 0; j < RPL; ++j) {
                const uint32_t rr = (uint32_t)(RPL * lane + j);
                a[j] = rr < rows ? rowoff[rr] : 0u;
                sum += a[j];
            }
            const uint32_t incl = wave_scan_incl_u32(sum);
            uint32_t acc = incl - sum;
#pragma unroll
            for (int j = 0; j < RPL; ++j) {
                const uint32_t rr = (uint32_t)(RPL * lane + j);
                if (rr < rows) rowoff[rr] = acc;
                acc += a[j];
            }
            if (lane == 63) rowoff[ROWS] = incl;
        }
        __syncthreads();
        // every entry finds its position and drops its splat index there
        auto position = [&](uint32_t r) {
            const uint32_t w = r >> 5;
            return rowoff[w >> 6] + (uint32_t)wprefix[w] + (uint32_t)__builtin_popcount(bitmap[w] & ((1u << (r & 31u)) - 1u));
        };
        if (in_lds) {   // registers -> LDS: no second read of the entries, no global round trip for the list
#pragma unroll
            for (int k = 0; k < UB; ++k) {
                const uint32_t r = e0[k].x - lo;
                if (r < span) lsorted[position(r)] = e0[k].y;
            }
        } else {
            for (uint32_t base = 0; base < n; base += UB * THREADS) {
                uint2 e[UB];
#pragma unroll
                for (int k = 0; k < UB; ++k) {
                    const uint32_t i = base + (uint32_t)(k * THREADS + tid);
                    e[k] = i < n ? entry(i) : make_uint2(0xFFFFFFFFu, 0u);
                }
#pragma unroll
                for (int k = 0; k < UB; ++k) {
                    const uint32_t r = e[k].x - lo;
                    if (r < span) sorted[placed + position(r)] = e[k].y;
                }
            }
        }
        return rowoff[ROWS];
    };
    uint32_t placed = 0;
    if (single) {
        placed = pass(0u, space, fast, 0u);
    } else {
        for (uint32_t lo = 0; lo < space; lo += per_pass) {
            placed += pass(lo, min(per_pass, space - lo), false, placed);
            __syncthreads();   // the next pass clears the bitmap
        }
    }
    __syncthreads();   // (global stores of this workgroup are visible to it after the barrier)
    // placed == n (ranks are unique inside a rank space)
    tile_streams<THREADS>(placed, n, tile, start, sorted, fast ? lsorted : nullptr, cntw, depths, keys, qpbase, qlbase, qcount);
}

}  // namespace gsr
