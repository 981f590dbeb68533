/*
 * gsr.h -- C ABI of the MI355X-native differentiable Gaussian-splat rasterizer.
 *
 * This is the drop-in boundary for the native half of the reference's
 * `diff_gaussian_rasterization` package (un-vendored submodule; its Python call site is
 * gaussian_renderer/__init__.py:15,37-52,86-94).  Upstream exposes three pybind entry points
 * taking torch tensors (`_C.rasterize_gaussians`, `_C.rasterize_gaussians_backward`,
 * `_C.mark_visible`); here the same three operations are plain C functions over raw device
 * pointers + sizes + a hipStream_t, so that any host language can bind them (ctypes stub in
 * INTEGRATION.md; the shipped binding is gaussianavatars_amd/rasterizer.py).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name ends in `_host`
 *   - all floating point is fp32; tensors are contiguous, row-major, shapes as in the reference
 *   - every entry returns 0 on success, >0 for "call again" conditions (GSR_E_CAPACITY) and <0
 *     on error; gsr_last_error() returns a thread-local message for the last non-zero return
 *   - work is enqueued on `stream` (pass torch's current stream).  gsr_forward performs exactly
 *     one host wait, scoped to `stream`, for the instance count (upstream blocks the whole device
 *     with a cudaMemcpy D2H at the same point).  Nothing else synchronises.
 */
#ifndef GSR_H
#define GSR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GSR_ABI_VERSION 11
#define GSR_BWD_SEGMENT 60        /* stream entries per backward segment (a multiple of both blend kernels' batches)  */
#define GSR_BWD_SEGMENTS 10       /* segments per quadrant stream; the last one takes whatever is left                */
#define GSR_UNIT_LISTS 64         /* fast blend: the backward's work units (quadrant, segment) are appended to this many lists (GsrImageLayout.units) */
#define GSR_BIN_BLOCKS 256        /* workgroups of the two binning passes (each owns a contiguous chunk of splats) */
#ifndef GSR_RANK_BLOCKS
#define GSR_RANK_BLOCKS 256       /* the same for the rank path (csrc/gsr_rank.hip), 1024 threads each: one per CU (swept 64..512)  */
#endif
#define GSR_BLOCK_X 16
#define GSR_BLOCK_Y 16

#define GSR_OK 0
#define GSR_E_CAPACITY 1   /* binning buffer too small; *num_rendered_host holds the needed count */
#define GSR_E_ARG (-1)
#define GSR_E_HIP (-2)
#define GSR_E_TIMEOUT (-3)
#define GSR_COUNT_SLOTS 128  /* persistent instance-count slots of the deferred forwards (GsrSettings.deferred_count) */

/* Mirrors the 12-field GaussianRasterizationSettings NamedTuple the reference constructs at
 * gaussian_renderer/__init__.py:37-50.  bg / viewmatrix / projmatrix / campos stay on the device
 * exactly as the reference passes them (no host copy, hence no sync). */
typedef struct GsrSettings {
    int32_t image_height;
    int32_t image_width;
    float tanfovx;
    float tanfovy;
    const float* bg;          /* (3,)   */
    float scale_modifier;
    const float* viewmatrix;  /* (4,4) = W2C^T, row-major       */
    const float* projmatrix;  /* (4,4) = (P*W2C)^T, row-major   */
    int32_t sh_degree;        /* active degree 0..3             */
    const float* campos;      /* (3,)   */
    int32_t prefiltered;      /* accepted, ignored */
    int32_t debug;            /* !=0: synchronise + check after every kernel */
    int32_t tile_culling;     /* 0: every tile of a splat's rect gets an instance, exactly as upstream (keys, point list,
                                 ranges and n_contrib equal the reference's bit for bit).
                                 !=0: only the tiles the splat's {alpha >= 1/255} ellipse can reach get one.  The dropped
                                 instances can never contribute to a pixel, so image, radii and gradients are the same
                                 bits; the binning state is the culled subsequence of the reference's and num_rendered
                                 counts it (20-26 % fewer instances to sort).  The rect-based count stays available at
                                 byte 8 of the binning buffer.
                                 1: production -- additionally the reference-format sorted `keys` / `point_list` arrays are
                                    not written (nothing downstream reads them; the blend walks the quadrant streams);
                                 2: culled, lists written (what the subsequence parity tests inspect);
                                 3: as 1 but always on the rank path (tiles ordered through bitmaps of global depth ranks: the default
                                    wherever the depth-ordered scatter does not apply), 4: as 1 but on the depth-ordered scatter whenever it
                                    applies, whatever the splat count, 5: as 1 but on round 1's per-tile bitonic sort (A/B of the three
                                    binnings, tests), 6: as 3 with the per-band ranks of large frames at any splat count (tests).
                                    The binning path of a call is GsrBinningLayout.path.                                       */
    int32_t forward_only;     /* !=0: no backward will follow this forward (inference, torch.no_grad): the forward skips zero-filling
                                 the backward's per-splat accumulators (48 B per visible splat) and writing what only the backward reads
                                 (GsrGeomLayout.cov3D, clamped; rect on the rank path; round 4: the image state as well -- final_T,
                                 n_contrib, n_contrib_q, c_final and the blend checkpoints stay unwritten, only out_color leaves the
                                 blend); gsr_backward on such a state is an error                                          */
    int32_t deterministic;    /* !=0: bit-reproducible backward.  The blend backward then adds its per-(wave, splat) partial sums as 64-bit
                                 FIXED-POINT integers (integer addition is associative: the result does not depend on the order the
                                 atomics land in), scaled by a power of two derived from max |dL/dpixel| (one extra reduction kernel);
                                 and from the splat's own footprint; resolution 2^-58 of that bound per term, i.e. finer than the fp32
                                 partials themselves.  Same
                                 value in the forward and the backward call of a frame (the forward zero-fills the accumulators
                                 of the mode).  Default 0: fp32 atomics (summation order varies run to run, like upstream's).    */
    int32_t exact_scale_grad; /* 0 (default): dL/dscales as upstream's computeCov3D backward returns it -- the gradient
                                 w.r.t. (scale_modifier * scale), WITHOUT the modifier's chain-rule factor;
                                 !=0: multiplied by scale_modifier (the mathematically exact gradient).  The two agree at
                                 the reference's scaling_modifier = 1.0 (gaussian_renderer/__init__.py:19). */
    int32_t deferred_count;   /* 0 (default): gsr_forward waits for the frame's instance count, returns it and reports GSR_E_CAPACITY when the
                                 binning buffer is too small (the caller grows it and calls again).
                                 k in 1..GSR_COUNT_SLOTS: nothing waits -- the call only enqueues, *num_rendered_host is -1, and the count is
                                 posted to persistent slot k-1 (gsr_count_slot_read) whenever the kernels get there.  Such a call contains no
                                 host synchronisation, so it can be stream-captured and replayed as a hipGraph.  A frame whose count exceeds
                                 binning_capacity skips its binning, blend and backward blend kernels (image and gradients of that frame are
                                 NOT valid): the caller over-allocates (288 GB of HBM) and checks the slot after the fact.  gsr_backward of
                                 such a state takes num_rendered = binning_capacity.                                              */
    int32_t fast_blend;       /* 0: the two blend kernels evaluate the reference's arithmetic expression by expression (exp through a pure-fmaf
                                 polynomial, no contraction): image, final_T and n_contrib are bit-identical to oracle/gsr_oracle.c -- the mode
                                 every bit-exact parity test runs in.
                                 !=0 (what render() passes): the blend works in the 2^x domain -- k_preprocess stores the conic pre-multiplied by
                                 -log2(e)/2 (the record's A, B, C slots then hold A' = -A log2(e)/2, B' = -B log2(e), C' = -C log2(e)/2) and,
                                 since ABI 10, log2(opacity) in the opacity's slot (alpha = 2^(quadratic + log2 opacity): see GsrGeomLayout.grec), the
                                 exponential is the hardware's v_exp_f32 (~1 ulp), products are fused -- about half the instructions per
                                 record of the exact forward kernel.  Everything integer (radii, rects, tiles_touched, keys, lists, ranges) is unchanged;
                                 the image agrees with the exact mode to ~1e-6 except where a record's alpha sits within an ulp of 1/255 or a
                                 pixel's transmittance within an ulp of 1e-4 (the record is then taken on one side and skipped on the other: a
                                 difference bounded by 1/255 resp. 1e-4 per such pixel; tests/test_fast_blend_gpu.py counts them).  Same value in
                                 the forward and the backward call of a frame.  Ignored (exact kernels) with `deterministic` and on the per-tile
                                 sort path (tile_culling 5), whose binning reads the conic from the record.                         */
} GsrSettings;

/* Byte offsets of the arrays inside the three opaque state buffers.  The state buffers play the
 * role of upstream's geomBuffer / binningBuffer / imgBuffer (saved on the autograd ctx between
 * forward and backward); the layout is published so tests can compare every intermediate with
 * the oracle bit for bit. */
typedef struct GsrGeomLayout {
    size_t depths;         /* float  [P]                                   */
    size_t grec;           /* float4 [3P]  per-splat blend record, 48 B in ONE place: what the quadrant test of the
                              tile sort gathers per instance and what the two blend kernels fetch (scalar loads) per
                              stream entry:  (x, y, A, B | C, opacity, r, g | b, e, 0, 0)
                              x,y = pixel centre, A,B,C = conic, e = (int32 bits) the splat's fixed-point exponent of the
                              deterministic backward.  FAST BLEND (GsrSettings.fast_blend, effective): the record is
                              (x, y, A', B' | C', L, r, g | b, hi, opacity, 0) -- the conic pre-scaled into the 2^x domain, L = log2(opacity)
                              (NaN for an opacity below 1/255: such a splat fails every test), hi = 2^L (1 + 2^-18) the upper bound of the
                              blend's acceptance test 1/255 <= 2^(A'dx^2 + B'dx dy + C'dy^2 + L) <= hi (= the reference's
                              `power <= 0`), and the opacity itself in slot 10 for the backward                          */
    size_t cov3D;          /* float  [6P]  xx xy xz yy yz zz                */
    size_t rect;           /* uint16 [4P]  tile rect min.x min.y max.x max.y */
    size_t tiles_touched;  /* uint32 [P]                                   */
    size_t clamped;        /* uint8  [P]   bit c set <=> channel c clamped  */
    size_t visible;        /* uint8  [P]   1 <=> radii > 0 (render()'s visibility_filter, written by the forward so that no
                              separate comparison kernel is needed)                                    */
    size_t brec;           /* float  [12P] production binning only: the 48-byte binning record of a splat --
                              (px, py, A, B | C, tau, -B/C, -B/A | quadrant rect: x0 | y0 << 16, x1 | y1 << 16, 0, 0) -- i.e. the
                              operands of the exact {alpha >= 1/255} reach test (tau = ln(255 opacity), padded) and the rect of
                              8x8 quadrants to run it on (= the splat's snug tile rect); an empty rect marks a splat that is
                              not binned                                                                    */
    size_t acc64;          /* int64  [10P] deterministic mode's accumulators (same nine sums as `acc`, fixed point, one pad)     */
    size_t acc;            /* float  [12P] backward accumulators of the screen-space gradients (dcolor 3, dmean2D 2,
                              dconic 3, dopacity 1, pad 3).  The forward zeroes the entries of visible splats and the
                              backward zeroes them again after consuming them, so a state is always ready for a
                              backward and no per-frame memset exists.                                 */
    size_t total;
} GsrGeomLayout;

typedef struct GsrBinningLayout {
    /* header (2304 bytes) at byte 0: uint64 instances of this frame (what the capacity must hold: tile instances in the parity modes,
       quadrant-stream entries in production), byte 8: uint64 sum of tiles_touched (the reference's num_rendered), byte 16:
       uint32 binned splats, 20/24: depth range bits, byte 32: uint64 tile instances after snug-rect culling, bytes 40..167:
       uint32 [32] binned splats per band of tile rows (rank path with nbands > 1)                                    */
    size_t keys;        /* uint64 [cap]  parity modes: sorted (tile << 32 | depth bits), upstream's point_list_keys */
    size_t point_list;  /* uint32 [cap]  parity modes: sorted splat index, upstream's point_list                    */
    size_t qlist;       /* uint32 [4*cap] parity modes only (tile_culling 0 / 2): twin of qpos holding each stream entry's
                           POSITION in the tile's sorted list (what the reference's n_contrib counts)                 */
    size_t qpos;        /* uint32: the 8x8-quadrant streams of SPLAT INDICES: quadrant q = 4 * tile + (row & 1) * 2 + (col & 1)
                           lists, in depth order, the splats whose ellipse can reach it, qcount[q] entries from qstart[q].
                           The blend kernels fetch the 48-byte per-splat record (GsrGeomLayout.grec) of each entry.
                           Production: [cap] entries back to back; parity modes: [4*cap], four n-slot streams per tile  */
    size_t qcount;      /* uint32 [4*tiles] entries in each quadrant stream                                       */
    size_t qstart;      /* uint32 [4*tiles] first entry of each quadrant stream inside qpos (and qlist)            */
    size_t ranges;      /* uint32 [2*tiles]  parity modes: [start,end) per tile, (0,0) when empty                 */
    size_t tile_count;  /* uint32 [tiles]    parity modes                                                        */
    size_t tile_start;  /* uint32 [tiles]    parity modes                                                        */
    size_t tile_cursor; /* uint32 [tiles]    parity modes                                                        */
    size_t tile_order;  /* uint32 [tiles]  launch order of the per-tile kernels: heaviest tiles first         */
    size_t block_hist;  /* uint32 [GSR_BIN_BLOCKS * tiles]  parity modes: per-workgroup tile histograms of the counting pass,
                           re-used by the scatter pass; absent (size 0) for tile grids beyond the LDS histogram        */
    /* production binning (depth-ordered scatter into the quadrant streams, csrc/gsr_binning.hip): */
    size_t dkeys;       /* uint64 [P]  (depth bits << 32 | splat) grouped by depth bucket, sorted inside the bucket  */
    size_t dtmp;        /* uint64 [P]  merge scratch of buckets beyond the in-LDS classes                          */
    size_t order;       /* uint32 [P]  the binned splats in (depth, index) order                                    */
    size_t bcount;      /* uint32 [nb] depth buckets: count, start, fill cursor, launch order                       */
    size_t bstart;
    size_t bcursor;
    size_t border;
    size_t bhist;       /* uint32 [GSR_BIN_BLOCKS][nb]  per-workgroup bucket histograms of the bucket count, re-used by the scatter */
    size_t qhist;       /* uint8  [chunks][4*ceil(Q/4)]  entries of chunk c (chunks consecutive runs of `order`) per quadrant */
    size_t qprefix;     /* uint32 [chunks][Q]  exclusive prefix of qhist along the chunk axis                        */
    size_t qmask;       /* uint64 [P][2]  hit masks of the counting pass (by position in `order`, first two rounds of 64 quadrants),
                           replayed by the scatter pass                                                          */
    /* rank path (csrc/gsr_rank.hip): splats ranked by depth once, every tile's instances ordered through an LDS bitmap */
    size_t ranks;       /* uint32 [cap][2] (depth rank, splat) of every (splat, tile) instance, grouped by tile (unordered inside a tile) */
    size_t rank;        /* uint32 [P]   rank of every binned splat in (depth, index) order; with bands (nbands > 1) uint32 [P][4]: its rank
                           among the splats touching the first .. fourth band of tile rows of its rect                            */
    size_t rank_over;   /* uint32 [P][nbands] bands only: the same for the fifth band onwards (only entries of rects that tall are written) */
    size_t srect;       /* uint16 [P][4] tile rect (minx, miny, maxx, maxy) every splat is binned into (snug in the culling modes);
                           zero area = not binned                                                                          */
    size_t sspan;       /* float  [P][8] operands of the per-quadrant reach test of a binned splat (csrc/gsr_device.h: Span)          */
    size_t pstat;       /* uint32 [ceil(P/256)][4] per k_preprocess workgroup: (min, max) depth bits of its visible splats, the tile
                           instances its splats are binned into (how evenly those are spread decides the next frame's chunking), 0;
                           behind the rows (round 6) four planes [ceil(P/256)][4] with the instance sums of the workgroup's sixteen 16-splat
                           groups and k_rcount's chunk boundaries (balanced chunks: csrc/gsr_device.h, rank_map)                          */
    size_t tdesc;       /* uint32 [tiles][4] (tile, entries, first entry, 0) in launch order (heaviest tiles first)                       */
    size_t obs;         /* uint32 [P][2] bands only: (splat, first band | last band << 8) of the binned splats in depth order         */
    size_t bandcnt;     /* uint32 [nbands][ceil(P/256)] bands only: splats of band b before each run of 256 consecutive depth ranks  */
    size_t path;        /* 0: rank path, 1: depth-ordered scatter, 2: round 1's per-tile sort (see gsr_binning_layout)       */
    size_t chunks;      /* production: number of chunks (waves) of the ordered walk                                */
    size_t nb;          /* production: number of depth buckets                                                     */
    size_t nbands;      /* rank path: 1 up to 262144 splats; beyond, the tile rows are cut into this many bands (<= 24) ...  */
    size_t band_rows;   /* ... of this many tile rows, each with a depth rank of its own (header bytes 40..167: uint32 splats per band) */
    size_t total;
} GsrBinningLayout;

typedef struct GsrImageLayout {
    size_t final_T;    /* float  [H*W] */
    size_t n_contrib;  /* uint32 [H*W] last contributor: index+1 in the TILE list (== the reference's n_contrib) in the
                          parity modes (tile_culling 0 / 2); in production (1) the same value as n_contrib_q          */
    size_t n_contrib_q;/* uint32 [H*W] the same position inside the pixel's quadrant stream (what the backward walks) */
    size_t c_final;    /* float  [3*H*W] the composited colour WITHOUT the background term (planar)               */
    size_t ck;         /* float4 [(GSR_BWD_SEGMENTS-1)*H*W] blend checkpoints: slot s-1 of a pixel = (T, C) before entry
                          s*GSR_BWD_SEGMENT of its quadrant stream.  The backward walks each segment of a pixel's stream on
                          its own wave, starting from the checkpoint (the serial walk was the kernel's critical path). */
    size_t gmax;       /* uint32 [1]  deterministic backward: bits of max |dL/dpixel| of the current backward               */
    size_t units;      /* uint32 [32 GSR_UNIT_LISTS + GSR_UNIT_LISTS * cap], cap = ceil(4 tiles / GSR_UNIT_LISTS) * GSR_BWD_SEGMENTS.  Fast blend: the
                          backward's WORK LIST.  A wave of the forward blend knows how deep its quadrant's last contributor sits and appends
                          one unit tile << 6 | quadrant << 4 | segment per 60-entry segment the backward has to walk to list (launch position
                          of the wave) mod GSR_UNIT_LISTS: the counters first (word 32 l = list l's, a 128-byte line each; zeroed by the frame's first kernel), then the lists, `cap` slots
                          each (the mapping wave -> list is static, so `cap` cannot overflow).  The backward's waves (one-wave workgroups, round 6) take the units of list
                          (wave id mod GSR_UNIT_LISTS) in turn, from the list's END -- the deep quadrants are appended last and would be the kernel's tail --: no
                          wave is launched for a (tile, segment) pair nothing reaches (four of five workgroups were, round 3).  Behind the lists (round 6, GSR_CONT_CHUNKS > 0 only):
                          the forward blend's continuation area -- counters, the list of quadrants whose walk was parked at entry
                          GSR_CONT_CHUNKS * GSR_BWD_SEGMENT, 1280 bytes of parked state per quadrant (csrc/gsr_forward.hip: k_render<true, 1 / 2>)  */
    size_t total;
} GsrImageLayout;

int gsr_abi_version(void);
const char* gsr_last_error(void);
/* the newest post to persistent count slot `slot` (0..GSR_COUNT_SLOTS-1): *count = the frame's instances (-1: nothing posted yet),
 * *seq = the posting frame's sequence number.  A plain read of mapped host memory, no synchronisation.                        */
int gsr_count_slot_read(int32_t slot, int64_t* count, int64_t* seq);
/* The slot's STICKY overflow mark: *worst = the instance count of the most recent frame posted to the slot that exceeded the binning
 * capacity of its call (0: every frame since the last reset fitted).  gsr_count_slot_read only shows the newest frame -- a replayed
 * recording overwrites it every launch -- so "did every replay fit" is asked here.  reset != 0 clears the mark (call it with no
 * frame of the slot in flight: before handing the slot to a new recording).  Plain accesses of mapped host memory.             */
int gsr_count_slot_overflow(int32_t slot, int64_t* worst, int32_t reset);
/* (ABI 11) A deferred forward whose caller does want the count before it hands the image on -- but not before it has done the rest of its own
 * host work: gsr_last_forward_seq() = the sequence number of the calling thread's newest gsr_forward* call; gsr_count_slot_wait spins (as
 * gsr_forward does) until persistent slot `slot` holds the post of that frame and returns its instance count.  The caller compares it with
 * the capacity it passed: a frame that did not fit rendered nothing and has to be replayed with a larger buffer, exactly as after GSR_E_CAPACITY. */
int64_t gsr_last_forward_seq(void);
int gsr_count_slot_wait(int32_t slot, int64_t want_seq, void* stream, int64_t* count);

/* sizes/layouts of the state buffers (pure host arithmetic, no device access) */
int gsr_geom_layout(int32_t P, GsrGeomLayout* out);
/* P and tile_culling select the binning path (GsrBinningLayout.path): 0 = the rank path (the default: splats ranked by depth once,
 * every tile ordered through an LDS bitmap of ranks; beyond 262144 splats the ranks are taken per band of tile rows so that a
 * band's ranks still fit the bitmap -- GsrBinningLayout.nbands / band_rows); 1 = the
 * depth-ordered scatter, taken by production (tile_culling 1) beyond 262144 splats when the grid has at most 16384 quadrants,
 * P / chunks <= 255 and the chunk-prefix table stays below 1 GiB; 2 = round 1's per-tile bitonic sort (tile_culling 5 only).
 * `capacity` counts the instances the path produces: tile instances on paths 0 and 2, quadrant-stream entries on path 1.   */
int gsr_binning_layout(int64_t capacity, int32_t width, int32_t height, int32_t P, int32_t tile_culling, GsrBinningLayout* out);
int gsr_image_layout(int32_t width, int32_t height, GsrImageLayout* out);

/*
 * Forward pass.  Replaces `_C.rasterize_gaussians` (upstream rasterize_points.cu
 * RasterizeGaussiansCUDA -> CudaRasterizer::Rasterizer::forward).
 *   P splats (at most 89 478 485: the blend addresses the 48-byte per-splat records with 32-bit byte offsets; GSR_E_ARG
 *   beyond), M SH coefficients per splat (shs is (P,M,3); 0 when colors_precomp is used).
 *   Exactly one of shs / colors_precomp and exactly one of (scales,rotations) / cov3D_precomp
 *   must be non-NULL.
 *   out_color (3,H,W) and radii (P,) are fully written by the call.
 *   geom / binning / img: caller-allocated state buffers of at least the sizes the layout
 *   functions report for (P), (binning_capacity, W, H), (W, H).
 *   *num_rendered_host receives the number of tile instances I.  If I > binning_capacity the
 *   call returns GSR_E_CAPACITY with nothing valid but *num_rendered_host; re-call with a larger
 *   binning buffer.
 */
int gsr_forward(const GsrSettings* settings, int32_t P, int32_t M,
                const float* means3D, const float* shs, const float* colors_precomp,
                const float* opacities, const float* scales, const float* rotations,
                const float* cov3D_precomp,
                float* out_color, int32_t* radii,
                void* geom, void* binning, int64_t binning_capacity, void* img,
                int64_t* num_rendered_host, void* stream);

/*
 * Backward pass.  Replaces `_C.rasterize_gaussians_backward` (RasterizeGaussiansBackwardCUDA ->
 * CudaRasterizer::Rasterizer::backward).  Inputs are the forward's inputs, its three state
 * buffers (with the binning capacity they were laid out for) and num_rendered, plus dL_dpix (3,H,W).  Every gradient buffer is fully written
 * (zeros for culled splats); pass NULL for dL_dsh when colours were precomputed, for
 * dL_dscales/dL_drotations when cov3D was precomputed.
 *   geom is read AND written: the per-splat accumulators live in it (GsrGeomLayout.acc) and are left zeroed.
 *   dL_dmeans2D (P,3): [:, :2] NDC-scaled screen-space gradient, [:, 2] = 0
 *   dL_dcolors  (P,3): gradient w.r.t. the per-splat RGB (the colors_precomp gradient)
 *   dL_dcov3D   (P,6): gradient w.r.t. the packed 3D covariance
 */
int gsr_backward(const GsrSettings* settings, int32_t P, int32_t M,
                 const float* means3D, const float* shs, const float* colors_precomp,
                 const float* scales, const float* rotations, const float* cov3D_precomp,
                 const int32_t* radii, void* geom, const void* binning, int64_t binning_capacity,
                 const void* img, int64_t num_rendered, const float* dL_dpix,
                 float* dL_dmeans3D, float* dL_dmeans2D, float* dL_dsh, float* dL_dcolors,
                 float* dL_dopacity, float* dL_dscales, float* dL_drotations, float* dL_dcov3D,
                 void* stream);

/*
 * Split-SH variants.  The model keeps the SH coefficients as two leaf tensors,
 * `_features_dc (P,1,3)` and `_features_rest (P,M-1,3)`, and the reference concatenates them every frame
 * (scene/gaussian_model.py:152-156: a 19 MB copy at 100k splats, plus the split of the gradient on the way
 * back).  These entries read / write the two tensors in place: `shs` = DC block, `shs_rest` = the rest, M = total
 * coefficient count (2..16).  With shs_rest == NULL they are exactly gsr_forward / gsr_backward.
 */
int gsr_forward_ex(const GsrSettings* settings, int32_t P, int32_t M,
                   const float* means3D, const float* shs, const float* shs_rest, const float* colors_precomp,
                   const float* opacities, const float* scales, const float* rotations, const float* cov3D_precomp,
                   float* out_color, int32_t* radii,
                   void* geom, void* binning, int64_t binning_capacity, void* img,
                   int64_t* num_rendered_host, void* stream);
int gsr_backward_ex(const GsrSettings* settings, int32_t P, int32_t M,
                    const float* means3D, const float* shs, const float* shs_rest, const float* colors_precomp,
                    const float* scales, const float* rotations, const float* cov3D_precomp,
                    const int32_t* radii, void* geom, const void* binning, int64_t binning_capacity,
                    const void* img, int64_t num_rendered, const float* dL_dpix,
                    float* dL_dmeans3D, float* dL_dmeans2D, float* dL_dsh, float* dL_dsh_rest, float* dL_dcolors,
                    float* dL_dopacity, float* dL_dscales, float* dL_drotations, float* dL_dcov3D,
                    void* stream);

/* Replaces `_C.mark_visible` (upstream markVisible / checkFrustum): present[i] = view-space z > 0.2.
 * Unused by GaussianAvatars (no call site) but part of the package surface. */
int gsr_mark_visible(int32_t P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                     uint8_t* present, void* stream);

/*
 * Optional per-kernel timing with hipEvents recorded on the launch stream (what bench.py's
 * `roofline` object is computed from).  Off by default; when on, every kernel launched by
 * gsr_forward / gsr_backward is bracketed by an event pair.  gsr_profile_read synchronises the
 * pending events and ADDS their elapsed times into total_ms[id] / launches[id]
 * (arrays of GSR_NUM_KERNELS), then forgets them.
 */
#define GSR_K_PREPROCESS 0
#define GSR_K_TILE_SCAN 1
#define GSR_K_SCATTER 2       /* rank path, one band: k_rsort_rscatter (the depth sort rides beside the scatter)       */
#define GSR_K_TILE_SORT 3
#define GSR_K_RENDER 4
#define GSR_K_RENDER_BWD 5
#define GSR_K_PREPROCESS_BWD 6
#define GSR_K_COUNT 7
#define GSR_K_DEPTH_SORT 8    /* k_dbucket + k_dscan + k_dscatter + k_dsort / rank path: k_rdscatter (+ tile scan), bands: + k_rdsort, k_band_* */
#define GSR_K_QCOUNT 9
#define GSR_K_QSCAN 10        /* k_qscan + k_qscan_glob                                                            */
#define GSR_K_QSCATTER 11
#define GSR_NUM_KERNELS 12
/* ---- bound entry: scene/gaussian_model.py:113-160 evaluated inside the rasterizer -----------------------
 * For a mesh-bound model the rasterizer's inputs are get_xyz / get_scaling / get_rotation / get_opacity, i.e. the model's leaves
 * carried into world space by the face each splat is bound to.  These two entries take the LEAVES (_xyz, _scaling = log scales,
 * _rotation, _opacity = logits) plus the per-face frames and evaluate that transform where the first kernel reads its inputs
 * (the same arithmetic, bit for bit, as libgab's gab_bind_forward: csrc/bind_math.h), so the world-space tensors are never
 * materialised and the bind launch and its backward's first pass disappear.  Colours come from SH; covariances from scale/rotation.
 * Backward: gradients of the leaves, and `rows` -- every splat's 17 contributions to its face's gradients (d_center 3, d_orien_mat 9,
 * d_scaling 1, d_orien_quat 4, padded to 20 floats), written at the splat's position `slot` in the per-face CSR of the binding
 * (what gab_bind_backward_csr builds) for libgab's per-face reduction gab_bind_backward_faces.
 * binding == NULL (F = 0, face pointers NULL): an UNBOUND model's leaves -- only the three activations of scene/gaussian_model.py:
 * 113-160 (scaling = exp, rotation = normalize, opacity = sigmoid) are applied, the position is the leaf itself, no rows. */
typedef struct GsrBound {
    const void* binding;        /* (P) face index of every splat                                     */
    int32_t binding_is_i64;     /* 0: int32, 1: int64                                                */
    int32_t F;                  /* faces                                                             */
    const float* face_R;        /* (F,3,3) face_orien_mat                                            */
    const float* face_scale;    /* (F,1)   face_scaling                                              */
    const float* face_center;   /* (F,3)   face_center                                               */
    const float* face_quat;     /* (F,4)   face_orien_quat, WXYZ                                     */
    const int32_t* slot;        /* backward: (P) position of every splat in the per-face CSR         */
    float* rows;                /* backward: (P,20) floats, fully written                            */
} GsrBound;
int gsr_forward_bound(const GsrSettings* settings, int32_t P, int32_t M, const GsrBound* bound, const float* xyz_local, const float* shs,
                      const float* shs_rest, const float* opacity_logit, const float* log_scales, const float* rot_local,
                      float* out_color, int32_t* radii, void* geom, void* binning, int64_t binning_capacity, void* img,
                      int64_t* num_rendered_host, void* stream);
/* scratch9: 9 P floats, kept in the signature (rounds 2-3 parked the colour / covariance gradients of the world-space entry there; since round 4
 * nothing is written to it: a leaves entry has no consumer for those two, 36 bytes per splat less to store) */
int gsr_backward_bound(const GsrSettings* settings, int32_t P, int32_t M, const GsrBound* bound, const float* xyz_local, const float* shs,
                       const float* shs_rest, const float* opacity_logit, const float* log_scales, const float* rot_local,
                       const int32_t* radii, void* geom, const void* binning, int64_t binning_capacity, const void* img,
                       int64_t num_rendered, const float* dL_dout_color, float* dL_dxyz_local, float* dL_dmeans2D, float* dL_dsh,
                       float* dL_dsh_rest, float* dL_dopacity_logit, float* dL_dlog_scales, float* dL_drot_local, float* scratch9,
                       void* stream);

int gsr_profile_enable(int on);
int gsr_profile_read(double* total_ms, int64_t* launches);
const char* gsr_kernel_name(int id);

/* Host-side diagnostic: time gsr_forward spent waiting for the device to post the instance count (the frame's only
 * host wait) and the number of waits, accumulated since the previous call.  A wait near zero means the host, not the
 * GPU, paces the frame loop. */
int gsr_wait_stats(double* total_wait_ms, int64_t* waits);

#ifdef __cplusplus
}
#endif
#endif /* GSR_H */
