// gsr_api.hip -- the C ABI of include/gsr.h: argument checks, state-buffer layouts, the host
// mailbox for the instance count, and the launch sequence.  No torch types; callers hand in raw
// device pointers and a hipStream_t.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <tuple>
#include <thread>
#include <vector>

#include "gsr_device.h"


namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                                      \
    do {                                                                                                   \
        hipError_t e_ = (expr);                                                                            \
        if (e_ != hipSuccess) return fail(GSR_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

#define KERNEL_CHECK(name, stream, debug)                                                                  \
    do {                                                                                                   \
        hipError_t e_ = hipGetLastError();                                                                 \
        if (e_ != hipSuccess) return fail(GSR_E_HIP, "launch of %s failed: %s", name, hipGetErrorString(e_)); \
        if (debug) {                                                                                       \
            e_ = hipStreamSynchronize(stream);                                                             \
            if (e_ != hipSuccess) return fail(GSR_E_HIP, "%s faulted: %s", name, hipGetErrorString(e_));   \
        }                                                                                                  \
    } while (0)

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---- host mailbox: a small ring of 8-byte slots in mapped pinned memory -----------------------
// k_tile_scan posts (seq << 40 | I) with one system-scope store; gsr_forward spins on the slot.
struct Mailbox {
    static constexpr int kSlots = 256;        // ring of the waiting forwards
    static constexpr int kPersistent = 2 * GSR_COUNT_SLOTS;   // after the ring: the count slots of the deferred forwards (GsrSettings.deferred_count),
                                                              // then one STICKY overflow word per slot (set by the device, cleared by the host only)
    unsigned long long* host = nullptr;
    unsigned long long* dev = nullptr;        // the same memory as the device sees it
    std::atomic<unsigned long long> seq{1};
    std::mutex init_mu;
    bool ready = false;
    int init()
    {
        std::lock_guard<std::mutex> lk(init_mu);
        if (ready) return 0;
        HIP_TRY(hipHostMalloc((void**)&host, (kSlots + kPersistent) * sizeof(unsigned long long), hipHostMallocMapped | hipHostMallocPortable));
        memset(host, 0, (kSlots + kPersistent) * sizeof(unsigned long long));
        HIP_TRY(hipHostGetDevicePointer((void**)&dev, (void*)host, 0));
        ready = true;
        return 0;
    }
};
Mailbox g_mail;
std::atomic<long long> g_wait_ns{0}, g_waits{0};
thread_local long long g_last_seq = 0;   // the sequence number of this thread's newest forward (gsr_last_forward_seq)

// ---- optional per-kernel event timing -------------------------------------------------------
struct Profiler {
    std::atomic<int> on{0};
    std::mutex mu;
    struct Pending { int id; hipEvent_t a, b; };
    std::vector<Pending> pending;
    std::vector<hipEvent_t> pool;
    hipEvent_t get()
    {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e = nullptr;
        (void)hipEventCreate(&e);
        return e;
    }
};
Profiler g_prof;

struct ScopedKernelTimer {
    int id; hipStream_t st; hipEvent_t a = nullptr, b = nullptr; bool active;
    ScopedKernelTimer(int id_, hipStream_t s) : id(id_), st(s), active(g_prof.on.load() != 0)
    {
        if (!active) return;
        std::lock_guard<std::mutex> lk(g_prof.mu);
        a = g_prof.get(); b = g_prof.get();
        (void)hipEventRecord(a, st);
    }
    ~ScopedKernelTimer()
    {
        if (!active) return;
        (void)hipEventRecord(b, st);
        std::lock_guard<std::mutex> lk(g_prof.mu);
        g_prof.pending.push_back({id, a, b});
    }
};
#define TIMED(id, stream) ScopedKernelTimer timer_##id(id, stream)

gsr::Settings to_dev_settings(const GsrSettings* s)
{
    gsr::Settings d;
    d.H = s->image_height;
    d.W = s->image_width;
    d.tanfovx = s->tanfovx;
    d.tanfovy = s->tanfovy;
    d.scale_modifier = s->scale_modifier;
    d.sh_degree = s->sh_degree;
    d.exact_scale_grad = s->exact_scale_grad;
    d.forward_only = s->forward_only;
    d.deterministic = s->deterministic;
    d.fast_blend = 0;   // the callers set the effective mode once the binning path is known (fast_effective)
    // fast blend: the entry (a multiple of the backward's segment) at which a quadrant's lone walk parks its state for the continuation kernel
    // (gsr_forward.hip, k_render<true, true>); GSR_CONT_CHUNKS=0 keeps every walk in one piece (A/B runs)
    static const int cont_chunks = [] { const char* e = getenv("GSR_CONT_CHUNKS"); const int v = e ? atoi(e) : GSR_CONT_CHUNKS_DEFAULT; return v < 0 ? 0 : v; }();
    d.cont_chunks = cont_chunks;
    static const int cont_mode = [] { const char* e = getenv("GSR_CONT_MODE"); const int v = e ? atoi(e) : GSR_CONT_MODE_DEFAULT; return v == 2 ? 2 : 1; }();
    d.cont_mode = cont_mode;
    d.bg = s->bg;
    d.viewmatrix = s->viewmatrix;
    d.projmatrix = s->projmatrix;
    d.campos = s->campos;
    return d;
}

int check_settings(const GsrSettings* s)
{
    if (!s) return fail(GSR_E_ARG, "settings is NULL");
    if (s->image_height <= 0 || s->image_width <= 0) return fail(GSR_E_ARG, "image size must be positive");
    if (s->image_height > 65535 * 16 || s->image_width > 65535 * 16) return fail(GSR_E_ARG, "image too large for 16-bit tile rects");
    if (!s->bg || !s->viewmatrix || !s->projmatrix || !s->campos) return fail(GSR_E_ARG, "bg/viewmatrix/projmatrix/campos must be device pointers");
    if (s->sh_degree < 0 || s->sh_degree > 3) return fail(GSR_E_ARG, "sh_degree must be in 0..3");
    if (s->deferred_count < 0 || s->deferred_count > GSR_COUNT_SLOTS) return fail(GSR_E_ARG, "deferred_count must be 0 or a slot in 1..%d", GSR_COUNT_SLOTS);
    return 0;
}

// GsrSettings.fast_blend as it applies to a frame: the exact kernels serve the deterministic mode (whose fixed-point scales are
// derived for the exact partial sums) and the per-tile sort path (whose binning kernels read the conic from the per-splat record,
// where the fast mode keeps it pre-scaled).  The forward and the backward of a frame evaluate this on the same inputs.
bool fast_effective(const GsrSettings* s, const GsrBinningLayout& bl) { return s->fast_blend != 0 && s->deterministic == 0 && bl.path != 2; }

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, size class): the attribute sticks, and the call is not
// something to repeat per frame (nor inside a stream capture)
int ensure_dynamic_lds(const void* fn, size_t bytes)
{
    struct Seen { const void* fn; int device; size_t bytes; };
    static std::mutex mu;
    static std::vector<Seen> seen;
    int device = 0;
    HIP_TRY(hipGetDevice(&device));   // the attribute belongs to the kernel's code object on ONE device
    std::lock_guard<std::mutex> lk(mu);
    for (auto& e : seen)
        if (e.fn == fn && e.device == device) {
            if (e.bytes >= bytes) return 0;
            HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
            e.bytes = bytes;
            return 0;
        }
    HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    seen.push_back({fn, device, bytes});
    return 0;
}

// ---- the posted word: seq << 40 | flag << 39 | I.  The flag is the rank path's "the tile instances are spread unevenly along the splat order" (k_rcount from
// k_preprocess's per-workgroup sums): the NEXT frames of that (device, image size) deal their rank-pass chunks out in groups of GSR_RANK_ILV_AUTO splats
// instead of contiguous runs (gsr_device.h: rank_chunk).  A running hint like the callers' capacity estimate: it changes how long a frame takes, never what it computes.
constexpr unsigned long long kCountMask = 0x7FFFFFFFFFull;
std::mutex g_ilv_mutex;
std::map<std::tuple<int, int, int>, int> g_ilv_hint;        // (device, W, H) -> uneven?
std::tuple<int, int, int> g_slot_frame[GSR_COUNT_SLOTS + 1]; // the frame whose count a slot holds (index GSR_COUNT_SLOTS: the blocking call's own slot)
void note_posted(int slot_index, unsigned long long v)
{
    if (!v) return;
    std::lock_guard<std::mutex> lock(g_ilv_mutex);
    const auto key = g_slot_frame[slot_index];
    if (std::get<1>(key) > 0) g_ilv_hint[key] = (int)((v >> 39) & 1ull);
}
int rank_ilv_for(int device, int W, int H)
{
    static const int forced = [] { const char* e = getenv("GSR_RANK_ILV"); return e ? atoi(e) : -1; }();   // -1 (default): from the previous frame; 0: contiguous; n: groups of n
    if (forced >= 0) return forced;
    std::lock_guard<std::mutex> lock(g_ilv_mutex);
    auto it = g_ilv_hint.find(std::make_tuple(device, W, H));
    return it != g_ilv_hint.end() && it->second ? GSR_RANK_ILV_AUTO : 0;
}
void slot_holds_frame(int slot_index, int device, int W, int H)
{
    std::lock_guard<std::mutex> lock(g_ilv_mutex);
    g_slot_frame[slot_index] = std::make_tuple(device, W, H);
}

}  // namespace

extern "C" {

int gsr_abi_version(void) { return GSR_ABI_VERSION; }
const char* gsr_last_error(void) { return g_err; }

int gsr_count_slot_read(int32_t slot, int64_t* count, int64_t* seq)
{
    if (slot < 0 || slot >= GSR_COUNT_SLOTS || !count) return fail(GSR_E_ARG, "gsr_count_slot_read: bad arguments");
    if (int rc = g_mail.init()) return rc;
    const unsigned long long v = __atomic_load_n(g_mail.host + Mailbox::kSlots + slot, __ATOMIC_ACQUIRE);
    *count = v ? (int64_t)(v & kCountMask) : -1;
    if (seq) *seq = (int64_t)(v >> 40);
    note_posted(slot, v);
    return GSR_OK;
}

int64_t gsr_last_forward_seq(void) { return g_last_seq; }

int gsr_count_slot_wait(int32_t slot, int64_t want_seq, void* stream_, int64_t* count)
{
    if (slot < 0 || slot >= GSR_COUNT_SLOTS || !count || want_seq <= 0) return fail(GSR_E_ARG, "gsr_count_slot_wait: bad arguments");
    if (int rc = g_mail.init()) return rc;
    hipStream_t stream = (hipStream_t)stream_;
    const unsigned long long* w = g_mail.host + Mailbox::kSlots + slot;
    const auto t0 = std::chrono::steady_clock::now();
    unsigned long long v;
    unsigned spins = 0;
    for (;;) {
        v = __atomic_load_n(w, __ATOMIC_ACQUIRE);
        if (v != 0 && (long long)(v >> 40) == want_seq) break;
        if (++spins > 2000) {
            std::this_thread::yield();
            if ((spins & 0xFFF) == 0) {
                if (hipStreamQuery(stream) != hipErrorNotReady) {   // stream drained (or faulted) without the post
                    v = __atomic_load_n(w, __ATOMIC_ACQUIRE);
                    if (v != 0 && (long long)(v >> 40) == want_seq) break;
                    hipError_t e = hipStreamSynchronize(stream);
                    return fail(GSR_E_HIP, "stream finished without posting the instance count: %s", hipGetErrorString(e));
                }
                if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120))
                    return fail(GSR_E_TIMEOUT, "timed out waiting for the instance count");
            }
        }
    }
    g_wait_ns.fetch_add((long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count());
    g_waits.fetch_add(1);
    *count = (int64_t)(v & kCountMask);
    note_posted(slot, v);
    return GSR_OK;
}

int gsr_count_slot_overflow(int32_t slot, int64_t* worst, int32_t reset)
{
    if (slot < 0 || slot >= GSR_COUNT_SLOTS) return fail(GSR_E_ARG, "gsr_count_slot_overflow: bad slot");
    if (int rc = g_mail.init()) return rc;
    unsigned long long* w = g_mail.host + Mailbox::kSlots + GSR_COUNT_SLOTS + slot;
    // read-and-clear is ONE exchange: a frame still in flight that marks the slot between a load and a store would lose its report
    const unsigned long long v = reset ? __atomic_exchange_n(w, 0ull, __ATOMIC_ACQ_REL) : __atomic_load_n(w, __ATOMIC_ACQUIRE);
    if (worst) *worst = (int64_t)v;
    return GSR_OK;
}

int gsr_geom_layout(int32_t P, GsrGeomLayout* o)
{
    if (!o || P < 0) return fail(GSR_E_ARG, "gsr_geom_layout: bad arguments");
    const size_t n = (size_t)P, A = 256;
    size_t off = 0;
    o->depths = off;        off = align_up(off + n * 4, A);
    o->grec = off;          off = align_up(off + n * 48, A);
    o->cov3D = off;         off = align_up(off + n * 24, A);
    o->rect = off;          off = align_up(off + n * 8, A);
    o->tiles_touched = off; off = align_up(off + n * 4, A);
    o->clamped = off;       off = align_up(off + n, A);
    o->visible = off;       off = align_up(off + n, A);
    o->brec = off;          off = align_up(off + n * 48, A);
    o->acc64 = off;         off = align_up(off + n * GSR_ACC64_STRIDE * 8, A);
    o->acc = off;           off = align_up(off + n * GSR_ACC_STRIDE * 4, A);
    o->total = off + A;
    return 0;
}

// Production binning parameters for (P, grid): chunks of the ordered walk (one wave each, S = ceil(P / chunks) <= 255 splats so
// that the per-quadrant counters of a chunk fit a byte) and depth buckets (about 512 splats each).  Returns false when the
// depth-ordered scatter does not apply (then the rank path is used).
static bool production_params(int32_t P, size_t tiles, int32_t tile_culling, size_t* chunks, size_t* nb)
{
    *chunks = 0; *nb = 0;
    // up to GSR_RANK_MAX_SPLATS the rank path is the faster one (200 k splats, 550x802: 1565 against 1215 frames/s); beyond it the
    // tile bitmaps no longer hold a frame's ranks and the depth-ordered scatter takes over where its tables fit
    if (tile_culling == 4 ? P <= 0 : (tile_culling != 1 || P <= GSR_RANK_MAX_SPLATS)) return false;   // 4: whenever it applies (tests, A/B)
    const size_t Q = 4 * tiles;
    if (Q > 16384) return false;                                   // the scatter wave keeps a 4-byte cursor per quadrant in LDS: <= 64 KB
    size_t c = ((size_t)P + 127) / 128;                           // ~128 splats per chunk: each of the chunk's 4 band waves walks them
    c = c < 64 ? 64 : (c > 4096 ? 4096 : c);
    if (((size_t)P + c - 1) / c > 255) return false;               // byte counters
    if (c * Q * 4 > ((size_t)1 << 30)) return false;               // qprefix table
    size_t b = 16;
    while (b < 8192 && b * 512 < (size_t)P) b <<= 1;
    *chunks = c; *nb = b;
    return true;
}
// Binning path of a frame: 1 production (depth-ordered scatter, gsr_binning.hip), 2 round 1's per-tile bitonic sort (tile_culling 5:
// kept for A/B runs), 0 the rank path (gsr_rank.hip: splats ranked by depth once, tiles ordered through an LDS bitmap).
static int binning_path(int32_t P, size_t tiles, int32_t tile_culling, size_t* chunks, size_t* nb)
{
    if (production_params(P, tiles, tile_culling, chunks, nb)) return 1;
    if (tile_culling == 5) return 2;
    // about 256 splats per depth bucket (GSR_RANK_BUCKET_SPLATS: experiment knob, a power of two up to 1024 -- a bucket is sorted in the registers
    // of 256 threads up to 2048 keys)
    static const size_t per = [] { const char* e = getenv("GSR_RANK_BUCKET_SPLATS"); const long v = e ? atol(e) : 0; return (size_t)(v >= 64 && v <= 1024 ? v : 256); }();
    size_t b = 16;
    while (b < (size_t)GSR_RANK_MAX_BUCKETS && b * per < (size_t)P) b <<= 1;
    *nb = b;
    return 0;
}

int gsr_binning_layout(int64_t capacity, int32_t width, int32_t height, int32_t P, int32_t tile_culling, GsrBinningLayout* o)
{
    if (!o || capacity < 0 || width <= 0 || height <= 0 || P < 0) return fail(GSR_E_ARG, "gsr_binning_layout: bad arguments");
    const size_t tiles = (size_t)((width + GSR_BLOCK_X - 1) / GSR_BLOCK_X) * (size_t)((height + GSR_BLOCK_Y - 1) / GSR_BLOCK_Y);
    const size_t cap = (size_t)capacity, A = 256, n = (size_t)P;
    size_t chunks, nb;
    const int path = binning_path(P, tiles, tile_culling, &chunks, &nb);
    const bool prod = path == 1, rankp = path == 0, old = path == 2;
    const bool lists = tile_culling == 0 || tile_culling == 2;     // the reference-format key / point lists are materialised
    const bool dsort = prod || rankp;                              // the splats are depth-sorted
    const size_t Q = 4 * tiles, qw = (Q + 3) / 4;
    const int lgx = (width + GSR_BLOCK_X - 1) / GSR_BLOCK_X;
    const bool no_hist = rankp ? gsr::rank_direct(lgx, (int)tiles) : tiles > (size_t)GSR_LDS_HIST_TILES;   // no per-workgroup histograms: L2 atomics
    size_t off = align_up(sizeof(gsr::BinHeader), A);  // header + statistics slots (gsr_device.h: BinHeader)
    o->keys = off;        off = align_up(off + (old || (rankp && lists) ? cap * 8 : 0), A);
    o->point_list = off;  off = align_up(off + (old || rankp ? cap * 4 : 0), A);
    o->qlist = off;       off = align_up(off + (!prod && lists ? cap * 4 * 4 : 0), A);
    o->qpos = off;        off = align_up(off + (prod ? cap * 4 : cap * 4 * 4), A);
    o->qcount = off;      off = align_up(off + tiles * 16, A);
    o->qstart = off;      off = align_up(off + tiles * 16, A);
    o->ranges = off;      off = align_up(off + (prod ? 0 : tiles * 8), A);
    o->tile_count = off;  off = align_up(off + (prod ? 0 : tiles * 4), A);
    o->tile_start = off;  off = align_up(off + (prod ? 0 : tiles * 4), A);
    o->tile_cursor = off; off = align_up(off + (prod ? 0 : tiles * 4), A);
    o->tile_order = off;  off = align_up(off + tiles * 4, A);
    const size_t blocks = rankp ? (size_t)GSR_RANK_BLOCKS : (size_t)GSR_BIN_BLOCKS;
    o->block_hist = off;  off = align_up(off + (prod || no_hist ? 0 : blocks * tiles * 4), A);
    o->dkeys = off;       off = align_up(off + (dsort ? n * 8 : 0), A);
    o->dtmp = off;        off = align_up(off + (dsort ? n * 8 : 0), A);
    o->order = off;       off = align_up(off + (prod ? n * 4 : 0), A);
    o->bcount = off;      off = align_up(off + (dsort ? nb * 4 : 0), A);
    o->bstart = off;      off = align_up(off + (dsort ? nb * 4 : 0), A);
    o->bcursor = off;     off = align_up(off + (dsort ? nb * 4 : 0), A);
    o->border = off;      off = align_up(off + (prod ? nb * 4 : 0), A);
    o->bhist = off;       off = align_up(off + (dsort ? blocks * nb * 4 : 0), A);
    o->qhist = off;       off = align_up(off + (prod ? chunks * qw * 4 : 0), A);
    o->qprefix = off;     off = align_up(off + (prod ? chunks * Q * 4 : 0), A);
    o->qmask = off;       off = align_up(off + (prod ? n * GSR_WALK_MASKS * 8 : 0), A);
    o->ranks = off;       off = align_up(off + (rankp ? cap * 8 : 0), A);
    int band_rows = 1;
    const size_t nbands = rankp ? (size_t)gsr::rank_bands((long long)P, (int)(tiles / (size_t)lgx), tile_culling == 6, &band_rows) : 1;
    const size_t nwc = (n + GSR_RANK_BAND_CHUNK - 1) / GSR_RANK_BAND_CHUNK;
    const bool bands = rankp && nbands > 1;
    o->rank = off;        off = align_up(off + (rankp ? n * (bands ? 16 : 4) : 0), A);
    o->rank_over = off;   off = align_up(off + (bands ? n * nbands * 4 : 0), A);
    o->obs = off;         off = align_up(off + (bands ? n * 8 : 0), A);
    o->bandcnt = off;     off = align_up(off + (bands ? nbands * nwc * 4 : 0), A);
    o->srect = off;       off = align_up(off + (rankp ? n * 8 : 0), A);
    o->sspan = off;       off = align_up(off + (rankp ? n * 32 : 0), A);
    o->pstat = off;       off = align_up(off + (rankp ? ((n + 255) / 256) * 80 + (GSR_RANK_BLOCKS + 4) * 4 : 0), A);   // (stats, four planes of group sums, k_rcount's chunk boundaries)
    o->tdesc = off;       off = align_up(off + (rankp ? tiles * 16 : 0), A);
    o->path = (size_t)path;
    o->chunks = chunks;
    o->nb = nb;
    o->nbands = nbands;
    o->band_rows = (size_t)band_rows;
    o->total = off + A;
    return 0;
}

int gsr_image_layout(int32_t width, int32_t height, GsrImageLayout* o)
{
    if (!o || width <= 0 || height <= 0) return fail(GSR_E_ARG, "gsr_image_layout: bad arguments");
    const size_t hw = (size_t)width * (size_t)height, A = 256;
    size_t off = 0;
    o->final_T = off;   off = align_up(off + hw * 4, A);
    o->n_contrib = off; off = align_up(off + hw * 4, A);
    o->n_contrib_q = off; off = align_up(off + hw * 4, A);
    o->c_final = off;   off = align_up(off + hw * 12, A);
    o->ck = off;        off = align_up(off + hw * 16 * (GSR_BWD_SEGMENTS - 1), A);
    o->gmax = off;      off = align_up(off + 4, A);
    const size_t tiles = (size_t)((width + GSR_BLOCK_X - 1) / GSR_BLOCK_X) * (size_t)((height + GSR_BLOCK_Y - 1) / GSR_BLOCK_Y);
    // the unit lists, then the forward blend's continuation area (gsr_forward.hip: header, the list of parked quadrants, 1280 bytes of state per quadrant slot)
    o->units = off;     off = align_up(off + (gsr::cont_hdr_word(tiles) + GSR_CONT_HDR_WORDS + 4 * tiles + 4 * tiles * (size_t)GSR_CONT_STATE_FLOATS) * 4, A);
    o->total = off + A;
    return 0;
}

static int to_bound(const GsrBound* b, int32_t P, bool backward, gsr::BoundDev* o)
{
    o->leaves = 0; o->binding = nullptr; o->is64 = 0; o->fR = o->fs = o->fc = o->fq = nullptr; o->slot = nullptr; o->rows = nullptr;
    if (!b) return 0;
    o->leaves = 1;
    if (!b->binding) {   // an unbound model's leaves: activations only
        if (P > 0 && (b->F != 0 || b->face_R || b->face_scale || b->face_center || b->face_quat)) return fail(GSR_E_ARG, "GsrBound: face buffers without a binding");   // (P == 0: an empty binding has no address)
        return 0;
    }
    if (b->F <= 0 || (P > 0 && (!b->face_R || !b->face_scale || !b->face_center || !b->face_quat)))
        return fail(GSR_E_ARG, "GsrBound: NULL face buffer or F <= 0");
    if (backward && P > 0 && (!b->slot || !b->rows)) return fail(GSR_E_ARG, "GsrBound: the backward needs slot and rows");
    o->binding = b->binding; o->is64 = b->binding_is_i64;
    o->fR = b->face_R; o->fs = b->face_scale; o->fc = b->face_center; o->fq = b->face_quat;
    o->slot = b->slot; o->rows = b->rows;
    return 0;
}

static int forward_impl(const GsrSettings* settings, int32_t P, int32_t M, const float* means3D, const float* shs, const float* shs_rest,
                        const float* colors_precomp, const float* opacities, const float* scales, const float* rotations,
                        const float* cov3D_precomp, float* out_color, int32_t* radii, void* geom, void* binning,
                        int64_t binning_capacity, void* img, int64_t* num_rendered_host, void* stream_, const GsrBound* bound)
{
    if (int rc = check_settings(settings)) return rc;
    hipStream_t stream = (hipStream_t)stream_;
    if (P < 0) return fail(GSR_E_ARG, "P must be >= 0");
    if ((long long)P * 48 > 0xFFFFFFFFll) return fail(GSR_E_ARG, "P = %d exceeds 89478485 splats: the blend addresses the 48-byte per-splat records with 32-bit offsets", P);
    if (!out_color || !num_rendered_host) return fail(GSR_E_ARG, "out_color / num_rendered_host is NULL");
    if (P > 0) {   // with no splats there is nothing to point at: empty tensors legitimately arrive as NULL
        if ((shs == nullptr) == (colors_precomp == nullptr))
            return fail(GSR_E_ARG, "Please provide excatly one of either SHs or precomputed colors!");
        const bool has_sr = scales != nullptr && rotations != nullptr;
        if (((scales == nullptr) != (rotations == nullptr)) || (has_sr == (cov3D_precomp != nullptr)))
            return fail(GSR_E_ARG, "Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!");
    }
    if (shs_rest && (!shs || M < 2 || M > 16)) return fail(GSR_E_ARG, "split SH needs shs (P,1,3) + shs_rest (P,M-1,3) with 2 <= M <= 16");
    if (shs && M < (settings->sh_degree + 1) * (settings->sh_degree + 1))
        return fail(GSR_E_ARG, "shs has %d coefficients per splat but sh_degree %d needs %d", M, settings->sh_degree,
                    (settings->sh_degree + 1) * (settings->sh_degree + 1));
    if (P > 0 && (!means3D || !opacities || !radii || !geom)) return fail(GSR_E_ARG, "NULL splat buffer");
    if (!binning || !img) return fail(GSR_E_ARG, "NULL state buffer");
    if (binning_capacity < 0 || binning_capacity >= (1ll << 32)) return fail(GSR_E_ARG, "binning_capacity out of range");
    {   // the rank and per-tile sort paths address the four quadrant streams of a tile as 32-bit ELEMENT offsets 4 * start + q * n into qpos
        GsrBinningLayout probe;
        gsr_binning_layout(0, settings->image_width, settings->image_height, P, settings->tile_culling, &probe);
        if (probe.path != 1 && binning_capacity > (1ll << 30))
            return fail(GSR_E_ARG, "binning_capacity %lld exceeds 2^30 tile instances (32-bit quadrant-stream offsets on this binning path)", (long long)binning_capacity);
    }

    const int W = settings->image_width, H = settings->image_height;
    const int gx = (W + GSR_BLOCK_X - 1) / GSR_BLOCK_X, gy = (H + GSR_BLOCK_Y - 1) / GSR_BLOCK_Y;
    const int tiles = gx * gy;
    const bool dbg = settings->debug != 0;
    if (int rc = g_mail.init()) return rc;

    GsrGeomLayout gl;
    GsrBinningLayout bl;
    GsrImageLayout il;
    gsr_geom_layout(P, &gl);
    gsr_binning_layout(binning_capacity, W, H, P, settings->tile_culling, &bl);
    gsr_image_layout(W, H, &il);
    char* g = (char*)geom;
    char* b = (char*)binning;
    char* im = (char*)img;
    unsigned long long* total_dev = (unsigned long long*)b;
    gsr::BinHeader* hdr = (gsr::BinHeader*)b;
    uint32_t* tile_count = (uint32_t*)(b + bl.tile_count);
    unsigned long long* rect_total = total_dev + 1;   // header word 1: sum of tiles_touched (the reference's num_rendered)
    const bool prod = bl.path == 1;                   // depth-ordered scatter into the quadrant streams (gsr_binning.hip)

    gsr::Settings ds = to_dev_settings(settings);
    ds.fast_blend = fast_effective(settings, bl) ? 1 : 0;
    if (prod) {
        HIP_TRY(hipMemsetAsync(hdr, 0, sizeof(gsr::BinHeader), stream));   // k_preprocess accumulates the frame statistics into it
    } else if (P == 0) {   // otherwise k_preprocess zeroes both
        HIP_TRY(hipMemsetAsync(tile_count, 0, (size_t)tiles * 4, stream));
        HIP_TRY(hipMemsetAsync(rect_total, 0, 8, stream));
        HIP_TRY(hipMemsetAsync(im + il.units, 0, (size_t)32 * GSR_UNIT_LISTS * 4, stream));
        HIP_TRY(hipMemsetAsync(im + il.units + gsr::cont_hdr_word((size_t)tiles) * 4, 0, (size_t)GSR_CONT_HDR_WORDS * 4, stream));
        HIP_TRY(hipMemsetAsync(im + il.units + (gsr::cont_hdr_word((size_t)tiles) + GSR_CONT_HDR_WORDS) * 4, 0xFF, (size_t)4 * tiles * 4, stream));
    }

    gsr::PreprocessArgs pa;
    pa.P = P; pa.M = M;
    pa.means3D = means3D; pa.shs = shs; pa.shs_rest = shs_rest; pa.colors_precomp = colors_precomp; pa.opacities = opacities;
    pa.scales = scales; pa.rotations = rotations; pa.cov3D_precomp = cov3D_precomp;
    pa.radii = radii;
    pa.depths = (float*)(g + gl.depths);
    pa.grec = (float4*)(g + gl.grec);
    pa.cov3D = (float*)(g + gl.cov3D);
    pa.rect = (ushort4*)(g + gl.rect);
    pa.tiles_touched = (uint32_t*)(g + gl.tiles_touched);
    pa.clamped = (uint8_t*)(g + gl.clamped);
    pa.visible = (uint8_t*)(g + gl.visible);
    pa.acc = (float4*)(g + gl.acc);
    pa.acc64 = (float4*)(g + gl.acc64);
    pa.tile_count = tile_count;
    pa.units = (uint32_t*)(im + il.units);
    pa.rect_total = rect_total;
    pa.tiles = tiles;
    pa.brec = prod ? (float4*)(g + gl.brec) : nullptr;
    pa.hdr = hdr;
    pa.bcount = (uint32_t*)(b + bl.bcount);
    pa.nb = (int)bl.nb;
    const bool rankp = bl.path == 0;                  // tiles ordered by global depth rank (gsr_rank.hip)
    pa.bcursor = (uint32_t*)(b + bl.bcursor);
    pa.pstat = rankp ? (uint4*)(b + bl.pstat) : nullptr;
    pa.srect = (ushort4*)(b + bl.srect);
    pa.sspan = (float4*)(b + bl.sspan);
    pa.cull = settings->tile_culling != 0;
    if (int rc = to_bound(bound, P, false, &pa.bound)) return rc;
    const int pblocks = (P + 255) / 256;
    if (pblocks > 0) {
        TIMED(GSR_K_PREPROCESS, stream);
        hipLaunchKernelGGL(gsr::k_preprocess, dim3(pblocks), dim3(256), 0, stream, ds, pa);
        KERNEL_CHECK("k_preprocess", stream, dbg);
    }

    const unsigned long long seq = (g_mail.seq.fetch_add(1) % 0xFFFFFEull) + 1;  // 1 .. 2^24-2, never 0
    g_last_seq = (long long)seq;
    // where the scan posts (seq, I): a ring slot this call spins on, or -- deferred count -- the caller's persistent slot, read
    // later through gsr_count_slot_read (nothing waits, so the call can sit inside a stream capture and be replayed as a hipGraph)
    const bool deferred = settings->deferred_count != 0;
    const size_t slot_index = deferred ? (size_t)Mailbox::kSlots + (size_t)(settings->deferred_count - 1) : (size_t)(seq % Mailbox::kSlots);
    volatile unsigned long long* slot = g_mail.host + slot_index;
    *slot = 0;   // (a persistent slot may still hold what its previous owner's last frame posted)
    unsigned long long* slot_dev = g_mail.dev + slot_index;
    const unsigned long long post_cap = deferred ? (unsigned long long)binning_capacity : ~0ull;   // deferred: an overflowing frame marks the slot's sticky word
    // rank path: contiguous chunks of splats per workgroup, or -- when the previous frame of this (device, size) reported its tile instances unevenly spread
    // along the splat order -- small groups dealt round-robin (gsr_device.h: rank_chunk / rank_splat; this frame's own report rides in the posted count)
    int dev_id = 0, rank_ilv = -1;
    if (rankp) {
        HIP_TRY(hipGetDevice(&dev_id));
        const int ilv = rank_ilv_for(dev_id, settings->image_width, settings->image_height);   // 0: contiguous, else a group size (rounded down to a power of two)
        rank_ilv = -1;
        for (int v = ilv; v > 0; v >>= 1) ++rank_ilv;                                                // log2, -1 for contiguous
        // ... and on frames of one band the answer to an uneven spread is not the interleave but contiguous chunks of equal WEIGHT, cut by k_rcount from
        // k_preprocess's group sums (gsr_device.h: rank_map; round 6): the template-like head's scatter 44 -> 31 us where the interleave had brought it from
        // 88 to 44, and k_rcount keeps its compact set of tiles.  (It costs k_rcount 3.5 us of prologue, so an even frame keeps its equal-count chunks.)
        // GSR_RANK_BALANCED=0: the interleave; =2: balanced chunks on every frame (A/B runs)
        static const int balanced = [] { const char* e = getenv("GSR_RANK_BALANCED"); return e ? atoi(e) : 1; }();
        static const bool forced_ilv = getenv("GSR_RANK_ILV") != nullptr;
        const bool can = (long long)P <= (long long)GSR_RANK_MAX_SPLATS && (int)bl.nbands <= 1 && !forced_ilv;
        if (can && ((balanced == 1 && ilv > 0) || balanced == 2)) rank_ilv = -2;
    }
    slot_holds_frame(deferred ? settings->deferred_count - 1 : GSR_COUNT_SLOTS, dev_id, rankp ? settings->image_width : 0, settings->image_height);
    const unsigned long long cap = (unsigned long long)binning_capacity;
    uint32_t* tile_order = (uint32_t*)(b + bl.tile_order);
    uint32_t* qpos = (uint32_t*)(b + bl.qpos);
    uint32_t* qcount = (uint32_t*)(b + bl.qcount);
    uint32_t* qstart = (uint32_t*)(b + bl.qstart);
    const bool write_lists = settings->tile_culling == 0 || settings->tile_culling == 2;   // production (1, 3, 4): the sorted key / point lists are not materialised
    uint32_t* qlist = (uint32_t*)(b + bl.qlist);

    if (prod) {
        // ---- production: sort the SPLATS by depth, then append them in that order to the quadrant streams ----------------
        const uint32_t nb = (uint32_t)bl.nb, chunks = (uint32_t)bl.chunks;
        const int Q = 4 * tiles;
        uint32_t* bcount = (uint32_t*)(b + bl.bcount);
        uint32_t* bstart = (uint32_t*)(b + bl.bstart);
        uint32_t* bcursor = (uint32_t*)(b + bl.bcursor);
        uint32_t* border = (uint32_t*)(b + bl.border);
        unsigned long long* dkeys = (unsigned long long*)(b + bl.dkeys);
        unsigned long long* dtmp = (unsigned long long*)(b + bl.dtmp);
        uint32_t* order = (uint32_t*)(b + bl.order);
        const uint32_t* brec_rect = (const uint32_t*)(g + gl.brec);
        {
            TIMED(GSR_K_DEPTH_SORT, stream);
            const int dblocks = pblocks < GSR_BIN_BLOCKS ? pblocks : GSR_BIN_BLOCKS;
            uint32_t* bhist = (uint32_t*)(b + bl.bhist);
            hipLaunchKernelGGL(gsr::k_dbucket, dim3(dblocks), dim3(256), (size_t)nb * 4, stream, P, brec_rect, (const float*)pa.depths, hdr, nb,
                               bcount, bhist);
            hipLaunchKernelGGL(gsr::k_dscan, dim3(1), dim3(1024), 0, stream, nb, (const uint32_t*)bcount, bstart, bcursor, border, hdr);
            hipLaunchKernelGGL(gsr::k_dscatter, dim3(dblocks), dim3(256), (size_t)nb * 4, stream, P, brec_rect, (const float*)pa.depths,
                               (const gsr::BinHeader*)hdr, nb, (const uint32_t*)bstart, bcursor, dkeys, (const uint32_t*)bhist);
            // one class: 2048 keys in registers per workgroup, heavier buckets (rare: the bucket map is linear in this frame's own depth
            // range) by chunked sorts + merges in global memory; heaviest buckets first
            hipLaunchKernelGGL((gsr::k_dsort<GSR_SORT_SMALL_KEYS, 256>), dim3(nb), dim3(256), 0, stream, 0u, 0xFFFFFFFFu,
                               (const uint32_t*)border, (const uint32_t*)bcount, (const uint32_t*)bstart, dkeys, dtmp, order);
            KERNEL_CHECK("depth sort", stream, dbg);
        }
        gsr::QBinArgs qa;
        qa.Q = Q; qa.gx = gx; qa.chunks = chunks;
        qa.qmask = (unsigned long long*)(b + bl.qmask);
        qa.hdr = hdr; qa.brec = (const float4*)(g + gl.brec); qa.order = order;
        qa.qhist = (uint32_t*)(b + bl.qhist); qa.qprefix = (const uint32_t*)(b + bl.qprefix); qa.qstart = qstart; qa.qpos = qpos;
        qa.capacity = cap;
        const size_t count_lds = (((size_t)Q + 3) / 4) * 4;   // byte counters, four to a word
        const size_t walk_lds = (size_t)Q * 4;               // absolute cursors
        if (walk_lds > 48 * 1024)
            if (int rc = ensure_dynamic_lds((const void*)gsr::k_qscatter, (size_t)walk_lds)) return rc;
        const int walk_blocks = (int)chunks;   // one workgroup per chunk, one wave per band
        {
            TIMED(GSR_K_QCOUNT, stream);
            hipLaunchKernelGGL(gsr::k_qcount, dim3(walk_blocks), dim3(256), count_lds, stream, qa);
            KERNEL_CHECK("k_qcount", stream, dbg);
        }
        {
            TIMED(GSR_K_QSCAN, stream);
            hipLaunchKernelGGL(gsr::k_qscan, dim3(((Q + 3) / 4 + 31) / 32), dim3(1024), 0, stream, Q, chunks, (const uint8_t*)(b + bl.qhist),
                               (uint32_t*)(b + bl.qprefix), qcount);
            hipLaunchKernelGGL(gsr::k_qscan_glob, dim3(1), dim3(1024), 0, stream, tiles, (const uint32_t*)qcount, qstart, tile_order, hdr,
                               slot_dev, seq, post_cap);
            KERNEL_CHECK("k_qscan", stream, dbg);
        }
        {
            TIMED(GSR_K_QSCATTER, stream);
            hipLaunchKernelGGL(gsr::k_qscatter, dim3(walk_blocks), dim3(64), walk_lds, stream, qa);
            KERNEL_CHECK("k_qscatter", stream, dbg);
        }
    } else if (rankp) {
        // ---- rank path: depth-rank the splats once, order every tile's instances through an LDS bitmap (gsr_rank.hip) ----
        const uint32_t nb = (uint32_t)bl.nb;
        const int bin_blocks = pblocks < GSR_RANK_BLOCKS ? pblocks : GSR_RANK_BLOCKS;
        const bool direct = gsr::rank_direct(gx, tiles);
        uint32_t* const chunk_bounds = reinterpret_cast<uint32_t*>((uint4*)(b + bl.pstat) + 5 * (size_t)pblocks);   // behind pstat's rows and the four planes of group sums
        const size_t hist_bytes = direct ? 0 : (size_t)tiles * sizeof(uint32_t);
        const size_t count_lds = (size_t)nb * 4 + (direct ? 0 : (size_t)(gx + 1) * (size_t)(gy + 1) * sizeof(uint32_t));   // corner grid of the tile rects
        if (count_lds > 48 * 1024) {
            if (int rc = ensure_dynamic_lds((const void*)gsr::k_rcount, (size_t)count_lds)) return rc;
        }
        // the scatter's dynamic LDS: the tile histogram, then (when they fit the 160 KB beside it and the sort's 16 KB of keys) the staging
        // rows of the balanced expansion -- GSR_RSCATTER_BALANCED=0 keeps the lockstep form (A/B runs)
        const bool one_band_lds = (int)bl.nbands <= 1;
        const size_t stage_bytes = GSR_RSCATTER_STAGE_BYTES(one_band_lds), hist_al = (hist_bytes + 15) & ~(size_t)15;
        static const bool balanced_on = [] { const char* e = getenv("GSR_RSCATTER_BALANCED"); return !(e && e[0] == '0'); }();
        const bool balanced = balanced_on && hist_al + stage_bytes + 20 * 1024 <= 160 * 1024;
        const int stage_off = balanced ? (int)hist_al : -1;
        const size_t scatter_lds = balanced ? hist_al + stage_bytes : hist_bytes;
        if (scatter_lds > 48 * 1024)
        {
            if (int rc = ensure_dynamic_lds((const void*)gsr::k_rsort_rscatter, scatter_lds)) return rc;
            if (int rc = ensure_dynamic_lds((const void*)gsr::k_rscatter<8>, scatter_lds)) return rc;
        }
        uint32_t* block_hist = (uint32_t*)(b + bl.block_hist);
        uint32_t* bcount = (uint32_t*)(b + bl.bcount);
        uint32_t* bstart = (uint32_t*)(b + bl.bstart);
        uint32_t* bcursor = (uint32_t*)(b + bl.bcursor);
        uint32_t* bhist = (uint32_t*)(b + bl.bhist);
        unsigned long long* dkeys = (unsigned long long*)(b + bl.dkeys);
        unsigned long long* dtmp = (unsigned long long*)(b + bl.dtmp);
        uint32_t* rank = (uint32_t*)(b + bl.rank);
        uint2* ranks = (uint2*)(b + bl.ranks);
        const int nbands = (int)bl.nbands;
        const float inv_band_rows = 1.0f / (float)bl.band_rows;
        uint2* obs = nbands > 1 ? (uint2*)(b + bl.obs) : nullptr;
        uint32_t* bandcnt = (uint32_t*)(b + bl.bandcnt);
        const uint32_t nwc = (uint32_t)(((size_t)P + GSR_RANK_BAND_CHUNK - 1) / GSR_RANK_BAND_CHUNK);
        gsr::BandTables bt;
        bt.nbands = (uint32_t)nbands;
        bt.inv_band_rows = inv_band_rows;
        bt.over = (const uint32_t*)(b + bl.rank_over);
        ushort4* srect = (ushort4*)(b + bl.srect);
        uint32_t* tile_start = (uint32_t*)(b + bl.tile_start);
        uint32_t* tile_cursor = (uint32_t*)(b + bl.tile_cursor);
        uint2* ranges = (uint2*)(b + bl.ranges);
        if (pblocks > 0) {
            TIMED(GSR_K_COUNT, stream);
            hipLaunchKernelGGL(gsr::k_rcount, dim3(bin_blocks), dim3(GSR_RANK_BIN_THREADS), count_lds, stream, rank_ilv, P, gx, tiles,
                               pblocks, nb, (const ushort4*)srect, (const uint32_t*)pa.tiles_touched,
                               (const float*)pa.depths, (const uint4*)pa.pstat, tile_count, rect_total, block_hist, bcount, bhist, hdr, chunk_bounds);
            KERNEL_CHECK("k_rcount", stream, dbg);
        }
        const bool one_band = nbands <= 1;
        {
            TIMED(GSR_K_DEPTH_SORT, stream);
            // the depth keys into their buckets; the last workgroup scans the tile counters and posts the instance count (launched even when P == 0)
            gsr::TileScanArgs ts;
            ts.tiles = tiles; ts.tile_count = tile_count; ts.tile_start = tile_start; ts.tile_cursor = tile_cursor; ts.ranges = ranges;
            ts.tile_order = tile_order; ts.tdesc = (uint4*)(b + bl.tdesc); ts.total_dev = total_dev; ts.mailbox = slot_dev; ts.seq = seq;
            ts.post_capacity = post_cap;
            ts.hdr = pblocks > 0 ? hdr : nullptr;   // (P == 0: k_rcount did not run, the header word is not written)
            hipLaunchKernelGGL(gsr::k_rdscatter, dim3((pblocks > 0 ? bin_blocks : 0) + 1), dim3(GSR_RANK_BIN_THREADS), (size_t)nb * 4, stream, rank_ilv, P, nb,
                               (const ushort4*)srect, (const float*)pa.depths, hdr, (const uint32_t*)bcount, bstart, bcursor, dkeys, (const uint32_t*)bhist, ts,
                               (const uint32_t*)chunk_bounds);
            KERNEL_CHECK("k_rdscatter", stream, dbg);
            if (!one_band && pblocks > 0) {
                // large frames: a rank per (splat, band of tile rows) -- see gsr_rank.hip
                hipLaunchKernelGGL(gsr::k_rdsort, dim3(nb), dim3(256), 0, stream, (const uint32_t*)bcount, (const uint32_t*)bstart, dkeys, dtmp, rank, obs,
                                   (const ushort4*)srect, (int)bl.band_rows);
                KERNEL_CHECK("k_rdsort", stream, dbg);
                const uint32_t wgs = (nwc + 3u) / 4u;
                hipLaunchKernelGGL(gsr::k_band_count, dim3(wgs), dim3(256), 0, stream, (const gsr::BinHeader*)hdr, (const uint2*)obs,
                                   (uint32_t)nbands, nwc, bandcnt);
                hipLaunchKernelGGL(gsr::k_band_scan, dim3(nbands), dim3(1024), 0, stream, hdr, nwc, bandcnt);
                hipLaunchKernelGGL(gsr::k_band_rank, dim3(wgs), dim3(256), 0, stream, (const gsr::BinHeader*)hdr, (const uint2*)obs,
                                   (uint32_t)nbands, nwc, (const uint32_t*)bandcnt, (uint4*)rank, (uint32_t*)(b + bl.rank_over));
                KERNEL_CHECK("k_band_rank", stream, dbg);
            }
        }
        if (pblocks > 0) {
            TIMED(GSR_K_SCATTER, stream);
            if (one_band) {   // the depth sort beside the scatter, one launch (k_rsort_rscatter): 4-byte entries, k_tile_rank gathers the ranks
                hipLaunchKernelGGL(gsr::k_rsort_rscatter, dim3(bin_blocks + nb), dim3(GSR_RANK_BIN_THREADS), scatter_lds, stream, rank_ilv, bin_blocks, P, gx, tiles,
                                   (const ushort4*)srect, (const float4*)pa.sspan, (const uint32_t*)tile_start, tile_cursor, (uint32_t*)ranks, cap,
                                   (const unsigned long long*)total_dev, (const uint32_t*)block_hist, (const uint32_t*)bcount, (const uint32_t*)bstart,
                                   dkeys, dtmp, rank, stage_off, (const uint32_t*)chunk_bounds);
                KERNEL_CHECK("k_rsort_rscatter", stream, dbg);
            } else {
                hipLaunchKernelGGL(gsr::k_rscatter<8>, dim3(bin_blocks), dim3(GSR_RANK_BIN_THREADS), scatter_lds, stream, rank_ilv, P, gx, tiles, bt,
                                   (const ushort4*)srect, (const uint32_t*)rank, (const float4*)pa.sspan, (const uint32_t*)tile_start, tile_cursor, ranks, cap,
                                   (const unsigned long long*)total_dev, (const uint32_t*)block_hist, stage_off, (const uint32_t*)chunk_bounds);
                KERNEL_CHECK("k_rscatter", stream, dbg);
            }
        }
        {
            TIMED(GSR_K_TILE_SORT, stream);
            // the bitmap holds a rank space (the frame's, or a band's) in one pass unless it exceeds GSR_RANK_MAX_SPLATS splats (then: several)
            const size_t space = (size_t)P < (size_t)GSR_RANK_MAX_SPLATS ? (size_t)P : (size_t)GSR_RANK_MAX_SPLATS;
            const uint32_t words = (uint32_t)((space + 2047) / 2048) * 64u;
            hipLaunchKernelGGL(gsr::k_tile_rank, dim3(tiles), dim3(GSR_RANK_TILE_THREADS), (size_t)words * 6, stream, words, gx, nbands,
                               inv_band_rows, (const uint4*)(b + bl.tdesc), (const uint2*)ranks, one_band ? (const uint32_t*)rank : nullptr, (const float*)pa.depths,
                               (const gsr::BinHeader*)hdr, write_lists ? (unsigned long long*)(b + bl.keys) : nullptr,
                               (uint32_t*)(b + bl.point_list), write_lists ? qlist : nullptr, qpos, qcount, qstart,
                               cap, (const unsigned long long*)total_dev);
            KERNEL_CHECK("k_tile_rank", stream, dbg);
        }
    } else {
    // round 1's per-tile bitonic sort (tile_culling 5).  Binning workgroups: each owns a contiguous chunk of splats and an LDS histogram over the tiles
    const int bin_blocks = pblocks < GSR_BIN_BLOCKS ? pblocks : GSR_BIN_BLOCKS;
    uint32_t* block_hist = (uint32_t*)(b + bl.block_hist);
    // (grids beyond GSR_LDS_HIST_TILES tiles -- past ~3200x3200 px -- fall back to per-instance L2 atomics)
    const size_t hist_bytes = tiles > GSR_LDS_HIST_TILES ? 0 : (size_t)tiles * sizeof(uint32_t);
    if (hist_bytes > 48 * 1024) {
        if (int rc = ensure_dynamic_lds((const void*)gsr::k_count<false>, (size_t)hist_bytes)) return rc;
        if (int rc = ensure_dynamic_lds((const void*)gsr::k_count<true>, (size_t)hist_bytes)) return rc;
        if (int rc = ensure_dynamic_lds((const void*)gsr::k_scatter<false>, (size_t)hist_bytes)) return rc;
        if (int rc = ensure_dynamic_lds((const void*)gsr::k_scatter<true>, (size_t)hist_bytes)) return rc;
    }
    const bool cull = settings->tile_culling != 0;
    if (pblocks > 0) {
        TIMED(GSR_K_COUNT, stream);
        hipLaunchKernelGGL(cull ? gsr::k_count<true> : gsr::k_count<false>, dim3(bin_blocks), dim3(256), hist_bytes, stream, P, gx, tiles,
                           (const ushort4*)pa.rect, (const uint32_t*)pa.tiles_touched, (const float4*)pa.grec, tile_count, rect_total, block_hist);
        KERNEL_CHECK("k_count", stream, dbg);
    }
    uint32_t* tile_start = (uint32_t*)(b + bl.tile_start);
    uint32_t* tile_cursor = (uint32_t*)(b + bl.tile_cursor);
    uint2* ranges = (uint2*)(b + bl.ranges);
    {
        TIMED(GSR_K_TILE_SCAN, stream);
        hipLaunchKernelGGL(gsr::k_tile_scan, dim3(1), dim3(1024), 0, stream, tiles, (const uint32_t*)tile_count, tile_start, tile_cursor,
                           ranges, tile_order, total_dev, slot_dev, seq, post_cap);
        KERNEL_CHECK("k_tile_scan", stream, dbg);
    }

    // Optimistic launch: the rest of the frame is enqueued against the caller's capacity before the
    // host knows I; every kernel re-checks *total_dev <= capacity on the device and does nothing
    // otherwise.  The host then waits only for the scan (early in the frame), not for the frame.
    unsigned long long* keys = (unsigned long long*)(b + bl.keys);
    uint32_t* point_list = (uint32_t*)(b + bl.point_list);
    if (pblocks > 0) {
        TIMED(GSR_K_SCATTER, stream);
        hipLaunchKernelGGL(cull ? gsr::k_scatter<true> : gsr::k_scatter<false>, dim3(bin_blocks), dim3(256), hist_bytes, stream, P, gx, tiles,
                           (const float*)pa.depths, (const ushort4*)pa.rect, (const uint32_t*)pa.tiles_touched, (const float4*)pa.grec,
                           (const uint32_t*)tile_start, tile_cursor, keys, cap, (const unsigned long long*)total_dev,
                           (const uint32_t*)block_hist);
        KERNEL_CHECK("k_scatter", stream, dbg);
    }
    {
        // per-tile sort + quadrant streams: three size classes, the long (few) ones first.  Launched even when
        // P == 0 so that every tile's quadrant counters are written.
        TIMED(GSR_K_TILE_SORT, stream);
        auto sort_class = [&](auto kernel, int threads, uint32_t n_lo, uint32_t n_hi) {
            hipLaunchKernelGGL(kernel, dim3(tiles), dim3(threads), 0, stream, n_lo, n_hi, gx, (const uint32_t*)tile_order,
                               (const uint32_t*)tile_count, (const uint32_t*)tile_start, keys, write_lists ? point_list : nullptr, write_lists ? qlist : nullptr, qpos, qcount, qstart,
                               (const float4*)pa.grec, cap,
                               (const unsigned long long*)total_dev);
        };
        sort_class(gsr::k_tile_sort<GSR_SORT_XL_KEYS, 1024>, 1024, (uint32_t)GSR_SORT_LDS_KEYS, 0xFFFFFFFFu);
        sort_class(gsr::k_tile_sort<GSR_SORT_LDS_KEYS, 1024>, 1024, (uint32_t)GSR_SORT_SMALL_KEYS, (uint32_t)GSR_SORT_LDS_KEYS);
        sort_class(gsr::k_tile_sort<GSR_SORT_SMALL_KEYS, 256>, 256, 0u, (uint32_t)GSR_SORT_SMALL_KEYS);
        KERNEL_CHECK("k_tile_sort", stream, dbg);
    }
    }   // per-tile sort path
    {
        TIMED(GSR_K_RENDER, stream);
        // fast blend with continuations (s.cont_chunks > 0): quadrants whose walk reaches entry cont_chunks * GSR_BWD_SEGMENT with pixels still open are
        // finished four chunks at a time by continuation workgroups -- at the end of the same grid, waiting for them (mode 1), or as a kernel of their own (mode 2)
        static const int cont_grid = [] { const char* e = getenv("GSR_CONT_GRID"); const int v = e ? atoi(e) : 0; return v > 0 ? v : GSR_CONT_GRID_DEFAULT; }();
        const int cgrid = 4 * tiles < cont_grid ? 4 * tiles : cont_grid;
        const bool conts = ds.fast_blend && ds.cont_chunks > 0;
        if (!conts) ds.cont_chunks = 0;
        const bool infer = ds.forward_only && !conts;   // (torch.no_grad() / fps benchmarks: the instances without the backward's bookkeeping)
        auto* const render_k = !ds.fast_blend ? (infer ? &gsr::k_render<false, 0, true> : &gsr::k_render<false, 0, false>)
                               : (conts && ds.cont_mode == 1 ? &gsr::k_render<true, 1, false> : (infer ? &gsr::k_render<true, 0, true> : &gsr::k_render<true, 0, false>));
#ifndef GSR_EXP_NO_SOLO
        const bool solo = !(conts && ds.cont_mode == 1);   // (k_render<., 0>: one wave per workgroup, four workgroups per tile -- gsr_forward.hip)
#else
        const bool solo = false;
#endif
        hipLaunchKernelGGL(render_k, solo ? dim3(4 * tiles) : dim3(tiles + cgrid), dim3(solo ? 64 : 256), 0, stream, ds, (const uint32_t*)tile_order, (const uint32_t*)qstart,
                           (const uint32_t*)qcount, (const float4*)pa.grec, (const uint32_t*)qpos, write_lists ? (const uint32_t*)qlist : nullptr, (float*)(im + il.final_T), (uint32_t*)(im + il.n_contrib),
                           (uint32_t*)(im + il.n_contrib_q), (float*)(im + il.c_final), (float4*)(im + il.ck), out_color, cap,
                           (const unsigned long long*)total_dev, (uint32_t*)(im + il.units), tiles);
        KERNEL_CHECK("k_render", stream, dbg);
        if (conts && ds.cont_mode == 2) {
            hipLaunchKernelGGL((gsr::k_render<true, 2, false>), dim3(cgrid), dim3(64 * GSR_CONT_WAVES), 0, stream, ds, (const uint32_t*)tile_order, (const uint32_t*)qstart,
                               (const uint32_t*)qcount, (const float4*)pa.grec, (const uint32_t*)qpos, write_lists ? (const uint32_t*)qlist : nullptr, (float*)(im + il.final_T), (uint32_t*)(im + il.n_contrib),
                               (uint32_t*)(im + il.n_contrib_q), (float*)(im + il.c_final), (float4*)(im + il.ck), out_color, cap,
                               (const unsigned long long*)total_dev, (uint32_t*)(im + il.units), tiles);
            KERNEL_CHECK("k_render_cont", stream, dbg);
        }
    }

    if (deferred) {   // the count of this frame is not known yet: the caller checks its slot after the fact
        *num_rendered_host = -1;
        return GSR_OK;
    }
    // wait for the scan's post
    const auto t0 = std::chrono::steady_clock::now();
    unsigned long long v;
    unsigned spins = 0;
    for (;;) {
        v = __atomic_load_n((unsigned long long*)slot, __ATOMIC_ACQUIRE);
        if (v != 0 && (v >> 40) == seq) break;
        if (++spins > 2000) {
            std::this_thread::yield();
            if ((spins & 0xFFF) == 0) {
                if (hipStreamQuery(stream) != hipErrorNotReady) {
                    // stream drained (or faulted) without a post
                    v = __atomic_load_n((unsigned long long*)slot, __ATOMIC_ACQUIRE);
                    if (v != 0 && (v >> 40) == seq) break;
                    hipError_t e = hipStreamSynchronize(stream);
                    return fail(GSR_E_HIP, "stream finished without posting the instance count: %s", hipGetErrorString(e));
                }
                if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120))
                    return fail(GSR_E_TIMEOUT, "timed out waiting for the instance count");
            }
        }
    }
    g_wait_ns.fetch_add((long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count());
    g_waits.fetch_add(1);
    const int64_t I = (int64_t)(v & kCountMask);
    note_posted(GSR_COUNT_SLOTS, v);
    *num_rendered_host = I;
    if (I > 0xFFFFFFFFll) return fail(GSR_E_ARG, "%lld (splat, tile) instances exceed the 32-bit offsets of the binning state", (long long)I);
    if (I > binning_capacity) return fail(GSR_E_CAPACITY, "binning capacity %lld < %lld instances", (long long)binning_capacity, (long long)I);
    return GSR_OK;
}

int gsr_forward_ex(const GsrSettings* settings, int32_t P, int32_t M, const float* means3D, const float* shs, const float* shs_rest,
                   const float* colors_precomp, const float* opacities, const float* scales, const float* rotations,
                   const float* cov3D_precomp, float* out_color, int32_t* radii, void* geom, void* binning,
                   int64_t binning_capacity, void* img, int64_t* num_rendered_host, void* stream)
{
    return forward_impl(settings, P, M, means3D, shs, shs_rest, colors_precomp, opacities, scales, rotations, cov3D_precomp, out_color, radii,
                        geom, binning, binning_capacity, img, num_rendered_host, stream, nullptr);
}

int gsr_forward_bound(const GsrSettings* settings, int32_t P, int32_t M, const GsrBound* bound, const float* xyz_local, const float* shs,
                      const float* shs_rest, const float* opacity_logit, const float* log_scales, const float* rot_local,
                      float* out_color, int32_t* radii, void* geom, void* binning, int64_t binning_capacity, void* img,
                      int64_t* num_rendered_host, void* stream)
{
    if (!bound) return fail(GSR_E_ARG, "gsr_forward_bound: bound is NULL");
    return forward_impl(settings, P, M, xyz_local, shs, shs_rest, nullptr, opacity_logit, log_scales, rot_local, nullptr, out_color, radii,
                        geom, binning, binning_capacity, img, num_rendered_host, stream, bound);
}

int gsr_forward(const GsrSettings* settings, int32_t P, int32_t M, const float* means3D, const float* shs,
                const float* colors_precomp, const float* opacities, const float* scales, const float* rotations,
                const float* cov3D_precomp, float* out_color, int32_t* radii, void* geom, void* binning,
                int64_t binning_capacity, void* img, int64_t* num_rendered_host, void* stream)
{
    return gsr_forward_ex(settings, P, M, means3D, shs, nullptr, colors_precomp, opacities, scales, rotations, cov3D_precomp, out_color,
                          radii, geom, binning, binning_capacity, img, num_rendered_host, stream);
}

static int backward_impl(const GsrSettings* settings, int32_t P, int32_t M, const float* means3D, const float* shs, const float* shs_rest,
                         const float* colors_precomp, const float* scales, const float* rotations, const float* cov3D_precomp,
                         const int32_t* radii, void* geom, const void* binning, int64_t binning_capacity, const void* img,
                         int64_t num_rendered, const float* dL_dpix, float* dL_dmeans3D, float* dL_dmeans2D,
                         float* dL_dsh, float* dL_dsh_rest, float* dL_dcolors, float* dL_dopacity, float* dL_dscales, float* dL_drotations,
                         float* dL_dcov3D, void* stream_, const GsrBound* bound, const float* opacity_logit)
{
    if (int rc = check_settings(settings)) return rc;
    hipStream_t stream = (hipStream_t)stream_;
    if (P < 0 || num_rendered < 0 || binning_capacity < num_rendered) return fail(GSR_E_ARG, "bad sizes");
    if (settings->forward_only) return fail(GSR_E_ARG, "this state comes from a forward_only forward: its backward accumulators were never zeroed");
    if (P == 0) return GSR_OK;
    if (!means3D || !radii || !geom || !binning || !img || !dL_dpix || !dL_dmeans3D || !dL_dmeans2D ||
        !dL_dcolors || !dL_dopacity || !dL_dcov3D)
        return fail(GSR_E_ARG, "NULL buffer");
    const bool pre_col = colors_precomp != nullptr, pre_cov = cov3D_precomp != nullptr;
    if (!pre_col && (!shs || !dL_dsh)) return fail(GSR_E_ARG, "shs / dL_dsh required when colours come from SH");
    if (shs_rest && (pre_col || !dL_dsh_rest || M < 2 || M > 16)) return fail(GSR_E_ARG, "split SH backward needs dL_dsh (P,1,3) + dL_dsh_rest (P,M-1,3), 2 <= M <= 16");
    if (!pre_cov && (!scales || !rotations || !dL_dscales || !dL_drotations))
        return fail(GSR_E_ARG, "scales/rotations and their gradients required when cov3D is not precomputed");

    const int W = settings->image_width, H = settings->image_height;
    const int gx = (W + GSR_BLOCK_X - 1) / GSR_BLOCK_X, gy = (H + GSR_BLOCK_Y - 1) / GSR_BLOCK_Y;
    const bool dbg = settings->debug != 0;
    GsrGeomLayout gl;
    GsrBinningLayout bl;
    GsrImageLayout il;
    gsr_geom_layout(P, &gl);
    gsr_binning_layout(binning_capacity, W, H, P, settings->tile_culling, &bl);
    gsr_image_layout(W, H, &il);
    char* g = (char*)geom;
    const char* b = (const char*)binning;
    const char* im = (const char*)img;
    gsr::Settings ds = to_dev_settings(settings);
    ds.fast_blend = fast_effective(settings, bl) ? 1 : 0;
    float* grad_scratch = (float*)(g + gl.acc);   // zeroed by the forward (and by the previous backward)
    long long* acc64 = (long long*)(g + gl.acc64);
    uint32_t* gmax = (uint32_t*)(const_cast<char*>(im) + il.gmax);   // scratch word of the image state
    const bool det = settings->deterministic != 0;
    if (det) {   // the fixed-point scale of this backward: max |dL/dpixel|
        HIP_TRY(hipMemsetAsync(gmax, 0, 4, stream));
        hipLaunchKernelGGL(gsr::k_gmax, dim3(256), dim3(256), 0, stream, (size_t)3 * W * H, dL_dpix, gmax);
        KERNEL_CHECK("k_gmax", stream, dbg);
    }
    if (num_rendered > 0) {
        TIMED(GSR_K_RENDER_BWD, stream);
        if (ds.fast_blend && settings->fast_blend != 2) {   // fast blend: the record-parallel kernel (fast_blend == 2 keeps the pixel-parallel walk: A/B runs)
            // a fixed grid: every wave takes the units of its list in turn (gsr.h: GsrImageLayout.units), so the size only sets how many
            // share the work -- about one unit per wave at 100 k splats and 802 x 550, never more workgroups than the old one per (tile, segment)
#ifndef GSR_EXP_RP_GRID
#define GSR_EXP_RP_GRID 3072
#endif
#ifndef GSR_RP_WAVES_PER_WG_DEFAULT
#define GSR_RP_WAVES_PER_WG_DEFAULT 1
#endif
            int rp_grid = gx * gy * GSR_BWD_SEGMENTS;
            rp_grid = rp_grid < GSR_EXP_RP_GRID ? (rp_grid + 15) / 16 * 16 : GSR_EXP_RP_GRID;   // a multiple of GSR_UNIT_LISTS / 4 workgroups: every list sees the same stride
            // ONE wave per workgroup (round 6): a wave gets about one unit and units differ in cost by the half-rows that reach them (1 .. 16); in a
            // workgroup of four the slot of a finished wave stays taken until the slowest of the four is done
            static const int rp_waves = [] { const char* e = getenv("GSR_RP_WAVES_PER_WG"); const int v = e ? atoi(e) : 0; return v == 1 || v == 2 || v == 4 ? v : GSR_RP_WAVES_PER_WG_DEFAULT; }();
            auto* const rp_k = rp_waves == 1 ? &gsr::k_render_bwd_rp<1> : (rp_waves == 2 ? &gsr::k_render_bwd_rp<2> : &gsr::k_render_bwd_rp<4>);
            hipLaunchKernelGGL(rp_k, dim3(rp_grid * (4 / rp_waves)), dim3(64 * rp_waves), 0, stream, ds,
                               (const uint32_t*)(b + bl.qstart), (const uint32_t*)(b + bl.qcount), (const float4*)(g + gl.grec), (const uint32_t*)(b + bl.qpos),
                               (const float*)(im + il.final_T), (const uint32_t*)(im + il.n_contrib_q), dL_dpix, grad_scratch,
                               (const float*)(im + il.c_final), (const float4*)(im + il.ck), gx * gy,
                               (unsigned long long)binning_capacity, (const unsigned long long*)b, (const uint32_t*)(im + il.units));
        } else {
        auto* const render_bwd = det ? &gsr::k_render_bwd<true, false> : (ds.fast_blend ? &gsr::k_render_bwd<false, true> : &gsr::k_render_bwd<false, false>);
        hipLaunchKernelGGL(render_bwd, dim3(gx * gy * GSR_BWD_SEGMENTS), dim3(256), 0, stream, ds, (const uint32_t*)(b + bl.tile_order),
                           (const uint32_t*)(b + bl.qstart), (const uint32_t*)(b + bl.qcount), (const float4*)(g + gl.grec), (const uint32_t*)(b + bl.qpos),
                           (const float*)(im + il.final_T), (const uint32_t*)(im + il.n_contrib_q), dL_dpix, grad_scratch,
                           (const float*)(im + il.c_final), (const float4*)(im + il.ck), gx * gy, acc64, (const uint32_t*)gmax,
                           (unsigned long long)binning_capacity, (const unsigned long long*)b);
        }
        KERNEL_CHECK("k_render_bwd", stream, dbg);
    }
    gsr::PreBwdArgs pa;
    pa.P = P; pa.M = M;
    pa.means3D = means3D; pa.shs = shs; pa.shs_rest = shs_rest; pa.scales = scales; pa.rotations = rotations;
    pa.cov3D = (const float*)(g + gl.cov3D);
    pa.radii = radii;
    pa.clamped = (const uint8_t*)(g + gl.clamped);
    pa.acc = grad_scratch;
    pa.acc64 = acc64;
    pa.gmax = gmax;
    pa.grec = (const float4*)(g + gl.grec);
    pa.use_precomp_cov = pre_cov ? 1 : 0;
    pa.use_precomp_color = pre_col ? 1 : 0;
    pa.dL_dmeans3D = dL_dmeans3D; pa.dL_dmeans2D = dL_dmeans2D; pa.dL_dsh = pre_col ? nullptr : dL_dsh; pa.dL_dsh_rest = shs_rest ? dL_dsh_rest : nullptr;
    // leaves entries (bound != nullptr): colours come from SH and the covariance from scale / rotation there, nobody reads these two
    pa.dL_dcolors = bound ? nullptr : dL_dcolors; pa.dL_dopacity = dL_dopacity;
    pa.dL_dscales = pre_cov ? nullptr : dL_dscales; pa.dL_drotations = pre_cov ? nullptr : dL_drotations;
    pa.dL_dcov3D = bound ? nullptr : dL_dcov3D;
    if (int rc = to_bound(bound, P, true, &pa.bound)) return rc;
    pa.opacities = opacity_logit;
    if (bound && !opacity_logit) return fail(GSR_E_ARG, "gsr_backward_bound: opacity_logit is NULL");
    {
        TIMED(GSR_K_PREPROCESS_BWD, stream);
        hipLaunchKernelGGL(gsr::k_preprocess_bwd, dim3((P + GSR_PREBWD_ROWS - 1) / GSR_PREBWD_ROWS), dim3(GSR_PREBWD_ROWS), 0, stream, ds, pa);
        KERNEL_CHECK("k_preprocess_bwd", stream, dbg);
    }
    return GSR_OK;
}

int gsr_backward_ex(const GsrSettings* settings, int32_t P, int32_t M, const float* means3D, const float* shs, const float* shs_rest,
                    const float* colors_precomp, const float* scales, const float* rotations, const float* cov3D_precomp,
                    const int32_t* radii, void* geom, const void* binning, int64_t binning_capacity, const void* img,
                    int64_t num_rendered, const float* dL_dpix, float* dL_dmeans3D, float* dL_dmeans2D,
                    float* dL_dsh, float* dL_dsh_rest, float* dL_dcolors, float* dL_dopacity, float* dL_dscales, float* dL_drotations,
                    float* dL_dcov3D, void* stream)
{
    return backward_impl(settings, P, M, means3D, shs, shs_rest, colors_precomp, scales, rotations, cov3D_precomp, radii, geom, binning,
                         binning_capacity, img, num_rendered, dL_dpix, dL_dmeans3D, dL_dmeans2D, dL_dsh, dL_dsh_rest, dL_dcolors, dL_dopacity,
                         dL_dscales, dL_drotations, dL_dcov3D, stream, nullptr, nullptr);
}

int gsr_backward_bound(const GsrSettings* settings, int32_t P, int32_t M, const GsrBound* bound, const float* xyz_local, const float* shs,
                       const float* shs_rest, const float* opacity_logit, const float* log_scales, const float* rot_local,
                       const int32_t* radii, void* geom, const void* binning, int64_t binning_capacity, const void* img,
                       int64_t num_rendered, const float* dL_dpix, float* dL_dxyz_local, float* dL_dmeans2D, float* dL_dsh,
                       float* dL_dsh_rest, float* dL_dopacity_logit, float* dL_dlog_scales, float* dL_drot_local, float* scratch9,
                       void* stream)
{
    if (!bound) return fail(GSR_E_ARG, "gsr_backward_bound: bound is NULL");
    if (P > 0 && !scratch9) return fail(GSR_E_ARG, "gsr_backward_bound: scratch9 (9 P floats) is NULL");
    return backward_impl(settings, P, M, xyz_local, shs, shs_rest, nullptr, log_scales, rot_local, nullptr, radii, geom, binning,
                         binning_capacity, img, num_rendered, dL_dpix, dL_dxyz_local, dL_dmeans2D, dL_dsh, dL_dsh_rest, scratch9 /*dL_dcolors*/,
                         dL_dopacity_logit, dL_dlog_scales, dL_drot_local, scratch9 + 3 * (size_t)P /*dL_dcov3D*/, stream, bound, opacity_logit);
}

int gsr_backward(const GsrSettings* settings, int32_t P, int32_t M, const float* means3D, const float* shs,
                 const float* colors_precomp, const float* scales, const float* rotations, const float* cov3D_precomp,
                 const int32_t* radii, void* geom, const void* binning, int64_t binning_capacity, const void* img,
                 int64_t num_rendered, const float* dL_dpix, float* dL_dmeans3D, float* dL_dmeans2D,
                 float* dL_dsh, float* dL_dcolors, float* dL_dopacity, float* dL_dscales, float* dL_drotations,
                 float* dL_dcov3D, void* stream)
{
    return gsr_backward_ex(settings, P, M, means3D, shs, nullptr, colors_precomp, scales, rotations, cov3D_precomp, radii, geom, binning,
                           binning_capacity, img, num_rendered, dL_dpix, dL_dmeans3D, dL_dmeans2D, dL_dsh, nullptr,
                           dL_dcolors, dL_dopacity, dL_dscales, dL_drotations, dL_dcov3D, stream);
}

int gsr_profile_enable(int on)
{
    g_prof.on.store(on ? 1 : 0);
    return GSR_OK;
}

int gsr_profile_read(double* total_ms, int64_t* launches)
{
    if (!total_ms || !launches) return fail(GSR_E_ARG, "gsr_profile_read: NULL output");
    std::lock_guard<std::mutex> lk(g_prof.mu);
    for (auto& p : g_prof.pending) {
        HIP_TRY(hipEventSynchronize(p.b));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, p.a, p.b));
        if (p.id >= 0 && p.id < GSR_NUM_KERNELS) { total_ms[p.id] += (double)ms; launches[p.id] += 1; }
        g_prof.pool.push_back(p.a);
        g_prof.pool.push_back(p.b);
    }
    g_prof.pending.clear();
    return GSR_OK;
}

int gsr_wait_stats(double* total_wait_ms, int64_t* waits)
{
    if (!total_wait_ms || !waits) return fail(GSR_E_ARG, "gsr_wait_stats: NULL output");
    *total_wait_ms = (double)g_wait_ns.exchange(0) * 1e-6;
    *waits = (int64_t)g_waits.exchange(0);
    return GSR_OK;
}

const char* gsr_kernel_name(int id)
{
    static const char* names[GSR_NUM_KERNELS] = {"k_preprocess", "k_tile_scan", "k_scatter", "k_tile_sort",
                                                 "k_render", "k_render_bwd", "k_preprocess_bwd", "k_count",
                                                 "k_depth_sort", "k_qcount", "k_qscan", "k_qscatter"};
    return (id >= 0 && id < GSR_NUM_KERNELS) ? names[id] : "?";
}

int gsr_mark_visible(int32_t P, const float* means3D, const float* viewmatrix, const float* projmatrix, uint8_t* present, void* stream_)
{
    (void)projmatrix;
    if (P < 0 || (P > 0 && (!means3D || !viewmatrix || !present))) return fail(GSR_E_ARG, "gsr_mark_visible: bad arguments");
    if (P == 0) return GSR_OK;
    hipStream_t stream = (hipStream_t)stream_;
    hipLaunchKernelGGL(gsr::k_mark_visible, dim3((P + 255) / 256), dim3(256), 0, stream, P, means3D, viewmatrix, present);
    KERNEL_CHECK("k_mark_visible", stream, false);
    return GSR_OK;
}

}  // extern "C"
