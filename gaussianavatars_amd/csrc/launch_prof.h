// launch_prof.h -- optional per-kernel timing of a library's launches with hipEvents on the launch stream (libgab / libgls; libgsr has its
// own slot-based twin in gsr_api.hip).  Off by default: a launch then costs one relaxed atomic load.  When on, every launch made through
// PROF_LAUNCH is bracketed by an event pair; <lib>_profile_collect() synchronises the pending pairs and adds their elapsed times to a
// small table keyed by the kernel's name (the text of the launch expression), which bench.py reads for roofline.all_kernels / roofline.step.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstring>
#include <mutex>
#include <vector>

namespace lprof {

struct Pending { const char* name; hipEvent_t a, b; };
struct Entry { const char* name; double ms; long long launches; };
struct State {
    std::atomic<int> on{0};
    std::mutex mu;
    std::vector<Pending> pending;
    std::vector<hipEvent_t> pool;
    std::vector<Entry> table;
};
static State g;

static inline hipEvent_t take_event()
{
    if (!g.pool.empty()) { hipEvent_t e = g.pool.back(); g.pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
struct Scope {   // records the pair around one launch
    hipEvent_t a = nullptr, b = nullptr;
    const char* name;
    hipStream_t stream;
    Scope(const char* n, hipStream_t s) : name(n), stream(s)
    {
        if (!g.on.load(std::memory_order_relaxed)) return;
        std::lock_guard<std::mutex> lk(g.mu);
        a = take_event(), b = take_event();
        if (a) (void)hipEventRecord(a, stream);
    }
    ~Scope()
    {
        if (!a || !b) return;
        (void)hipEventRecord(b, stream);
        std::lock_guard<std::mutex> lk(g.mu);
        g.pending.push_back({name, a, b});
    }
};
static inline int collect()
{
    std::lock_guard<std::mutex> lk(g.mu);
    for (auto& p : g.pending) {
        float ms = 0.f;
        if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            Entry* e = nullptr;
            for (auto& t : g.table)
                if (t.name == p.name || std::strcmp(t.name, p.name) == 0) { e = &t; break; }
            if (!e) { g.table.push_back({p.name, 0.0, 0}); e = &g.table.back(); }
            e->ms += (double)ms;
            e->launches += 1;
        }
        g.pool.push_back(p.a);
        g.pool.push_back(p.b);
    }
    g.pending.clear();
    return (int)g.table.size();
}
static inline int entry(int i, const char** name, double* ms, long long* launches)
{
    std::lock_guard<std::mutex> lk(g.mu);
    if (i < 0 || i >= (int)g.table.size()) return -1;
    if (name) *name = g.table[i].name;
    if (ms) *ms = g.table[i].ms;
    if (launches) *launches = g.table[i].launches;
    return 0;
}
static inline void reset()
{
    std::lock_guard<std::mutex> lk(g.mu);
    g.table.clear();
}

}  // namespace lprof

// hipLaunchKernelGGL with the optional event pair around it; the table's key is the kernel expression as written (e.g. "gab::k_flame_fused<true>")
#define PROF_LAUNCH(kernel, grid, block, shmem, stream, ...)                          \
    do {                                                                              \
        lprof::Scope prof_scope_(#kernel, (hipStream_t)(stream));                     \
        hipLaunchKernelGGL(kernel, grid, block, shmem, (hipStream_t)(stream), __VA_ARGS__); \
    } while (0)
