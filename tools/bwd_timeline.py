"""Per-wave timeline of k_render_bwd (needs the -DGSR_EXPERIMENT_TIMELINE debug build copied over libgsr_hip.so)."""
import ctypes as C, sys, numpy as np, torch
sys.path.insert(0, '.')
import bench
from gaussianavatars_amd import _lib
dev = torch.device('cuda:0')
g, cam = bench.build_scene(dev, 100_000, 3, 550, 802, 4, "fused", True)
bg = torch.ones(3, device=dev); target = torch.ones(3, 802, 550, device=dev)
for i in range(12):
    bench.zero_grads(g); bench.one_step(g, cam, bg, target, 0, True)
torch.cuda.synchronize()
lib = _lib.gsr()
buf = (C.c_ulonglong * (4 * 16384))()
lib.gsr_debug_read.restype = C.c_int
assert lib.gsr_debug_read(buf, 4 * 16384) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 4)[:7140].astype(np.int64)
t0, t1, jm, hw = a[:, 0], a[:, 1], a[:, 2], a[:, 3]
live = t1 > 0
t0, t1, jm, hw = t0[live], t1[live], jm[live], hw[live]
if not live.any():      # fast blend (the default): the backward is k_render_bwd_rp, which carries no stamps
    print("no stamps from k_render_bwd (fast blend runs k_render_bwd_rp); GSR_FAST_BLEND=0 for the pixel-parallel walk")
    t0 = t1 = jm = hw = np.zeros(1, np.int64)
base = t0.min()
dur = (t1 - t0) / 100.0            # wall_clock64 ticks at 100 MHz -> us
print("waves", len(t0), "kernel span %.1f us" % ((t1.max() - base) / 100.0), "latest start %.1f us" % ((t0.max() - base) / 100.0))
print("wave duration us: mean %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f" % (dur.mean(), *np.percentile(dur, [50, 90, 99]), dur.max()))
order = np.argsort(-(t1 - base))[:10]
print("last finishers: (end us, start us, dur us, jmax, us per record)")
for w in order:
    print("   %.1f %.1f %.1f %d %.3f" % ((t1[w] - base) / 100.0, (t0[w] - base) / 100.0, dur[w], jm[w], dur[w] / max(jm[w], 1)))
deep = np.argsort(-jm)[:10]
print("deepest walks: (jmax, dur us, us per record, end us)")
for w in deep:
    print("   %d %.1f %.3f %.1f" % (jm[w], dur[w], dur[w] / max(jm[w], 1), (t1[w] - base) / 100.0))
sel = jm > 50
print("us per record vs depth: corr(dur, jmax) = %.3f;  median us/record (jmax>50) %.3f" % (np.corrcoef(dur, jm)[0, 1], np.median(dur[sel] / jm[sel])))

# ---- forward (k_render)
lib.gsr_debug_read_fwd.restype = C.c_int
assert lib.gsr_debug_read_fwd(buf, 4 * 16384) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 4)[:7140].astype(np.int64)
t0, t1, nj, tm = a[:, 0], a[:, 1], a[:, 2], a[:, 3]
live = t1 > 0
t0, t1, nj, tm = t0[live], t1[live], nj[live], tm[live]
n, jmain = nj >> 32, nj & 0xFFFFFFFF
base = t0.min(); dur = (t1 - t0) / 100.0; tail = (t1 - tm) / 100.0
print("FORWARD waves", len(t0), "kernel span %.1f us" % ((t1.max() - base) / 100.0), "latest start %.1f us" % ((t0.max() - base) / 100.0))
print("wave duration us: mean %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f;  sum walked (pixel-parallel) %d;  waves entering tail mode %d, tail time total %.0f us" %
      (dur.mean(), *np.percentile(dur, [50, 90, 99]), dur.max(), jmain.sum(), int((jmain < n).sum()), tail[jmain < n].sum()))
print("last finishers: (end us, dur us, stream n, walked pixel-parallel, tail us, us per walked record)")
for w in np.argsort(-(t1 - base))[:10]:
    print("   %.1f %.1f %d %d %.1f %.3f" % ((t1[w] - base) / 100.0, dur[w], n[w], jmain[w], tail[w], (dur[w] - tail[w]) / max(jmain[w], 1)))
# distribution of the pixel-parallel walk lengths and what a cap on the serial walk would leave
for cap in (600, 400, 300, 240, 180, 120):
    print("   walks longer than %d records: %d waves, %d records beyond the cap" % (cap, int((jmain > cap).sum()), int(np.maximum(jmain - cap, 0).sum())))
print("   walk length percentiles 50/90/99/99.9/max:", *np.percentile(jmain, [50, 90, 99, 99.9, 100]).astype(int))
busy = dur.sum()
span = (t1.max() - base) / 100.0
print("   sum of wave durations %.0f us = %.1f waves in flight on average over the %.1f us span (1024 SIMDs)" % (busy, busy / span, span))
# time per walked record by how many records the wave walks: lone deep waves vs crowded shallow ones
for lo, hi in ((0, 60), (60, 120), (120, 240), (240, 400), (400, 10000)):
    s = (jmain > lo) & (jmain <= hi)
    if s.any():
        print("   walks of %d..%d records: %d waves, median %.3f us per record, median end %.1f us" % (lo, hi, int(s.sum()), np.median((dur[s] - tail[s]) / jmain[s]), np.median((t1[s] - base) / 100.0)))
