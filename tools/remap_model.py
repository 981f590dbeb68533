"""CPU model of the blend kernels' issue cost under two lane mappings, from gpurun_out/stream_dump_<wl>.npz (tools/stream_dump.py):
lanes = pixels (the serial walk of today) against lanes = records (one pixel at a time over a chunk of the stream)."""
import sys
import numpy as np
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
d = np.load("gpurun_out/stream_dump_%s.npz" % wl)
nq, qcount, fT = d["n_contrib_q"].astype(np.int64), d["qcount"].astype(np.int64), d["final_T"]
H, W = nq.shape
gx, gy = (W + 15) // 16, (H + 15) // 16
pad = np.zeros((gy * 16, gx * 16), np.int64); pad[:H, :W] = nq
# quadrant q = 4*tile + (row&1)*2 + (col&1); pixels of a quadrant
q = pad.reshape(gy, 2, 8, gx, 2, 8).transpose(0, 3, 1, 4, 2, 5).reshape(gy * gx * 4, 64)
n = qcount.reshape(-1)
assert len(n) == len(q)
jmax = q.max(1)
print("quadrants", len(q), "stream entries", n.sum(), "bwd wave-records (sum jmax)", jmax.sum(), "useful pixel-records", q.sum(),
      "lane utilisation %.3f" % (q.sum() / (64.0 * jmax.sum())))
for C, L in ((60, 64), (64, 64), (30, 32), (32, 32)):
    bodies = np.ceil(q / C).sum()                     # (pixel, chunk) bodies
    chunks = np.ceil(jmax / C).sum()                  # wave-chunks (record loads, atomics)
    print(f"bwd lanes=records chunk {C} on {L} lanes: pixel-chunk bodies {int(bodies)}, wave-chunks {int(chunks)}, lane slots {int(bodies*L)} "
          f"(util {q.sum()/(bodies*L):.3f}); bodies per wave-chunk {bodies/chunks:.1f}")
cur = jmax.sum() * (73 + 19 + 3)
for C, L, per in ((60, 64, 60), (30, 32, 31)):
    bodies = np.ceil(q / C).sum() * (L / 64.0); chunks = np.ceil(jmax / C).sum()
    new = bodies * per + chunks * 60
    print(f"  issue slots: today {cur/1e6:.1f} M, chunk {C}: {new/1e6:.1f} M  ratio {cur/new:.2f}")
# forward: the walk of today stops when every pixel of the quadrant is closed; model the stop of a pixel as its last contributor (saturated:
# final_T small) or the end of its stream (never saturated)
fpad = np.ones((gy * 16, gx * 16), np.float32); fpad[:H, :W] = fT
fq = fpad.reshape(gy, 2, 8, gx, 2, 8).transpose(0, 3, 1, 4, 2, 5).reshape(gy * gx * 4, 64)
inside = np.zeros((gy * 16, gx * 16), bool); inside[:H, :W] = True
iq = inside.reshape(gy, 2, 8, gx, 2, 8).transpose(0, 3, 1, 4, 2, 5).reshape(gy * gx * 4, 64)
for thr in (2e-4, 1e-3, 1e-2):
    sat = fq < thr
    stop = np.where(sat, np.minimum(q + 1, n[:, None]), n[:, None]) * iq
    walk = stop.max(1)
    print(f"fwd (saturated = final_T < {thr}): saturated pixels {sat.mean():.3f}; wave-records today {walk.sum()} (x62 = {walk.sum()*62/1e6:.1f} M slots); "
          + "; ".join(f"chunk {C}: bodies {int(np.ceil(stop / C).sum())} (x{per} = {np.ceil(stop / C).sum()*per*(L/64)/1e6:.1f} M)" for C, L, per in ((64, 64, 45), (60, 64, 45), (32, 32, 44))))
print("stream length percentiles", {p: int(np.percentile(n, p)) for p in (50, 90, 99)}, "max", n.max(), "; jmax", {p: int(np.percentile(jmax, p)) for p in (50, 90, 99)}, jmax.max())
