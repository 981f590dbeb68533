"""k_render_bwd_rp: how many four-pixel bodies does the backward run, and how many would it run with the alive pixels of a (quadrant, chunk)
packed four to a body?  Reads gpurun_out/stream_dump_<wl>.npz (tools/stream_dump.py).  A pixel is alive in chunk c of its quadrant's stream iff
its last contributing entry lies above 60 c; the kernel skips a half-row (four horizontally adjacent pixels) only when none of the four is."""
import sys
import numpy as np

wl = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
d = np.load("gpurun_out/stream_dump_%s.npz" % wl)
nq = d["n_contrib_q"].astype(np.int64)
H, W = nq.shape
Hp, Wp = (H + 7) // 8 * 8, (W + 7) // 8 * 8
pad = np.zeros((Hp, Wp), np.int64)
pad[:H, :W] = nq
q = pad.reshape(Hp // 8, 8, Wp // 8, 8).transpose(0, 2, 1, 3).reshape(-1, 8, 2, 4)   # quadrant, row, half, pixel
jtop = q.reshape(len(q), -1).max(axis=1)
CH = 60
bodies = packed = units = alive_tot = rows_any = 0
hist = np.zeros(5, np.int64)
for c in range(int((jtop.max() + CH - 1) // CH)):
    sel = jtop > CH * c
    if not sel.any():
        break
    a = q[sel] > CH * c                          # alive pixels of the quadrants that reach chunk c
    hr = a.any(axis=3)                           # half-rows with something alive
    n_alive = a.reshape(len(a), -1).sum(axis=1)
    bodies += int(hr.sum())
    packed += int(((n_alive + 3) // 4).sum())
    units += int(sel.sum())
    alive_tot += int(n_alive.sum())
    rows_any += int(a.reshape(len(a), 8, 8).any(axis=2).sum())
    per_body = a.sum(axis=3)[hr]
    hist += np.bincount(per_body, minlength=5)[:5]
print(f"{wl}: chunks (units incl. the open-ended tail) {units}, four-pixel bodies run {bodies}, alive pixel-chunks {alive_tot}")
print(f"  alive pixels per body run: {alive_tot / bodies:.2f} of 4   histogram 1..4: {hist[1:].tolist()}")
print(f"  bodies with the alive pixels packed four to a body: {packed} ({100.0 * packed / bodies:.1f} % of today's)")
print(f"  rows (8 pixels) with something alive: {rows_any}; bodies if a row were the unit of skipping: {2 * rows_any}")
