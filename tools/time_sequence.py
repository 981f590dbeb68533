"""k_blend_seq_mfma (include/gab.h: gab_blend_sequence) and the per-frame FLAME forward with / without its table row, timed with events."""
import ctypes as C, os, sys, torch
sys.path.insert(0, '.')
from gaussianavatars_amd import _lib, binding as B, synthetic as S
from tests.test_binding_gpu import _Head
dev = torch.device('cuda:0')
rig = S.flame_rig(4)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seq = S.flame_sequence(T, 4)
head = _Head(rig, dev, 300)
fp = {k: torch.as_tensor(v, device=dev).clone() for k, v in seq.items()}
def timed(fn, reps=200):
    for _ in range(20): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / reps
with torch.no_grad():
    os.environ["GAA_MESH_SEQUENCE"] = "off"
    off = timed(lambda: B.flame_forward_timestep(head, fp, 5))
    os.environ["GAA_MESH_SEQUENCE"] = "eager"
    B.flame_forward_timestep(head, fp, 5)
    on = timed(lambda: B.flame_forward_timestep(head, fp, 5))
    key, table, prepared, expr = head._gab_sequence
    lib = _lib.gab(); rs = B._rig_struct(head)
    mf = timed(lambda: lib.gab_blend_sequence(C.byref(rs), prepared.data_ptr(), expr.data_ptr(), T, table.data_ptr(), _lib.raw_stream(dev)), 100)
flops = 2.0 * T * rig["shapedirs"].shape[0] * 3 * 100
print(f"T = {T}: per-frame forward {off:.2f} us (host-paced loop) -> {on:.2f} us with its table row; k_blend_seq_mfma {mf:.2f} us = {flops / mf / 1e6:.2f} TFLOP/s fp32 "
      f"({flops / 1e9:.2f} GFLOP, {(6.2e6 + 4.0 * T * 15429) / mf / 1e3:.0f} GB/s of table + output)")
