"""Run-to-run spread of the FLAME-row gradients of the mesh node (tests/test_binding_gpu.py::test_mesh_backward_gather_equals_the_scatter_form, 40
repetitions of its body in both backward forms, the allocator churned in between): the float atomics of the two launches land in arrival order.
prints the worst relative difference between two backwards over one forward per (mode, gradient at the vertices, parameter)."""
import os, sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from gaussianavatars_amd import binding as B, synthetic as S
from test_binding_gpu import _Head
dev = torch.device('cuda:0')
rig = S.flame_rig(4); seq = S.flame_sequence(8, 4); head = _Head(rig, dev, 300)
faces = torch.as_tensor(rig["faces"], device=dev); F = faces.shape[0]
gen = torch.Generator(device="cpu").manual_seed(7)
wts = [torch.randn(s, generator=gen).to(dev) for s in ((F, 3), (F, 3, 3), (F, 1), (F, 4), (1, rig["v_template"].shape[0], 3))]
keys = ("expr", "rotation", "neck_pose", "jaw_pose", "eyes_pose", "translation")
worst = {}
for it in range(40):
    # churn the allocator so that torch.empty hands out dirty blocks
    junk = [torch.full((n,), float('nan'), device=dev) for n in (512, 3 * 5143, 8 * 100, 17 * 10144, 24, 48)]
    del junk
    for vg in (False, True):
        for mode in ("merged", "split"):
            os.environ["GAA_MESH_BWD"] = mode
            fp = {k: torch.as_tensor(v, device=dev).clone().requires_grad_(k in keys) for k, v in seq.items()}
            verts, cano, center, R, scale, quat = B.mesh_frames_timestep(head, fp, 5, faces)
            loss = (center * wts[0]).sum() + (R * wts[1]).sum() + (scale * wts[2]).sum() + (quat * wts[3]).sum()
            if vg: loss = loss + (verts * wts[4]).sum()
            loss.backward(retain_graph=True)
            first = {k: fp[k].grad.clone() for k in keys}
            for k in keys: fp[k].grad = None
            loss.backward()
            for k in keys:
                d = float((fp[k].grad - first[k]).abs().max()); m = float(first[k].abs().max())
                rel = d / m if m > 0 else float('inf') if d > 0 else 0.0
                key = (mode, vg, k)
                if rel != rel or rel > worst.get(key, (0, 0, 0))[0]: worst[key] = (rel, d, m)
                if not (rel <= 1e-4): print("BAD", it, mode, vg, k, "rel", rel, "absdiff", d, "max", m, "nan?", bool(torch.isnan(fp[k].grad).any()), bool(torch.isnan(first[k]).any()))
for k, v in sorted(worst.items()): print(k, "worst rel %.3e diff %.3e max %.3e" % v)
