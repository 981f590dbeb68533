"""Where does the fp32 error of the FLAME-row gradients come from?  Compares, per timestep of tests/golden/model_pins.npz, the fused kernels and the
composed-torch statement (both on the GPU) with the reference's fp64 evaluation: posed vertices, dL/d(vertices) and the six FLAME-row gradients
under the smooth weights.  `python tools/flame_precision.py` on a GPU box."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.model_pin_inputs import MODEL_PINS, model_pin_inputs  # noqa: E402
from tests.test_model_pins import _model  # noqa: E402

pins = np.load(os.path.join(ROOT, "tests", "golden", "model_pins.npz"))
n = lambda t: t.detach().cpu().numpy().astype(np.float64)
rel = lambda a, b: np.abs(a - b).max() / np.abs(b).max()
dev = "cuda:0" if torch.cuda.is_available() else "cpu"
for impl in (("fused", "unfused") if dev != "cpu" else ("unfused",)):
    g, wt, pd = _model(dev, impl)
    w2 = wt["smooth"]
    for ts in [int(x) for x in os.environ.get("STEPS", ",".join(map(str, MODEL_PINS["steps"]))).split(",")]:
        pre = f"t{ts}_"
        for p in (g._xyz, g._scaling, g._rotation, g._opacity, *g.flame_param.values()):
            p.grad = None
        g.select_mesh_by_timestep(ts)
        v = g.verts
        try:
            v.retain_grad()
        except Exception as e:  # noqa: BLE001
            print("  (verts has no graph:", e, ")")
        ((g.get_xyz * w2["xyz"]).sum() + (g.get_scaling * w2["scaling"]).sum() + (g.get_rotation * w2["rotation"]).sum() +
         (g.get_opacity * w2["opacity"]).sum()).backward()
        v64 = pins[pre + "verts64"]
        line = [f"{impl:8s} t{ts}: verts max|err| {np.abs(n(v)[0] - v64).max():.2e} (ref fp32 {np.abs(pins[pre + 'verts'] - v64).max():.2e})",
                f"cano max|diff to ref fp32| {np.abs(n(g.verts_cano)[0] - pins[pre + 'verts_cano']).max():.2e}"]
        if v.grad is not None:
            line.append(f"dverts {rel(n(v.grad)[0], pins[pre + 'gfs64_verts']):.2e} (ref {float(pins[pre + 'gfs_dev_verts']):.2e})")
        for k in ("expr", "rotation", "neck_pose", "jaw_pose", "eyes_pose", "translation"):
            line.append(f"{k} {rel(n(g.flame_param[k].grad)[ts], pins[pre + 'gfs64_' + k]):.2e} (ref {float(pins[pre + 'gfs_dev_' + k]):.2e})")
        print("; ".join(line), flush=True)
