#!/usr/bin/env python3
"""Distribution of the quadrant stream lengths of the bench scene (production mode):  python tools/qhist.py [cfg3|cfg4]"""
import math, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from gaussianavatars_amd import debug as D, rasterizer as R
from gaussianavatars_amd.rasterizer import GaussianRasterizationSettings
w = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
N = 200_000 if w == "cfg4" else 100_000
dev = torch.device("cuda:0")
g, cam = bench.build_scene(dev, N, 3, 550, 802, 4, "fused", False)
with torch.no_grad():
    g.select_mesh_by_timestep(0)
    rs = GaussianRasterizationSettings(802, 550, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), torch.ones(3, device=dev), 1.0,
                                       cam.world_view_transform, cam.full_proj_transform, 3, cam.camera_center, False, False)
    prev = R.set_tile_culling(1)
    st = D._forward_state(rs, g.get_xyz, g.get_features, None, g.get_opacity, g.get_scaling, g.get_rotation, None)
    R.set_tile_culling(prev)
q = st["qcount"].cpu().numpy().astype(np.int64).reshape(-1)
nc = st["n_contrib_q"].cpu().numpy().astype(np.int64)
print("quadrants", q.size, "non-empty", int((q > 0).sum()), "entries", int(q.sum()), "binned tile instances", st["num_rendered"], "max", int(q.max()))
for t in (0, 30, 60, 90, 120, 180, 240, 300, 400, 500):
    m = q > t
    print(f"  n > {t:3d}: {int(m.sum()):5d} quadrants, {int(q[m].sum()):8d} entries, entries beyond {t}: {int((q[m] - t).sum()):8d}")
print("pixel walk depth (n_contrib_q): mean %.1f  p50 %d  p90 %d  p99 %d  max %d" % (nc.mean(), np.percentile(nc, 50), np.percentile(nc, 90), np.percentile(nc, 99), nc.max()))
# per-quadrant walked length = max n_contrib_q over its pixels (what the forward wave actually walks, roughly)
H, W = nc.shape
