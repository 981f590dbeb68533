"""How many (splat, tile) instances survive an exact ellipse-vs-tile test?  (union of the four quadrant streams)"""
import sys, numpy as np, torch
sys.path.insert(0, '.')
import bench
from gaussianavatars_amd.debug import forward_state
from gaussianavatars_amd.rasterizer import GaussianRasterizationSettings
import math
dev = torch.device('cuda:0')
for wl in ("cfg3", "cfg5"):
    if wl == "cfg3":
        g, cam = bench.build_scene(dev, 100_000, 3, 550, 802, 1, "fused", False)
        g.select_mesh_by_timestep(0)
    else:
        g, cam = bench.build_unbound_scene(dev, 2_000_000, 3, 1100, 1600)
    with torch.no_grad():
        rs = GaussianRasterizationSettings(cam.image_height, cam.image_width, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5),
                                           torch.ones(3, device=dev), 1.0, cam.world_view_transform, cam.full_proj_transform, 3,
                                           cam.camera_center, False, False)
        hs = forward_state(rs, g.get_xyz, g.get_features, None, g.get_opacity, g.get_scaling, g.get_rotation, None)
    I = hs["num_rendered"]
    qpos = hs["qlist"].cpu().numpy().astype(np.int64); qcnt = hs["qcount"].cpu().numpy().astype(np.int64); rng = hs["ranges"].cpu().numpy().astype(np.int64)
    surv = 0; pairs = int(qcnt.sum())
    for t in range(rng.shape[0]):
        n, start = rng[t, 1] - rng[t, 0], rng[t, 0]
        if n == 0: continue
        seen = np.zeros(n, bool)
        for q in range(4):
            seen[qpos[4 * start + q * n: 4 * start + q * n + qcnt[t, q]]] = True
        surv += int(seen.sum())
    print(wl, "I", I, "tile-level survivors", surv, "%.3f" % (surv / I), "quadrant pairs", pairs, "%.3f" % (pairs / (4 * I)))
