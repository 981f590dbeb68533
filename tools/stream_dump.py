"""Dumps what decides the cost of a blend mapping at cfg3: the quadrant stream lengths (forward), every pixel's last contributing stream
entry (backward) and, per quadrant stream entry, nothing else -- into gpurun_out/stream_dump.npz, for tools/remap_model.py (CPU)."""
import math, os, sys
import numpy as np, torch
sys.path.insert(0, '.')
import bench
from gaussianavatars_amd.debug import forward_state
from gaussianavatars_amd.rasterizer import GaussianRasterizationSettings
dev = torch.device('cuda:0')
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
n, w, h = {"cfg3": (100_000, 550, 802), "cfg4": (200_000, 550, 802)}[wl]
g, cam = bench.build_scene(dev, n, 3, w, h, 1, "fused", False); g.select_mesh_by_timestep(0)
with torch.no_grad():
    rs = GaussianRasterizationSettings(cam.image_height, cam.image_width, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5),
                                       torch.ones(3, device=dev), 1.0, cam.world_view_transform, cam.full_proj_transform, 3, cam.camera_center, False, False)
    hs = forward_state(rs, g.get_xyz, g.get_features, None, g.get_opacity, g.get_scaling, g.get_rotation, None, tile_culling=1)
os.makedirs("gpurun_out", exist_ok=True)
np.savez_compressed("gpurun_out/stream_dump_%s.npz" % wl, n_contrib_q=hs["n_contrib_q"].cpu().numpy().astype(np.int32),
                    qcount=hs["qcount"].cpu().numpy().astype(np.int32), final_T=hs["final_T"].cpu().numpy())
print("dumped", wl, int(hs["qcount"].sum()))
