"""scratch: busiest / mean instance count of the rank passes' equal-count chunks"""
import math, sys, torch
sys.path.insert(0, '.')
import bench
from gaussianavatars_amd.debug import forward_state
from gaussianavatars_amd.rasterizer import GaussianRasterizationSettings
dev = torch.device('cuda:0')
for name, n, w, h in (("cfg5", 2_000_000, 1100, 1600), ("cfg4", 200_000, 550, 802), ("cfg3", 100_000, 550, 802)):
    g, cam = bench.build_scene(dev, n, 3, w, h, 1, "fused", False); g.select_mesh_by_timestep(0)
    with torch.no_grad():
        rs = GaussianRasterizationSettings(cam.image_height, cam.image_width, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5),
                                           torch.ones(3, device=dev), 1.0, cam.world_view_transform, cam.full_proj_transform, 3, cam.camera_center, False, False)
        hs = forward_state(rs, g.get_xyz, g.get_features, None, g.get_opacity, g.get_scaling, g.get_rotation, None, tile_culling=1, fast_blend=True)
    r = hs["rect"].to(torch.int64)
    nt = ((r[:, 2] - r[:, 0]) * (r[:, 3] - r[:, 1])).clamp(min=0)      # rect-based count (an upper bound of the culled one)
    P = n; nblk = 256
    chunk = ((P + nblk - 1) // nblk + 31) // 32 * 32
    pad = chunk * nblk - P
    s = torch.cat([nt, torch.zeros(pad, dtype=nt.dtype, device=nt.device)]).view(nblk, chunk).sum(1).double()
    print(name, "instances", int(nt.sum()), "chunks of", chunk, "splats: busiest / mean = %.2f" % float(s.max() / s.mean()), "p90/mean %.2f" % float(s.quantile(0.9) / s.mean()), flush=True)
    del g, hs
    torch.cuda.empty_cache()
