import sys, numpy as np, torch, math
sys.path.insert(0, '.')
import bench
from gaussianavatars_amd.debug import forward_state
from gaussianavatars_amd.rasterizer import GaussianRasterizationSettings
dev = torch.device('cuda:0')
for wl in ("cfg3", "cfg5"):
    if wl == "cfg3":
        g, cam = bench.build_scene(dev, 100_000, 3, 550, 802, 1, "fused", False); g.select_mesh_by_timestep(0)
    else:
        g, cam = bench.build_unbound_scene(dev, 2_000_000, 3, 1100, 1600)
    with torch.no_grad():
        rs = GaussianRasterizationSettings(cam.image_height, cam.image_width, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5),
                                           torch.ones(3, device=dev), 1.0, cam.world_view_transform, cam.full_proj_transform, 3, cam.camera_center, False, False)
        hs = forward_state(rs, g.get_xyz, g.get_features, None, g.get_opacity, g.get_scaling, g.get_rotation, None, tile_culling=1)
    tc = hs["tile_count"].cpu().numpy().astype(np.int64)
    edges = [0, 1, 256, 512, 1024, 2048, 3072, 4096, 6144, 8192, 16384, 1 << 30]
    h, _ = np.histogram(tc, edges)
    w, _ = np.histogram(tc, edges, weights=tc)
    print(wl, "tiles", len(tc), "I", tc.sum(), "max", tc.max())
    for a, b, c, d in zip(edges[:-1], edges[1:], h, w):
        print(f"  [{a:6d},{b:10d})  tiles {c:5d}  instances {int(d):9d}")
