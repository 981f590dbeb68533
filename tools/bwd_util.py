"""Lane utilisation of the blend backward: per 8x8 quadrant, how many stream records are walked with few active pixels?"""
import sys, numpy as np, torch, math
sys.path.insert(0, '.')
import bench
from gaussianavatars_amd.debug import forward_state
from gaussianavatars_amd.rasterizer import GaussianRasterizationSettings
dev = torch.device('cuda:0')
g, cam = bench.build_scene(dev, 100_000, 3, 550, 802, 1, "fused", False); g.select_mesh_by_timestep(0)
with torch.no_grad():
    rs = GaussianRasterizationSettings(cam.image_height, cam.image_width, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5),
                                       torch.ones(3, device=dev), 1.0, cam.world_view_transform, cam.full_proj_transform, 3, cam.camera_center, False, False)
    hs = forward_state(rs, g.get_xyz, g.get_features, None, g.get_opacity, g.get_scaling, g.get_rotation, None, tile_culling=1)
nq = hs["n_contrib_q"].cpu().numpy().astype(np.int64)
H, W = nq.shape
Hp, Wp = (H + 7) // 8 * 8, (W + 7) // 8 * 8
pad = np.zeros((Hp, Wp), np.int64); pad[:H, :W] = nq
q = pad.reshape(Hp // 8, 8, Wp // 8, 8).transpose(0, 2, 1, 3).reshape(-1, 64)
q = np.sort(q, axis=1)[:, ::-1]          # descending per quadrant
jmax = q[:, 0]
tot = jmax.sum()
print("quadrant waves", len(q), "sum jmax (wave-records walked)", tot, "sum of per-pixel last (useful pixel-records)", q.sum(), "utilisation %.3f" % (q.sum() / (64.0 * tot)))
for k in (2, 4, 8, 16, 32):
    print(f"  records walked with fewer than {k:2d} active pixels: {100.0 * (jmax - q[:, k - 1]).sum() / tot:5.1f} %")
print("jmax percentiles:", {p: int(np.percentile(jmax, p)) for p in (50, 90, 99, 99.9)}, "max", int(jmax.max()), "top5", np.sort(jmax)[-5:])
qc = hs["qcount"].cpu().numpy().astype(np.int64).reshape(-1)
print("forward stream lengths: sum", qc.sum(), "percentiles", {p: int(np.percentile(qc, p)) for p in (50, 90, 99, 99.9)}, "max", int(qc.max()))
