"""How many pixels of a deep quadrant are still open along its walk (cfg3 scene): open(j) = #pixels whose last contributor lies beyond stream entry j."""
import sys; sys.path.insert(0, '.')
import math, numpy as np, torch
import bench
from gaussianavatars_amd import debug as D, rasterizer as R
from gaussianavatars_amd.rasterizer import GaussianRasterizationSettings
dev = torch.device('cuda:0')
H, W = 802, 550
g, cam = bench.build_scene(dev, 100_000, 3, W, H, 4, "fused", True)
g.bound_render = False
g.select_mesh_by_timestep(1)
tfx, tfy = math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5)
rs = GaussianRasterizationSettings(H, W, tfx, tfy, torch.ones(3, device=dev), 1.0, cam.world_view_transform, cam.full_proj_transform, 3, cam.camera_center, False, False)
with torch.no_grad():
    st = D.forward_state(rs, g.get_xyz, g.get_features, None, g.get_opacity, g.get_scaling, g.get_rotation, None, tile_culling=True)
nq = st["n_contrib_q"].cpu().numpy()          # (H, W) last contributor position in the quadrant stream (0 = none)
gy, gx = (H + 15) // 16, (W + 15) // 16
pad = np.zeros((gy * 16, gx * 16), np.int64); pad[:H, :W] = nq
q = pad.reshape(gy, 2, 8, gx, 2, 8).transpose(0, 3, 1, 4, 2, 5).reshape(gy * gx * 4, 64)   # per quadrant: 64 pixel depths
depth = q.max(1)
print("quadrants", len(depth), "walk depth mean %.1f p90 %d p99 %d max %d" % (depth[depth > 0].mean(), np.percentile(depth, 90), np.percentile(depth, 99), depth.max()))
total = int(depth.sum())
print("wave-records walked (sum of quadrant depths): %.3f M" % (total / 1e6))
for thr in (32, 16, 8, 4):
    # entries a quadrant walks with more than `thr` pixels open = the (64 - thr)-th smallest... = sorted depth at index 63 - thr
    srt = np.sort(q, axis=1)
    wide = srt[:, 63 - thr]       # beyond this entry at most thr pixels are open
    print("beyond '<= %2d open': %.3f M wave-records (%.1f %%); deepest quadrant: %d of %d entries" % (
        thr, (depth - wide).sum() / 1e6, 100.0 * (depth - wide).sum() / total, int((depth - wide)[depth.argmax()]), int(depth.max())))
top = np.argsort(-depth)[:10]
for t in top:
    s_ = np.sort(q[t])
    print("quadrant depth %4d: open>32 until %4d, >16 until %4d, >8 until %4d, >4 until %4d" % (depth[t], s_[31], s_[47], s_[55], s_[59]))
