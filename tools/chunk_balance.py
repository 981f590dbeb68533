"""How evenly do the rank path's chunks (k_rcount / k_rdscatter / k_rscatter: contiguous runs of splats, one per workgroup) share the tile
instances?  Binned rect sizes of the bench scene from the forward state; prints max / mean of the per-chunk instance sums for the chunking in
use and for chunks cut at equal instance counts on 256-splat boundaries (what a prefix over k_preprocess's per-workgroup sums could give)."""
import math, sys
import numpy as np, torch
sys.path.insert(0, '.')
import bench
from gaussianavatars_amd.debug import forward_state
from gaussianavatars_amd.rasterizer import GaussianRasterizationSettings
dev = torch.device('cuda:0')
for wl in sys.argv[1:] or ["cfg3", "cfg4"]:
    n, w, h = {"cfg3": (100_000, 550, 802), "cfg4": (200_000, 550, 802)}[wl]
    g, cam = bench.build_scene(dev, n, 3, w, h, 1, "fused", False); g.select_mesh_by_timestep(0)
    with torch.no_grad():
        rs = GaussianRasterizationSettings(cam.image_height, cam.image_width, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5),
                                           torch.ones(3, device=dev), 1.0, cam.world_view_transform, cam.full_proj_transform, 3, cam.camera_center, False, False)
        hs = forward_state(rs, g.get_xyz, g.get_features, None, g.get_opacity, g.get_scaling, g.get_rotation, None, tile_culling=1)
    t = hs["tiles_touched"].cpu().numpy().astype(np.int64)      # rect-based; the snug rects are ~0.8 of these, evenly enough for this question
    P = len(t)
    nblk = min((P + 255) // 256, 256)
    chunk = ((P + nblk - 1) // nblk + 31) // 32 * 32
    sums = np.array([t[i:i + chunk].sum() for i in range(0, P, chunk)])
    print(f"{wl}: P {P}, instances (rect) {t.sum()}, chunks of {chunk} splats: {len(sums)} workgroups, max / mean {sums.max() / sums.mean():.2f}, "
          f"p90 / mean {np.percentile(sums, 90) / sums.mean():.2f}, largest rect {t.max()} tiles")
    g256 = np.add.reduceat(t, np.arange(0, P, 256))
    cum = np.cumsum(g256)
    cuts = np.searchsorted(cum, np.arange(1, nblk) * cum[-1] / nblk)
    b = np.concatenate([[0], cuts, [len(g256)]])
    bs = np.array([g256[b[i]:b[i + 1]].sum() for i in range(nblk)])
    print(f"   cut at equal instance counts on 256-splat boundaries: max / mean {bs.max() / bs.mean():.2f}")
