"""Seeded test scenes shared by the CPU and GPU tests (rasterizer-level inputs)."""
import math

import numpy as np

from gaussianavatars_amd import synthetic as S


def scene(name):
    """-> (camera, splat dict, bg, sh_degree, scale_modifier)"""
    if name == "cfg1":  # BASELINE.json config 1: 1k SH-0 splats, 256x256
        return S.orbit_camera(256, 256), S.random_splats(1000, 0, 1), [1.0, 1.0, 1.0], 0, 1.0
    if name == "sh3_small":  # SH-3, odd image size (ragged right/bottom tiles), coloured bg
        sp = S.random_splats(3000, 3, 11, xyz_sigma=0.05, log_scale_mean=math.log(0.003))
        return S.orbit_camera(203, 141), sp, [0.2, 0.7, 0.4], 3, 1.0
    if name == "sh2_mod":  # active degree below the stored one, scale modifier != 1
        sp = S.random_splats(2000, 3, 12, xyz_sigma=0.05, log_scale_mean=math.log(0.003))
        return S.orbit_camera(160, 96, yaw_deg=25, pitch_deg=-10), sp, [0.0, 0.0, 0.0], 2, 1.4
    if name == "culls":  # splats behind the camera, outside the frustum, degenerate and huge ones
        sp = S.random_splats(1500, 1, 13, xyz_sigma=0.3, log_scale_mean=math.log(0.01), log_scale_sigma=1.2)
        sp["means3D"][:100, 2] += 1.5  # behind / at the camera plane (camera sits at z=+1)
        sp["means3D"][100:200, 0] += 0.6  # far outside the 1.3x frustum
        sp["scales"][200:210] = 0.0  # degenerate covariance
        sp["scales"][210:215] = 0.5  # screen-filling
        return S.orbit_camera(128, 128), sp, [1.0, 1.0, 1.0], 1, 1.0
    if name == "dense_tile":  # > 8192 instances in one tile: the global-memory sort fallback
        sp = S.random_splats(12000, 0, 14, xyz_sigma=0.002, log_scale_mean=math.log(0.0008), log_scale_sigma=0.2)
        return S.orbit_camera(64, 64), sp, [1.0, 1.0, 1.0], 0, 1.0
    if name == "dense_tile_xl":  # > 16384 instances in one tile: the global-memory sort fallback
        sp = S.random_splats(20000, 0, 16, xyz_sigma=0.002, log_scale_mean=math.log(0.0008), log_scale_sigma=0.2)
        return S.orbit_camera(64, 64), sp, [1.0, 1.0, 1.0], 0, 1.0
    if name == "huge_grid":  # 207x207 = 42849 tiles: beyond the LDS tile histogram, the L2-atomic binning path
        sp = S.random_splats(1500, 1, 17, xyz_sigma=0.12, log_scale_mean=math.log(0.004), log_scale_sigma=0.6)
        return S.orbit_camera(3300, 3300), sp, [0.3, 0.3, 0.3], 1, 1.0
    if name == "depth_ties":  # many splats at exactly the same depth: the (depth, index) tie order of the stable sort
        sp = S.random_splats(4000, 0, 18, xyz_sigma=0.05, log_scale_mean=math.log(0.004))
        sp["means3D"][:, 2] = np.round(sp["means3D"][:, 2] * 40.0) / 40.0   # the default camera looks down -z: depth = 1 - z
        return S.orbit_camera(96, 96), sp, [0.5, 0.5, 0.5], 0, 1.0
    if name == "deep_stack":  # ~1500 faint splats over the same few tiles: every pixel's blend walks through all the backward's
        # segments (GSR_BWD_SEGMENT entries each, the last one open-ended), many pixels never saturate (tail mode, and the
        # checkpoints it writes), the opaque sixth closes others mid-stream
        sp = S.random_splats(1500, 1, 19, xyz_sigma=0.012, log_scale_mean=math.log(0.012), log_scale_sigma=0.3)
        sp["opacities"][:] = 0.006 + 0.01 * np.random.default_rng(20).random((1500, 1), dtype=np.float32)
        sp["opacities"][::6] = 0.2
        sp["means3D"][::6, :2] += 0.02
        return S.orbit_camera(48, 40), sp, [0.4, 0.1, 0.6], 1, 1.0
    if name == "empty_view":  # nothing visible
        sp = S.random_splats(500, 0, 15)
        sp["means3D"][:, 2] += 3.0
        return S.orbit_camera(96, 80), sp, [0.1, 0.2, 0.3], 0, 1.0
    raise KeyError(name)


def settings_args(cam, bg, deg, mod):
    return dict(H=cam.image_height, W=cam.image_width, tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5),
                bg=np.asarray(bg, np.float32), scale_modifier=mod, viewmatrix=cam.world_view_transform,
                projmatrix=cam.full_proj_transform, sh_degree=deg, campos=cam.camera_center)
