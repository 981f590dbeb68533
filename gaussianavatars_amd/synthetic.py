"""Seeded synthetic stand-ins for the assets BASELINE.json names but the reference snapshot lacks
(media/306/point_cloud.ply, flame_param.npz, flame2023.pkl -- SURVEY.md F4).

Everything here is plain numpy on the host so that the same bytes are produced on the CPU
container and on the GPU box.  Shapes, distributions and seeds follow SURVEY.md section 8(d).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Tuple

import numpy as np

# --------------------------------------------------------------------------------------------
# cameras
# --------------------------------------------------------------------------------------------


@dataclass
class SynthCamera:
    """Duck-type of what gaussian_renderer/__init__.py:34-47 reads from a camera."""
    FoVx: float
    FoVy: float
    image_height: int
    image_width: int
    world_view_transform: np.ndarray  # (4,4) = W2C^T   (scene/cameras.py:44)
    full_proj_transform: np.ndarray   # (4,4) = (P @ W2C)^T
    camera_center: np.ndarray         # (3,)
    timestep: int = 0


def orbit_camera(width: int, height: int, r: float = 1.0, fovy_deg: float = 20.0, znear: float = 0.01,
                 zfar: float = 10.0, yaw_deg: float = 0.0, pitch_deg: float = 0.0) -> SynthCamera:
    """Equivalent of fps_benchmark_demo.py:21-33 (OrbitCamera(W,H,r=1,fovy=20,'opencv'),
    utils/viewer_utils.py:73-170) restated: camera on the +z axis at distance r looking at the
    origin, OpenCV axes, OpenGL-style projection with z_sign=+1."""
    focal = height / (2.0 * math.tan(math.radians(fovy_deg) / 2.0))
    fovx = 2.0 * math.atan(width / (2.0 * focal))
    cx, cy = width // 2, height // 2
    # camera-to-world in OpenGL axes, then flip y,z columns for OpenCV
    pose = np.eye(4, dtype=np.float64)
    pose[2, 3] = r
    cyw, syw = math.cos(math.radians(yaw_deg)), math.sin(math.radians(yaw_deg))
    cp, sp = math.cos(math.radians(pitch_deg)), math.sin(math.radians(pitch_deg))
    Ry = np.array([[cyw, 0, syw, 0], [0, 1, 0, 0], [-syw, 0, cyw, 0], [0, 0, 0, 1]], np.float64)
    Rx = np.array([[1, 0, 0, 0], [0, cp, -sp, 0], [0, sp, cp, 0], [0, 0, 0, 1]], np.float64)
    pose = Ry @ Rx @ pose
    pose[:, [1, 2]] *= -1
    w2c = np.linalg.inv(pose)
    proj = np.zeros((4, 4), np.float64)
    proj[0, 0] = focal * 2 / width
    proj[1, 1] = focal * 2 / height
    proj[0, 2] = (width - 2 * cx) / width
    proj[1, 2] = (height - 2 * cy) / height
    proj[2, 2] = (zfar + znear) / (zfar - znear)
    proj[2, 3] = -2 * zfar * znear / (zfar - znear)
    proj[3, 2] = 1.0
    full = proj @ w2c
    return SynthCamera(
        FoVx=float(fovx), FoVy=float(math.radians(fovy_deg)), image_height=int(height), image_width=int(width),
        world_view_transform=np.ascontiguousarray(w2c.T.astype(np.float32)),
        full_proj_transform=np.ascontiguousarray(full.T.astype(np.float32)),
        camera_center=pose[:3, 3].astype(np.float32),
    )


# --------------------------------------------------------------------------------------------
# un-bound splat clouds (configs 1 and 5)
# --------------------------------------------------------------------------------------------


def random_splats(n: int, sh_degree: int, seed: int, xyz_sigma: float = 0.06, log_scale_mean: float = math.log(0.004),
                  log_scale_sigma: float = 0.4, opacity_mu: float = 0.0, opacity_sigma: float = 1.5) -> Dict[str, np.ndarray]:
    """Rasterizer-level inputs (already activated), SURVEY.md 8(d) cfg 1 / cfg 5."""
    g = np.random.default_rng(seed)
    M = (sh_degree + 1) ** 2
    means3D = g.normal(0.0, xyz_sigma, (n, 3)).astype(np.float32)
    scales = np.exp(g.normal(log_scale_mean, log_scale_sigma, (n, 3))).astype(np.float32)
    rotations = g.normal(0.0, 1.0, (n, 4)).astype(np.float32)  # un-normalised, as the kernel receives it
    opacities = (1.0 / (1.0 + np.exp(-g.normal(opacity_mu, opacity_sigma, (n, 1))))).astype(np.float32)
    shs = np.zeros((n, M, 3), np.float32)
    shs[:, 0, :] = g.normal(0.5, 0.5, (n, 3))
    if M > 1:
        shs[:, 1:, :] = g.normal(0.0, 0.08, (n, M - 1, 3))
    return dict(means3D=means3D, scales=scales, rotations=rotations, opacities=opacities, shs=shs)


# --------------------------------------------------------------------------------------------
# FLAME-shaped synthetic rig (V=5143, F=10144, J=5) and bound splats (configs 2-4)
# --------------------------------------------------------------------------------------------

FLAME_V = 5143      # 5023 + 120 teeth vertices (flame_model/flame.py:186-483)
FLAME_F = 10144     # 9976 + 168 teeth faces
FLAME_J = 5
FLAME_PARENTS = (-1, 0, 1, 1, 1)
N_SHAPE, N_EXPR = 300, 100


# Face-area statistics of the reference's head template (flame_model/assets/flame/head_template_mesh.obj, 9976 faces), the only thing taken from it:
# its LORENZ CURVE -- (fraction of the faces, smallest first; fraction of the surface area they hold).  Half of the template's faces (eyes, lips,
# nose, ears) hold 4.5 % of its area, the largest tenth (scalp, neck) 59 %; the lat/long ellipsoid below is far more even (half: 29 %, tenth: 16 %).
TEMPLATE_AREA_LORENZ = ((0.0, 0.0), (0.01, 0.00011), (0.02, 0.00026), (0.05, 0.00087), (0.1, 0.0025), (0.2, 0.0077), (0.3, 0.01621), (0.4, 0.02878),
                        (0.5, 0.04498), (0.6, 0.06627), (0.7, 0.10212), (0.75, 0.12977), (0.8, 0.17308), (0.85, 0.26119), (0.9, 0.41049),
                        (0.95, 0.63332), (0.98, 0.82026), (0.99, 0.89986), (1.0, 1.0))
TEMPLATE_HALF_EXTENT = (0.1036, 0.1568, 0.1109)   # the template's bounding box / 2 (metres)
TEMPLATE_DENSE_AXIS = (0.0, 0.3, 0.95)            # where its small faces sit, seen from the box centre: the front, a little above the middle


def _template_like_directions(d: np.ndarray) -> np.ndarray:
    """Moves unit directions `d` along great circles through TEMPLATE_DENSE_AXIS so that the cap holding a fraction F of the (evenly spread)
    directions ends up covering the fraction TEMPLATE_AREA_LORENZ(F) of the sphere: the faces of a mesh built on them get the template's
    uneven area distribution -- dense at the front, coarse at the back -- on the same lat/long topology."""
    c = np.asarray(TEMPLATE_DENSE_AXIS, np.float64)
    c /= np.linalg.norm(c)
    cosa = np.clip(d @ c, -1.0, 1.0)
    perp = d - cosa[:, None] * c
    pn = np.linalg.norm(perp, axis=1, keepdims=True)
    perp = np.where(pn > 1e-12, perp / np.maximum(pn, 1e-12), 0.0)
    F = (1.0 - cosa) / 2.0
    kf, ka = np.asarray(TEMPLATE_AREA_LORENZ, np.float64).T
    A = np.interp(F, kf, ka)
    cos2 = 1.0 - 2.0 * A
    sin2 = np.sqrt(np.maximum(0.0, 1.0 - cos2 * cos2))
    return cos2[:, None] * c + sin2[:, None] * perp


def head_mesh(half_extent: float = 0.12, kind: str = "ellipsoid") -> Tuple[np.ndarray, np.ndarray]:
    """An ellipsoid 'head' with exactly FLAME_V vertices and FLAME_F faces: a 53-ring x 97-segment
    lat/long sphere (2 + 53*97 = 5143 vertices, 10282 faces) with a 138-face neck hole cut at the
    south pole.  The true template (flame_model/assets/flame/head_template_mesh.obj) is licensed and
    does not travel to the GPU box; only the counts and the index dtype matter to the kernels."""
    rings, seg = 53, 97
    verts = [(0.0, 1.0, 0.0)]
    for i in range(1, rings + 1):
        th = math.pi * i / (rings + 1)
        for j in range(seg):
            ph = 2 * math.pi * j / seg
            verts.append((math.sin(th) * math.cos(ph), math.cos(th), math.sin(th) * math.sin(ph)))
    verts.append((0.0, -1.0, 0.0))
    verts = np.asarray(verts, np.float64)
    assert verts.shape[0] == FLAME_V
    faces = []
    ring = lambda i, j: 1 + (i - 1) * seg + (j % seg)
    for j in range(seg):
        faces.append((0, ring(1, j + 1), ring(1, j)))
    for i in range(1, rings):
        for j in range(seg):
            a, b, c, d = ring(i, j), ring(i, j + 1), ring(i + 1, j), ring(i + 1, j + 1)
            faces.append((a, b, d))
            faces.append((a, d, c))
    south = FLAME_V - 1
    for j in range(seg):
        faces.append((south, ring(rings, j), ring(rings, j + 1)))
    faces = np.asarray(faces, np.int64)
    assert faces.shape[0] == 10282
    faces = faces[: FLAME_F]  # drops the 97 south-pole fan triangles + 41 of the last band
    if kind == "template_like":
        # the same topology with the template's face-area distribution and extent (statistics only: TEMPLATE_AREA_LORENZ)
        verts = _template_like_directions(verts) * np.asarray(TEMPLATE_HALF_EXTENT)
        return verts.astype(np.float32), faces
    if kind != "ellipsoid":
        raise ValueError(f"head_mesh: unknown kind {kind!r}")
    verts = verts * np.array([0.8, 1.0, 0.9]) * half_extent
    return verts.astype(np.float32), faces


def flame_rig(seed: int = 4, kind: str = "ellipsoid") -> Dict[str, np.ndarray]:
    """Buffers with the schemas FlameHead registers (flame_model/flame.py:98-129, after add_teeth).
    kind="template_like": the rig of `flame_pickle_dict` (what tools/ref_on_gpu.py stages on the real template) on `head_mesh(kind="template_like")`."""
    if kind != "ellipsoid":
        v_template, faces = head_mesh(kind=kind)
        d = flame_pickle_dict(v_template, faces, seed)
        return dict(v_template=v_template, shapedirs=d["shapedirs"], posedirs=np.ascontiguousarray(d["posedirs"].reshape(-1, 36).T),
                    J_regressor=d["J_regressor"].astype(np.float32), lbs_weights=d["weights"].astype(np.float32),
                    parents=np.asarray(FLAME_PARENTS, np.int64), faces=faces)
    g = np.random.default_rng(seed)
    v_template, faces = head_mesh()
    V = FLAME_V
    # smooth-ish blend directions: low-frequency functions of position times random weights
    basis = np.concatenate([v_template / 0.12, np.sin(v_template / 0.12 * 3.0), np.cos(v_template / 0.12 * 2.0)], 1)  # (V,9)
    mix = g.normal(0.0, 1.0, (9, 3 * (N_SHAPE + N_EXPR)))
    shapedirs = (basis @ mix).reshape(V, 3, N_SHAPE + N_EXPR) * (2e-3 / 3.0)
    shapedirs += g.normal(0.0, 2e-4, shapedirs.shape)
    posedirs = g.normal(0.0, 1e-3, (36, 3 * V))
    # joints: root at the neck base, neck, jaw, two eyes
    joints = np.array([[0, -0.10, 0], [0, -0.05, 0], [0, -0.02, 0.03], [0.03, 0.04, 0.08], [-0.03, 0.04, 0.08]], np.float64)
    J_regressor = np.zeros((FLAME_J, V))
    lbs_weights = np.zeros((V, FLAME_J))
    for j in range(FLAME_J):
        d2 = ((v_template - joints[j]) ** 2).sum(1)
        near = np.argsort(d2)[:50]
        w = np.exp(-d2[near] / (2 * 0.02 ** 2))
        J_regressor[j, near] = w / w.sum()
        lbs_weights[:, j] = np.exp(-d2 / (2 * (0.06 if j < 3 else 0.015) ** 2))
    lbs_weights[:, 0] += 1e-3
    lbs_weights /= lbs_weights.sum(1, keepdims=True)
    return dict(
        v_template=v_template.astype(np.float32),
        shapedirs=shapedirs.astype(np.float32),
        posedirs=posedirs.astype(np.float32),
        J_regressor=J_regressor.astype(np.float32),
        lbs_weights=lbs_weights.astype(np.float32),
        parents=np.asarray(FLAME_PARENTS, np.int64),
        faces=faces,
    )


def flame_sequence(T: int, seed: int = 4) -> Dict[str, np.ndarray]:
    """flame_param.npz schema (scene/flame_gaussian_model.py:61-71): smooth random walk of the
    expression, sinusoidal jaw, small neck/global rotation, 1 cm translation jitter."""
    g = np.random.default_rng(seed + 1000)
    t = np.arange(T)[:, None]
    expr = np.cumsum(g.normal(0, 0.15, (T, N_EXPR)), 0)
    expr = expr / max(1.0, np.abs(expr).max() / 2.0)
    jaw = np.zeros((T, 3))
    jaw[:, 0] = 0.15 * (1 - np.cos(2 * np.pi * t[:, 0] / 60.0))
    return dict(
        shape=g.normal(0, 1.0, (N_SHAPE,)).astype(np.float32),
        expr=expr.astype(np.float32),
        rotation=(0.05 * np.sin(2 * np.pi * t / 150.0 + np.array([0, 1, 2]))).astype(np.float32),
        neck_pose=(0.08 * np.sin(2 * np.pi * t / 90.0 + np.array([1, 2, 3]))).astype(np.float32),
        jaw_pose=jaw.astype(np.float32),
        eyes_pose=(0.1 * np.sin(2 * np.pi * t / 45.0 + np.arange(6))).astype(np.float32),
        translation=g.normal(0, 0.01 / 3, (T, 3)).astype(np.float32),
        static_offset=g.normal(0, 2e-4, (1, FLAME_V, 3)).astype(np.float32),
        dynamic_offset=np.zeros((T, FLAME_V, 3), np.float32),
    )


def bound_splats(n: int, n_faces: int, sh_degree: int, seed: int) -> Dict[str, np.ndarray]:
    """Leaf parameters of a mesh-bound GaussianModel (scene/gaussian_model.py:185-205 schema):
    every face gets at least one splat, the remainder is uniform (SURVEY.md 8(d) cfg 2)."""
    g = np.random.default_rng(seed)
    assert n >= n_faces
    binding = np.concatenate([np.arange(n_faces), g.integers(0, n_faces, n - n_faces)]).astype(np.int64)
    g.shuffle(binding)
    M = (sh_degree + 1) ** 2
    rot = g.normal(0, 1, (n, 4))
    rot /= np.linalg.norm(rot, axis=1, keepdims=True)
    out = dict(
        _xyz=g.normal(0, 0.35, (n, 3)).astype(np.float32),
        _scaling=g.normal(math.log(0.35), 0.35, (n, 3)).astype(np.float32),
        _rotation=rot.astype(np.float32),
        _opacity=g.normal(1.0, 1.5, (n, 1)).astype(np.float32),
        _features_dc=g.normal(0.5, 0.5, (n, 1, 3)).astype(np.float32),
        _features_rest=g.normal(0.0, 0.08, (n, M - 1, 3)).astype(np.float32),
        binding=binding,
    )
    return out


# --------------------------------------------------------------------------------------------
# assets in the REFERENCE's on-disk formats: what an unchanged entry script needs to start
# --------------------------------------------------------------------------------------------


def read_obj_topology(path: str) -> Tuple[np.ndarray, np.ndarray]:
    """(verts (V,3) f32, faces (F,3) int64, 0-based) of a triangle OBJ with `f v/vt v/vt v/vt` records (the FLAME template's layout)."""
    v, f = [], []
    with open(path) as fh:
        for ln in fh:
            if ln.startswith("v "):
                v.append([float(x) for x in ln.split()[1:4]])
            elif ln.startswith("f "):
                f.append([int(tok.split("/")[0]) - 1 for tok in ln.split()[1:4]])
    return np.asarray(v, np.float32), np.asarray(f, np.int64)


def flame_pickle_dict(v_template: np.ndarray, faces: np.ndarray, seed: int = 4) -> Dict[str, np.ndarray]:
    """A dict with the keys and shapes FlameHead.__init__ reads from flame2023.pkl (flame_model/flame.py:98-129): `v_template (V,3)`,
    `shapedirs (V,3,400)` (first 300 shape, then 100 expression directions), `posedirs (V,3,36)`, `J_regressor (5,V)`,
    `kintree_table (2,5)`, `weights (V,5)`, `f (F,3)` -- synthetic values on the given topology (the licensed model cannot be shipped):
    smooth blend directions, joints at neck base / neck / jaw / two eyes of the head's bounding box."""
    g = np.random.default_rng(seed)
    V = v_template.shape[0]
    lo, hi = v_template.min(0), v_template.max(0)
    c, ext = (lo + hi) / 2, float((hi - lo).max()) / 2
    u = (v_template - c) / ext                                              # roughly [-1, 1]^3
    basis = np.concatenate([u, np.sin(3.0 * u), np.cos(2.0 * u)], 1)        # (V,9)
    mix = g.normal(0.0, 1.0, (9, 3 * (N_SHAPE + N_EXPR)))
    shapedirs = (basis @ mix).reshape(V, 3, N_SHAPE + N_EXPR) * (ext * 2e-2 / 3.0)
    posedirs = g.normal(0.0, ext * 8e-3, (V, 3, 36))
    joints = c + ext * np.array([[0, -0.85, -0.1], [0, -0.45, -0.1], [0, -0.2, 0.25], [0.27, 0.3, 0.65], [-0.27, 0.3, 0.65]])
    J_regressor = np.zeros((FLAME_J, V))
    weights = np.zeros((V, FLAME_J))
    for j in range(FLAME_J):
        d2 = ((v_template - joints[j]) ** 2).sum(1)
        near = np.argsort(d2)[:50]
        w = np.exp(-d2[near] / (2 * (0.17 * ext) ** 2))
        J_regressor[j, near] = w / w.sum()
        weights[:, j] = np.exp(-d2 / (2 * ((0.5 if j < 3 else 0.12) * ext) ** 2))
    weights[:, 0] += 1e-3
    weights /= weights.sum(1, keepdims=True)
    kintree = np.array([[4294967295, 0, 1, 1, 1], [0, 1, 2, 3, 4]], np.int64)    # row 0 = parents (the reader overwrites entry 0 with -1)
    return dict(v_template=v_template.astype(np.float64), shapedirs=shapedirs.astype(np.float32), posedirs=posedirs.astype(np.float32), J_regressor=J_regressor,
                kintree_table=kintree, weights=weights, f=faces.astype(np.uint32))


def flame_masks_dict(v_template: np.ndarray) -> Dict[str, np.ndarray]:
    """FLAME_masks.pkl's regions (flame_model/flame.py:625-637) cut geometrically from the head's bounding box: every region the
    reference's FlameMask.create_custom_mask combines is present and non-empty, none claims to be anatomically right."""
    lo, hi = v_template.min(0), v_template.max(0)
    u = (v_template - (lo + hi) / 2) / ((hi - lo) / 2)      # [-1,1] per axis: x left/right, y up, z front
    ids = np.arange(v_template.shape[0])
    x, y, z = u[:, 0], u[:, 1], u[:, 2]
    sel = lambda m: ids[m].astype(np.int64)
    eye_l = (np.abs(x - 0.35) < 0.18) & (np.abs(y - 0.25) < 0.14) & (z > 0.3)
    eye_r = (np.abs(x + 0.35) < 0.18) & (np.abs(y - 0.25) < 0.14) & (z > 0.3)
    ball_l = (np.abs(x - 0.35) < 0.08) & (np.abs(y - 0.25) < 0.07) & (z > 0.3)
    ball_r = (np.abs(x + 0.35) < 0.08) & (np.abs(y - 0.25) < 0.07) & (z > 0.3)
    return dict(
        face=sel((z > 0.0) & (y > -0.55) & (y < 0.6)), neck=sel(y < -0.45), scalp=sel((y > 0.45) | ((z < 0.0) & (y > -0.3))),
        boundary=sel(y < -0.92), right_eyeball=sel(ball_r), left_eyeball=sel(ball_l), right_ear=sel((x < -0.8) & (np.abs(y) < 0.3)),
        left_ear=sel((x > 0.8) & (np.abs(y) < 0.3)), forehead=sel((z > 0.2) & (y > 0.4) & (y < 0.75)), eye_region=sel(eye_l | eye_r),
        nose=sel((np.abs(x) < 0.15) & (np.abs(y) < 0.2) & (z > 0.6)), lips=sel((np.abs(x) < 0.3) & (np.abs(y + 0.3) < 0.1) & (z > 0.5)),
        right_eye_region=sel(eye_r), left_eye_region=sel(eye_l))


# bench.py --scene template_like: the splat-scale offset on head_mesh(kind="template_like").  Calibrated with the CPU oracle (tools/template_like_stats.py,
# 100 000 splats, 802x550, timestep 0) against the avatar tools/ref_on_gpu.py stages on the real template (offset 0.55 there):
#                       rect instances   deepest contributor per 8x8 quadrant, p50 / p90 / p99 / max   contributors per pixel
#   staged on template     2.13 M          243 / 547 / 1285 / 3808                                        238
#   template_like, 0.5     2.02 M          303 / 624 / 1503 / 3989                                        286
#   ellipsoid (default)    1.83 M          263 / 570 / 1133 / 2191                                        178
# i.e. a little DEEPER than the staged avatar at 0.95 x its instances (the conservative side for a blend-kernel time).
TEMPLATE_LIKE_LOG_SCALE_OFFSET = 0.5
BENCHMARK_LOG_SCALE_OFFSET = 0.55   # measured with the CPU oracle at 100 000 splats, 802x550: 2.2 M rect instances (bench.py cfg2: 2.23 M); -0.5 gives 0.49 M, 0.8 gives 3.3 M


def write_reference_assets(asset_dir: str, avatar_dir: str, template_obj: str, n_splats: int = FLAME_F, n_frames: int = 4, sh_degree: int = 3,
                           seed: int = 4, benchmark_scale: bool = False, log_scale_offset=None) -> Dict[str, str]:
    """Everything an unchanged `fps_benchmark_demo.py` / `render.py` of the reference opens, in the reference's own formats:
      asset_dir/flame2023.pkl, asset_dir/FLAME_masks.pkl   what FlameHead() unpickles (flame_model/flame.py:37-38,98-129,625-637);
      avatar_dir/point_cloud.ply + avatar_dir/flame_param.npz   a mesh-bound avatar as GaussianModel.save_ply / FlameGaussianModel.save_ply
                                                              write it (scene/gaussian_model.py:253-275, scene/flame_gaussian_model.py:219-224).
    `template_obj` is the checkout's flame_model/assets/flame/head_template_mesh.obj (the pickle's `f` has to equal its faces,
    flame.py:170).  The avatar is bound to the template + teeth topology the reference builds (10144 faces, 5143 vertices)."""
    import os
    import pickle

    from . import io as gio

    v, f = read_obj_topology(template_obj)
    os.makedirs(asset_dir, exist_ok=True)
    os.makedirs(avatar_dir, exist_ok=True)
    out = dict(flame_model=os.path.join(asset_dir, "flame2023.pkl"), flame_masks=os.path.join(asset_dir, "FLAME_masks.pkl"),
               point_cloud=os.path.join(avatar_dir, "point_cloud.ply"), flame_param=os.path.join(avatar_dir, "flame_param.npz"))
    with open(out["flame_model"], "wb") as fh:
        pickle.dump(flame_pickle_dict(v, f, seed), fh, protocol=2)
    with open(out["flame_masks"], "wb") as fh:
        pickle.dump(flame_masks_dict(v), fh, protocol=2)
    arrs = bound_splats(max(n_splats, FLAME_F), FLAME_F, sh_degree, seed=2)
    ext = float((v.max(0) - v.min(0)).max())
    if log_scale_offset is None:
        # default: smaller splats than the benchmark scene (an avatar for STARTING scripts); benchmark_scale: the offset at which this avatar, on the
        # real template's triangles, bins about as many tile instances per frame as bench.py's stand-in on its ellipsoid (tools/ref_on_gpu.py times it)
        log_scale_offset = BENCHMARK_LOG_SCALE_OFFSET if benchmark_scale else -0.5
    arrs["_scaling"] = arrs["_scaling"] + np.float32(log_scale_offset)
    gio.save_ply(out["point_cloud"], arrs)
    seq = flame_sequence(n_frames, seed)
    # the template sits where the FLAME fitting left it (its head around y = 1.5 m): the per-frame translation brings it in front of the
    # benchmark's orbit camera, which looks at the origin (fps_benchmark_demo.py:21-33)
    seq["translation"] = (seq["translation"] * (ext / 0.24) - (v.min(0) + v.max(0)) / 2).astype(np.float32)
    gio.save_flame_param(out["flame_param"], seq)
    return out


def silhouette_image(verts: np.ndarray, cam: SynthCamera, radius: int = 2) -> np.ndarray:
    """(H, W, 4) uint8 RGBA: a flat pseudo-render of a vertex cloud (a disc per projected vertex, coloured smoothly by the vertex's
    position, alpha 255 where covered) -- a target a training script can fit, produced without any rasterizer."""
    H, W = cam.image_height, cam.image_width
    full = cam.full_proj_transform.astype(np.float64)          # (P W2C)^T : row-vector convention
    hom = np.concatenate([verts.astype(np.float64), np.ones((len(verts), 1))], 1) @ full
    ndc = hom[:, :2] / hom[:, 3:4]
    px = ((ndc[:, 0] + 1.0) * W - 1.0) * 0.5                    # ndc2Pix
    py = ((ndc[:, 1] + 1.0) * H - 1.0) * 0.5
    lo, hi = verts.min(0), verts.max(0)
    col = 0.15 + 0.7 * (verts - lo) / np.maximum(hi - lo, 1e-9)
    img = np.zeros((H, W, 4), np.uint8)
    order = np.argsort(-hom[:, 3])                              # far first, near vertices overwrite
    ix, iy = np.rint(px[order]).astype(np.int64), np.rint(py[order]).astype(np.int64)
    c8 = np.rint(col[order] * 255).astype(np.uint8)
    for dy in range(-radius, radius + 1):
        for dx in range(-radius, radius + 1):
            if dx * dx + dy * dy > radius * radius:
                continue
            x, y = ix + dx, iy + dy
            ok = (x >= 0) & (x < W) & (y >= 0) & (y < H)
            img[y[ok], x[ok], :3] = c8[ok]
            img[y[ok], x[ok], 3] = 255
    return img


def write_reference_dataset(data_dir: str, template_obj: str, n_timesteps: int = 4, yaws=(-25.0, 0.0, 25.0), width: int = 112, height: int = 160,
                            seed: int = 4) -> Dict[str, object]:
    """A dataset an unchanged `train.py -s <data_dir> --bind_to_mesh` / `render.py` / `fps_benchmark_dataset.py` of the reference opens
    (scene/__init__.py:80-88 picks the "DynamicNerf" reader on `canonical_flame_param.npz`; scene/dataset_readers.py:189-352):

      data_dir/canonical_flame_param.npz                the marker file (the reference only tests that it exists)
      data_dir/transforms_{train,val,test}.json         {"frames": [{file_path, transform_matrix (camera-to-world, OpenGL axes),
                                                         camera_angle_x, w, h, timestep_index, camera_index, flame_param_path}]}
      data_dir/flame_param/<t>.npz                      per-timestep FLAME parameters, one row each: shape (300,), expr (1,100),
                                                         rotation / neck_pose / jaw_pose / translation (1,3), eyes_pose (1,6),
                                                         static_offset (1,5023,3) (FlameGaussianModel.load_meshes pads it to the
                                                         5143 vertices with teeth, scene/flame_gaussian_model.py:52-60)
      data_dir/images/<t>_<cam>.png                     RGBA targets (CameraDataset.__getitem__ composes them over the background)

    Cameras: the benchmark's orbit camera (fps_benchmark_demo.py:21-33) at the given yaw angles; the last yaw is the validation view of
    every timestep, the last timestep the test split.  The targets are flat pseudo-renders of the posed-by-translation template
    (silhouette_image): no rasterizer is needed to write them, and a training run has something to fit.
    Returns {'timesteps', 'cameras', 'train', 'val', 'test' (frame counts), 'flame_sequence'}."""
    import json
    import os

    v, _f = read_obj_topology(template_obj)
    os.makedirs(os.path.join(data_dir, "flame_param"), exist_ok=True)
    os.makedirs(os.path.join(data_dir, "images"), exist_ok=True)
    seq = flame_sequence(n_timesteps, seed)
    ext = float((v.max(0) - v.min(0)).max())
    seq["translation"] = (seq["translation"] * (ext / 0.24) - (v.min(0) + v.max(0)) / 2).astype(np.float32)   # in front of the orbit camera (as write_reference_assets)
    static_offset = seq["static_offset"][:, : len(v)]            # the template's vertices: the reference pads the teeth
    np.savez(os.path.join(data_dir, "canonical_flame_param.npz"), shape=seq["shape"], static_offset=static_offset,
             **{k: np.zeros_like(seq[k][:1]) for k in ("expr", "rotation", "neck_pose", "jaw_pose", "eyes_pose", "translation")})
    for t in range(n_timesteps):
        np.savez(os.path.join(data_dir, "flame_param", f"{t:05d}.npz"), shape=seq["shape"], static_offset=static_offset,
                 **{k: seq[k][t: t + 1] for k in ("expr", "rotation", "neck_pose", "jaw_pose", "eyes_pose", "translation")})
    splits: Dict[str, list] = dict(train=[], val=[], test=[])
    for t in range(n_timesteps):
        for ci, yaw in enumerate(yaws):
            cam = orbit_camera(width, height, yaw_deg=float(yaw))
            w2c = cam.world_view_transform.T.astype(np.float64)      # OpenCV axes
            c2w = np.linalg.inv(w2c)
            c2w[:3, 1:3] *= -1                                        # back to the OpenGL axes transforms_*.json stores (the reader flips them: dataset_readers.py:205)
            name = f"images/{t:05d}_{ci:02d}"
            from PIL import Image

            Image.fromarray(silhouette_image(v + seq["translation"][t], cam), "RGBA").save(os.path.join(data_dir, name + ".png"))
            frame = dict(file_path=name, transform_matrix=c2w.tolist(), camera_angle_x=float(cam.FoVx), w=int(width), h=int(height),
                         timestep_index=int(t), camera_index=int(ci), flame_param_path=f"flame_param/{t:05d}.npz")
            split = "test" if (t == n_timesteps - 1 and n_timesteps > 1) else ("val" if (ci == len(yaws) - 1 and len(yaws) > 1) else "train")
            splits[split].append(frame)
    for split, frames in splits.items():
        with open(os.path.join(data_dir, f"transforms_{split}.json"), "w") as fh:
            json.dump(dict(frames=frames), fh)
    return dict(timesteps=n_timesteps, cameras=len(yaws), train=len(splits["train"]), val=len(splits["val"]), test=len(splits["test"]), flame_sequence=seq)
