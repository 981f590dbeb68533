"""CPU tests of the on-disk formats: round trips, and the byte layout the reference's
plyfile-based writer produces (scene/gaussian_model.py:236-275)."""
import os

import numpy as np

from gaussianavatars_amd import io as gio
from gaussianavatars_amd import synthetic as S


def test_ply_round_trip_and_layout(tmp_path):
    sp = S.bound_splats(1234, 400, 3, seed=9)
    sp["binding"] = sp["binding"] % 400
    p = str(tmp_path / "point_cloud" / "iteration_1" / "point_cloud.ply")
    gio.save_ply(p, sp)
    back = gio.load_ply(p, sh_degree=3)
    for k in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation"):
        assert back[k].shape == sp[k].shape and back[k].dtype == np.float32
        np.testing.assert_array_equal(back[k], sp[k])
    assert back["binding"].dtype == np.int32
    np.testing.assert_array_equal(back["binding"], sp["binding"].astype(np.int32))
    # layout: header property order and the channel-major f_rest block the reference writes
    head = open(p, "rb").read(4096).split(b"end_header\n")[0].decode()
    props = [l.split()[-1] for l in head.splitlines() if l.startswith("property")]
    assert props == gio.ply_property_names(45, True) and all("property float" in l for l in head.splitlines() if l.startswith("property"))
    t = gio.read_ply_table(p)
    assert t.dtype.itemsize == 4 * len(props)
    # f_rest_k = coefficient (k % 15)+1 of channel k // 15
    np.testing.assert_array_equal(np.asarray(t["f_rest_17"]), sp["_features_rest"][:, 17 % 15, 17 // 15])
    np.testing.assert_array_equal(np.asarray(t["nx"]), 0)


def test_unbound_ply_has_no_binding(tmp_path):
    sp = S.bound_splats(50, 10, 1, seed=1)
    sp.pop("binding")
    p = str(tmp_path / "a.ply")
    gio.save_ply(p, sp)
    back = gio.load_ply(p, sh_degree=1)
    assert "binding" not in back and back["_features_rest"].shape == (50, 3, 3)


def test_flame_param_npz_round_trip(tmp_path):
    seq = S.flame_sequence(5, seed=3)
    p = str(tmp_path / "flame_param.npz")
    gio.save_flame_param(p, seq)
    back = gio.load_flame_param(p)
    assert set(back) == set(gio.FLAME_PARAM_KEYS)
    for k in back:
        np.testing.assert_array_equal(back[k], seq[k])
    assert back["expr"].shape == (5, 100) and back["static_offset"].shape == (1, S.FLAME_V, 3)


def test_morton_order_keeps_every_splat_whole_and_neighbours_close():
    """io.spatial_sort (a loader-side layout choice): a permutation applied to every per-splat array alike -- a bound splat keeps its face --,
    consecutive splats close in space (unbound: by position; bound: by the template centre of their face)."""
    rng = np.random.default_rng(5)
    sp = S.bound_splats(5000, 400, 2, seed=9)
    sp["binding"] = sp["binding"] % 400
    tag = np.arange(5000)
    sp["tag"] = tag                                        # rides along like any per-splat array
    centers = rng.normal(0, 1, (400, 3))
    out = gio.spatial_sort(sp, centers)
    perm = out["tag"]
    assert sorted(perm.tolist()) == list(range(5000))
    for k in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation", "binding"):
        np.testing.assert_array_equal(out[k], np.asarray(sp[k])[perm])
    where = centers[out["binding"]]
    assert np.linalg.norm(np.diff(where, axis=0), axis=1).mean() < 0.2 * np.linalg.norm(np.diff(centers[sp["binding"]], axis=0), axis=1).mean()
    un = dict(_xyz=rng.normal(0, 1, (20000, 3)).astype(np.float32), _opacity=rng.normal(0, 1, (20000, 1)).astype(np.float32))
    o2 = gio.spatial_sort(un)
    assert np.linalg.norm(np.diff(o2["_xyz"], axis=0), axis=1).mean() < 0.15 * np.linalg.norm(np.diff(un["_xyz"], axis=0), axis=1).mean()
    np.testing.assert_array_equal(np.sort(o2["_opacity"], 0), np.sort(un["_opacity"], 0))
    assert list(gio.morton_order(np.array([[0., 0, 0], [1, 1, 1], [0, 0, 1e-3], [1, 1, 0.999]]))) == [0, 2, 3, 1]


def test_load_ply_spatial_sort_on_both_model_classes(tmp_path):
    """GaussianModel.load_ply: the file's splats in Morton order of their positions, a bound model's by the template centre of their faces --
    the DEFAULT since round 4 (GAA_SPATIAL_SORT=0 or spatial_sort=False: row i of the tensors = row i of the file)."""
    import torch

    from gaussianavatars_amd.gaussian_model import FlameGaussianModel, GaussianModel, template_face_centers

    sp = S.bound_splats(S.FLAME_F + 500, S.FLAME_F, 1, seed=3)
    seq = S.flame_sequence(3, seed=4)
    d = tmp_path / "point_cloud" / "iteration_1"
    p = str(d / "point_cloud.ply")
    gio.save_ply(p, sp)
    gio.save_flame_param(str(d / "flame_param.npz"), seq)
    rig = S.flame_rig(seed=4)
    plain = FlameGaussianModel(1, rig, binding_impl="unfused", device="cpu")
    plain.load_ply(p, device="cpu", spatial_sort=False)
    np.testing.assert_array_equal(plain._xyz.detach().numpy(), sp["_xyz"])
    srt = FlameGaussianModel(1, rig, binding_impl="unfused", device="cpu")
    srt.load_ply(p, device="cpu")                       # the default
    env = FlameGaussianModel(1, rig, binding_impl="unfused", device="cpu")
    os.environ["GAA_SPATIAL_SORT"] = "0"
    try:
        env.load_ply(p, device="cpu")                   # the opt-out
    finally:
        del os.environ["GAA_SPATIAL_SORT"]
    np.testing.assert_array_equal(env._xyz.detach().numpy(), sp["_xyz"])
    centers = template_face_centers(srt)
    assert centers.shape == (S.FLAME_F, 3)
    b0, b1 = plain.binding.long().numpy(), srt.binding.long().numpy()
    assert sorted(b0.tolist()) == sorted(b1.tolist()) and not np.array_equal(b0, b1)
    spread = lambda b: np.linalg.norm(np.diff(centers[b], axis=0), axis=1).mean()
    assert spread(b1) < 0.3 * spread(b0)
    # every splat still carries its own data: match rows through (binding, xyz)
    key = lambda m: {(int(b), tuple(np.round(x, 6))) for b, x in zip(m.binding.tolist(), m._xyz.detach().numpy().tolist())}
    assert key(plain) == key(srt)
    assert int(srt.binding_counter.sum()) == len(b1) and torch.equal(srt.binding_counter, plain.binding_counter)
    un = dict(sp)
    un.pop("binding")
    q = str(tmp_path / "u.ply")
    gio.save_ply(q, un)
    g = GaussianModel(1)
    g.load_ply(q, device="cpu")
    x = g._xyz.detach().numpy()
    assert np.linalg.norm(np.diff(x, axis=0), axis=1).mean() < 0.5 * np.linalg.norm(np.diff(sp["_xyz"], axis=0), axis=1).mean()


def test_reference_dataset_layout(tmp_path):
    """synthetic.write_reference_dataset: the "DynamicNerf" layout the reference's reader walks (scene/dataset_readers.py:189-352), checked
    here without the reference: split files, per-frame keys, per-timestep FLAME rows, RGBA targets with a covered silhouette, and the
    camera-to-world matrices in OpenGL axes (the reader flips columns 1 and 2 and inverts)."""
    import json
    import os

    from PIL import Image

    from gaussianavatars_amd import synthetic as S

    obj = tmp_path / "t.obj"
    g = np.random.default_rng(0)
    v = g.normal(0, 0.06, (200, 3)) + np.array([0.0, 1.5, 0.0])
    obj.write_text("".join(f"v {a:.6f} {b:.6f} {c:.6f}\n" for a, b, c in v) + "f 1/1 2/2 3/3\n")
    info = S.write_reference_dataset(str(tmp_path / "d"), str(obj), n_timesteps=3, yaws=(-20.0, 20.0), width=64, height=96)
    d = tmp_path / "d"
    assert (info["train"], info["val"], info["test"]) == (2, 2, 2) and (d / "canonical_flame_param.npz").exists()
    seen = set()
    for split in ("train", "val", "test"):
        frames = json.loads((d / f"transforms_{split}.json").read_text())["frames"]
        for fr in frames:
            assert {"file_path", "transform_matrix", "camera_angle_x", "w", "h", "timestep_index", "camera_index", "flame_param_path"} <= set(fr)
            img = np.asarray(Image.open(d / (fr["file_path"] + ".png")))
            assert img.shape == (96, 64, 4) and 0.01 < (img[..., 3] == 255).mean() < 0.9
            fp = np.load(d / fr["flame_param_path"])
            assert fp["shape"].shape == (300,) and fp["expr"].shape == (1, 100) and fp["eyes_pose"].shape == (1, 6)
            assert fp["static_offset"].shape == (1, 200, 3) and fp["translation"].shape == (1, 3)
            c2w = np.array(fr["transform_matrix"])
            c2w[:3, 1:3] *= -1                                  # what readCamerasFromTransforms does
            w2c = np.linalg.inv(c2w)
            cam = S.orbit_camera(64, 96, yaw_deg=(-20.0, 20.0)[fr["camera_index"]])
            np.testing.assert_allclose(w2c.T, cam.world_view_transform, atol=1e-6)
            seen.add((fr["timestep_index"], fr["camera_index"]))
    assert len(seen) == 6


def test_pillow_compat_serves_the_references_int8_images():
    """scene/__init__.py:51 builds its targets with np.byte data and an explicit mode; shims.pillow_compat keeps that call working on a
    Pillow that type-checks it, and leaves every other call alone."""
    from PIL import Image

    from gaussianavatars_amd import shims

    raw = np.array([[[255, 128, 0], [1, 2, 3]]], np.uint8)
    try:
        Image.fromarray(raw.view(np.int8), "RGB")
        needed = False
    except TypeError:
        needed = True
    assert shims.pillow_compat() == needed
    try:
        got = np.asarray(Image.fromarray(raw.view(np.int8), "RGB"))
        np.testing.assert_array_equal(got, raw)
        np.testing.assert_array_equal(np.asarray(Image.fromarray(raw)), raw)
        assert shims.pillow_compat() is False                   # idempotent
    finally:
        shims.uninstall()
    assert not getattr(Image.fromarray, "__gaussianavatars_amd_shim__", False)
