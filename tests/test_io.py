"""CPU tests of the on-disk formats: round trips, and the byte layout the reference's
plyfile-based writer produces (scene/gaussian_model.py:236-275)."""
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
