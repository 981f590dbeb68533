"""The rasterizer's DEFAULT mode -- GsrSettings.fast_blend (include/gsr.h): pre-scaled conic, hardware 2^x, fused products in the two
blend kernels -- against the oracle, with the tolerance BASELINE.json's north_star grants ("pixel-match within a stated fp32
tolerance (and bit-exact tile/key indices)").

Stated tolerance.  Forward: every integer output (radii, rects, tiles_touched, sorted keys, point list, tile ranges) bit-exact; image
within ABS_TOL = 1e-4 of the oracle's at every pixel EXCEPT a counted set of pixels where a record's alpha sits within rounding of
1/255 or the pixel's transmittance within rounding of 1e-4, so that the record is taken by one evaluation and skipped by the other:
those differ by at most (largest colour) / 255 + ABS_TOL and there are at most max(2, 0.1 % of the pixels) of them; `n_contrib`
equals the oracle's except on those pixels' neighbours in the same sense (at most 0.5 %).  Backward: each gradient tensor within
REL_GRAD of the oracle's, relative to the tensor's largest magnitude (the exact kernels' bar is 2e-4; float atomics in both)."""
import math

import numpy as np
import pytest
import torch

from tests.scenes import scene, settings_args
from tests.test_gsr_gpu import _dev, _np

pytestmark = [pytest.mark.gpu, pytest.mark.fast_blend]

ABS_TOL = 1e-4
REL_GRAD = 3e-4
SCENES = ["cfg1", "sh3_small", "sh2_mod", "culls", "dense_tile", "dense_tile_xl", "depth_ties", "deep_stack", "empty_view", "huge_grid"]


def check_image(img, ref, rgb_max, what):
    """-> number of threshold pixels.  img, ref (3,H,W)."""
    d = np.abs(img.astype(np.float64) - ref.astype(np.float64)).max(axis=0)
    flips = d > ABS_TOL
    n = int(flips.sum())
    assert n <= max(2, int(1e-3 * d.size)), f"{what}: {n} of {d.size} pixels beyond {ABS_TOL}"
    bound = max(1.0, float(rgb_max)) / 255.0 * 1.01 + ABS_TOL
    assert d.max() <= bound, f"{what}: a pixel differs by {d.max():.3e} > {bound:.3e} (more than one record at a blend threshold)"
    return n


def _both(oracle, name):
    from gaussianavatars_amd.debug import forward_state
    from gaussianavatars_amd.rasterizer import GaussianRasterizationSettings

    dev = _dev()
    cam, sp, bg, deg, mod = scene(name)
    a = settings_args(cam, bg, deg, mod)
    s = oracle.make_settings(**a)
    st = oracle.forward(s, sp["means3D"], sp["shs"], None, sp["opacities"], sp["scales"], sp["rotations"], None)
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    rs = GaussianRasterizationSettings(a["H"], a["W"], a["tanfovx"], a["tanfovy"], t(a["bg"]), mod, t(a["viewmatrix"]),
                                       t(a["projmatrix"]), deg, t(a["campos"]), False, False)
    hs = forward_state(rs, t(sp["means3D"]), t(sp["shs"]), None, t(sp["opacities"]), t(sp["scales"]), t(sp["rotations"]), None, fast_blend=True)
    return s, st, hs, rs, sp


@pytest.mark.parametrize("name", SCENES)
def test_fast_forward_vs_oracle(oracle, name):
    s, st, hs, rs, sp = _both(oracle, name)
    # integers: exactly the reference's, as in the exact mode
    assert hs["num_rendered"] == st.num_rendered
    np.testing.assert_array_equal(_np(hs["radii"]), st.radii)
    np.testing.assert_array_equal(_np(hs["tiles_touched"]).astype(np.uint32), st.tiles_touched)
    vis = st.radii > 0
    np.testing.assert_array_equal(_np(hs["rect"]).astype(np.int32)[vis], st.rect[vis])
    np.testing.assert_array_equal(_np(hs["keys"]).view(np.uint64), st.keys)
    np.testing.assert_array_equal(_np(hs["point_list"]).astype(np.uint32), st.point_list)
    np.testing.assert_array_equal(_np(hs["ranges"]).astype(np.uint32), st.ranges)
    for k in ("depths", "xy", "rgb", "cov3D"):   # per-splat floats: only the conic slots of the record differ (pre-scaled, see below)
        a, b = _np(hs[k])[vis], getattr(st, k)[vis]
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), k
    co, ref = _np(hs["conic_opacity"])[vis].astype(np.float64), st.conic_opacity[vis].astype(np.float64)
    k2 = -0.5 * math.log2(math.e)
    np.testing.assert_allclose(co[:, [0, 2]], k2 * ref[:, [0, 2]], rtol=3e-7, atol=0)
    np.testing.assert_allclose(co[:, 1], 2 * k2 * ref[:, 1], rtol=3e-7, atol=0)
    assert np.array_equal(co[:, 3], ref[:, 3])
    # image, final_T, n_contrib
    img = _np(hs["color"])
    rgb_max = float(st.rgb[vis].max()) if vis.any() else 1.0
    n = check_image(img, st.color, rgb_max, name)
    fT = np.abs(_np(hs["final_T"]).astype(np.float64) - st.final_T)
    assert (fT > ABS_TOL).sum() <= max(2, int(1e-3 * fT.size)) and fT.max() <= 0.011   # one record (alpha <= 0.99 ... >= 1/255) more or less
    nc = _np(hs["n_contrib"]).astype(np.int64) != st.n_contrib.astype(np.int64)
    assert nc.sum() <= max(2, int(5e-3 * nc.size)), f"{name}: n_contrib differs on {int(nc.sum())} pixels"
    print(f"{name}: {n} threshold pixel(s) of {img[0].size}, image max|diff| {np.abs(img - st.color).max():.2e}, n_contrib differs on {int(nc.sum())}")


@pytest.mark.parametrize("name", ["cfg1", "sh3_small", "sh2_mod", "culls", "dense_tile", "depth_ties", "deep_stack", "huge_grid"])
def test_fast_backward_vs_oracle(oracle, name):
    from gaussianavatars_amd.rasterizer import GaussianRasterizer, set_fast_blend

    assert set_fast_blend(True) == 1, "the product default is the fast blend"
    dev = _dev()
    s, st, hs, rs, sp = _both(oracle, name)
    H, W = rs.image_height, rs.image_width
    gpix = np.random.default_rng(5).normal(0, 1, (3, H, W)).astype(np.float32)
    ref = oracle.backward(s, st, gpix)
    tt = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev).requires_grad_(True)
    m3, sh, op, sc, ro = tt(sp["means3D"]), tt(sp["shs"]), tt(sp["opacities"]), tt(sp["scales"]), tt(sp["rotations"])
    m2 = torch.zeros_like(m3, requires_grad=True)
    color, radii = GaussianRasterizer(rs)(means3D=m3, means2D=m2, shs=sh, opacities=op, scales=sc, rotations=ro)
    (color * torch.from_numpy(gpix).to(dev)).sum().backward()
    rtol = 1e-3 if name == "huge_grid" else REL_GRAD     # (the exact kernels' bars: 1e-3 / 2e-4)
    for k, g in dict(means3D=m3.grad, means2D=m2.grad, shs=sh.grad, opacities=op.grad, scales=sc.grad, rotations=ro.grad).items():
        r = ref[k]
        scale = np.abs(r).max() + 1e-20
        err = np.abs(_np(g).reshape(r.shape) - r).max() / scale
        assert err < rtol, f"{name}/{k}: rel err {err:.3e} (max |ref| {scale:.3e})"


@pytest.mark.parametrize("culling", [1, 4, 6])
@pytest.mark.parametrize("name", ["sh3_small", "deep_stack"])
def test_fast_backward_on_poisoned_state_buffers(oracle, name, culling):
    """Every state buffer of a forward is a fresh torch.empty: with GSR_POISON_STATE they start as 0xFF bytes, so a word a kernel reads
    before another one of the same frame wrote it cannot hide behind a recycled allocation.  Round 3's advisor found one: the per-tile
    backward depth (GsrImageLayout.seg_need) was only zeroed on the rank path, and a stale word with the high bit set made the
    record-parallel backward drop a tile's gradients on the depth-ordered scatter path.  tile_culling 1 = rank path, 4 = depth-ordered
    scatter (production beyond 262 144 splats), 6 = rank path with per-band ranks (large frames), all in the default fast-blend mode."""
    from gaussianavatars_amd.rasterizer import GaussianRasterizer, set_fast_blend, set_poison_state, set_tile_culling, last_forward_info

    assert set_fast_blend(True) == 1
    dev = _dev()
    s, st, hs, rs, sp = _both(oracle, name)
    H, W = rs.image_height, rs.image_width
    gpix = np.random.default_rng(7).normal(0, 1, (3, H, W)).astype(np.float32)
    ref = oracle.backward(s, st, gpix)
    tt = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev).requires_grad_(True)
    prev_c, prev_p = set_tile_culling(culling), set_poison_state(True)
    try:
        m3, sh, op, sc, ro = tt(sp["means3D"]), tt(sp["shs"]), tt(sp["opacities"]), tt(sp["scales"]), tt(sp["rotations"])
        m2 = torch.zeros_like(m3, requires_grad=True)
        color, radii = GaussianRasterizer(rs)(means3D=m3, means2D=m2, shs=sh, opacities=op, scales=sc, rotations=ro)
        path = last_forward_info()["binning_path"]
        (color * torch.from_numpy(gpix).to(dev)).sum().backward()
        torch.cuda.synchronize()
    finally:
        set_tile_culling(prev_c), set_poison_state(prev_p)
    assert path == (1 if culling == 4 else 0)
    np.testing.assert_array_equal(_np(radii), st.radii)
    vis = st.radii > 0
    check_image(_np(color), st.color, float(st.rgb[vis].max()) if vis.any() else 1.0, f"{name}/culling {culling}")
    for k, g in dict(means3D=m3.grad, means2D=m2.grad, shs=sh.grad, opacities=op.grad, scales=sc.grad, rotations=ro.grad).items():
        r = ref[k]
        assert np.isfinite(_np(g)).all(), f"{name}/{k}: poison reached a gradient"
        err = np.abs(_np(g).reshape(r.shape) - r).max() / (np.abs(r).max() + 1e-20)
        assert err < REL_GRAD, f"{name}/culling {culling}/{k}: rel err {err:.3e}"


def test_fast_and_exact_agree_on_the_benchmark_frame(oracle):
    """BASELINE configs[2] (100 k bound splats, 802x550) through render(): the two modes on the same frame -- integers equal, image within
    the stated tolerance, leaf gradients within REL_GRAD of each other."""
    import bench
    from gaussianavatars_amd.gaussian_renderer import render
    from gaussianavatars_amd.rasterizer import set_fast_blend

    dev = _dev()
    g, cam = bench.build_scene(dev, 100_000, 3, 550, 802, 2, "fused", True)
    bg = torch.ones(3, device=dev)
    out = {}
    leaves = (g._xyz, g._features_dc, g._features_rest, g._opacity, g._scaling, g._rotation)
    for fast in (False, True):
        set_fast_blend(fast)
        bench.zero_grads(g)
        g.select_mesh_by_timestep(1)
        pkg = render(cam, g, bench.Pipe, bg)
        (pkg["render"] - 1.0).abs().mean().backward()
        out[fast] = dict(img=_np(pkg["render"]), radii=_np(pkg["radii"]), grads=[_np(p.grad).copy() for p in leaves],
                         flame={k: _np(v.grad).copy() for k, v in g.flame_param.items() if v.grad is not None})
    set_fast_blend(True)
    np.testing.assert_array_equal(out[True]["radii"], out[False]["radii"])
    n = check_image(out[True]["img"], out[False]["img"], 4.0, "cfg3")
    for a, b, nm in zip(out[True]["grads"], out[False]["grads"], ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation")):
        err = np.abs(a - b).max() / (np.abs(b).max() + 1e-20)
        assert err < REL_GRAD, f"{nm}: {err:.3e}"
    # the FLAME rows of the two modes against each other: each within twice its own bar against fp64 (tests/test_fullsize_gpu.py: _flame_row_bars -- a multiple
    # of the composed-torch binding's fp32-vs-fp64 deviation on this frame's gradients; both modes are held to that bar against the oracle in
    # test_config3_benchmarked_step_in_the_benchmarked_mode / the exact-mode tests)
    from tests.test_fullsize_gpu import _flame_row_bars, _leaf_gradients_fp64

    O = oracle

    tfx, tfy = math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5)
    s = O.make_settings(802, 550, tfx, tfy, [1, 1, 1], 1.0, _np(cam.world_view_transform), _np(cam.full_proj_transform), 3, _np(cam.camera_center))
    g.select_mesh_by_timestep(1)
    with torch.no_grad():
        w = {k: _np(v) for k, v in dict(means3D=g.get_xyz, opacities=g.get_opacity, scales=g.get_scaling, rotations=g.get_rotation).items()}
        shs = _np(g.get_features)
    st = O.forward(s, w["means3D"], shs, None, w["opacities"], w["scales"], w["rotations"], None)
    ref = O.backward(s, st, (np.sign(st.color - 1.0) / st.color.size).astype(np.float32))
    bars = _flame_row_bars(g, 1, ref, _leaf_gradients_fp64(g, 1, ref))
    for k in out[False]["flame"]:
        a, b = out[True]["flame"][k], out[False]["flame"][k]
        if k not in bars:
            continue
        err = np.abs(a - b).max() / (np.abs(b).max() + 1e-20)
        print(f"cfg3 fast vs exact d flame {k}: {err:.2e} (bar 2 x {bars[k][1]:.2e})")
        assert err <= 2.0 * bars[k][1], k
    print(f"cfg3: {n} threshold pixel(s), image max|diff| {np.abs(out[True]['img'] - out[False]['img']).max():.2e}")


def test_fast_bound_entry_on_the_rigged_200k_frame_vs_oracle(oracle):
    """BASELINE configs[3] (200 000 splats on the 5143-vertex rig) through what bench.py runs by default: the bound entry in fast-blend mode.
    Image against the oracle (fed the world-space splats of the accessor path) within the stated tolerance, leaf gradients against the
    oracle's world-space gradients carried to the leaves in fp64 (tests/test_fullsize_gpu.py: _leaf_gradients_fp64)."""
    import bench
    from gaussianavatars_amd.gaussian_renderer import l1_loss, render
    from gaussianavatars_amd.rasterizer import set_fast_blend
    from tests.test_fullsize_gpu import _check_flame_rows, _leaf_gradients_fp64

    dev = _dev()
    H, W, N, T, ts = 802, 550, 200_000, 300, 137
    g, cam = bench.build_scene(dev, N, 3, W, H, T, "fused", True)
    bg = torch.ones(3, device=dev)
    target = torch.ones((3, H, W), device=dev)
    tfx, tfy = math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5)
    s = oracle.make_settings(H, W, tfx, tfy, [1, 1, 1], 1.0, _np(cam.world_view_transform), _np(cam.full_proj_transform), 3, _np(cam.camera_center))
    g.select_mesh_by_timestep(ts)
    with torch.no_grad():
        a = {k: _np(v) for k, v in dict(means3D=g.get_xyz, opacities=g.get_opacity, scales=g.get_scaling, rotations=g.get_rotation).items()}
        shs = _np(g.get_features)
    st = oracle.forward(s, a["means3D"], shs, None, a["opacities"], a["scales"], a["rotations"], None)
    ref = oracle.backward(s, st, (np.sign(st.color - 1.0) / st.color.size).astype(np.float32))
    want = _leaf_gradients_fp64(g, ts, ref)
    assert set_fast_blend(True) == 1
    g.bound_render = True
    bench.zero_grads(g)
    g.select_mesh_by_timestep(ts)
    pkg = render(cam, g, bench.Pipe, bg)
    l1_loss(pkg["render"], target).backward()
    np.testing.assert_array_equal(_np(pkg["radii"]), st.radii)
    n = check_image(_np(pkg["render"]), st.color, float(st.rgb[st.radii > 0].max()), "cfg4")
    got = dict(_xyz=g._xyz.grad, _scaling=g._scaling.grad, _rotation=g._rotation.grad, _opacity=g._opacity.grad,
               _features_dc=g._features_dc.grad, _features_rest=g._features_rest.grad)
    for k, v in got.items():
        r = np.asarray(want[k], np.float64).reshape(tuple(v.shape))
        err = np.abs(_np(v).astype(np.float64) - r).max() / (np.abs(r).max() + 1e-30)
        assert err < 5e-4, f"bound entry, fast blend: d{k} rel err {err:.2e}"
    _check_flame_rows(g, ts, ref, want, "cfg4 fast blend")
    print(f"cfg4 fast blend: {n} threshold pixel(s)")
