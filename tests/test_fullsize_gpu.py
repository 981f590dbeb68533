"""GPU parity at BASELINE.json's full sizes.

config 2/3 (100 000 mesh-bound SH-3 splats, 802x550): the oracle still finishes in seconds, so the
forward is compared bit for bit and the backward within tolerance, in addition to the size-independent
properties.  config 5 (2 000 000 SH-3 splats, 1600x1100, forward only): sortedness / range / count
invariants, run-to-run bit-stability, and a bit-exact image against the oracle.
"""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _np(x):
    return x.detach().cpu().numpy()


def _invariants(hs, H, W):
    I = hs["num_rendered"]
    keys = _np(hs["keys"]).view(np.uint64)
    pl = _np(hs["point_list"]).astype(np.int64)
    assert int(_np(hs["tiles_touched"]).astype(np.int64).sum()) == I
    if I:
        d = np.diff(keys.astype(np.uint64))
        assert (keys[1:] >= keys[:-1]).all()                      # sorted by (tile, depth)
        same = d == 0
        assert (np.diff(pl)[same] > 0).all()                      # stable: ties keep ascending splat index
    rng = _np(hs["ranges"]).astype(np.int64)
    cnt = rng[:, 1] - rng[:, 0]
    assert cnt.sum() == I and (cnt >= 0).all()
    tiles_of_keys = (keys >> np.uint64(32)).astype(np.int64)
    nz = np.nonzero(cnt)[0]
    assert (tiles_of_keys[rng[nz, 0]] == nz).all() and (tiles_of_keys[rng[nz, 1] - 1] == nz).all()
    gx = (W + 15) // 16
    ys, xs = np.mgrid[0:H, 0:W]
    tile_of_px = (ys // 16) * gx + xs // 16
    assert (_np(hs["n_contrib"]).astype(np.int64) <= cnt[tile_of_px]).all()
    fT = _np(hs["final_T"])
    assert np.isfinite(fT).all() and (fT >= 0).all() and (fT <= 1).all()
    assert np.isfinite(_np(hs["color"])).all()
    assert (_np(hs["qcount"]).astype(np.int64) <= cnt[:, None]).all()


def test_config2_100k_bound_forward_backward_vs_oracle(oracle):
    import bench
    from gaussianavatars_amd.debug import forward_state
    from gaussianavatars_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer

    dev = _dev()
    H, W = 802, 550
    g, cam = bench.build_scene(dev, 100_000, 3, W, H, 4, "fused", False)
    with torch.no_grad():
        g.select_mesh_by_timestep(1)
        ins = dict(means3D=g.get_xyz, shs=g.get_features, opacities=g.get_opacity, scales=g.get_scaling, rotations=g.get_rotation)
        ins = {k: v.detach().clone() for k, v in ins.items()}
    tfx, tfy = math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5)
    bg = torch.ones(3, device=dev)
    rs = GaussianRasterizationSettings(H, W, tfx, tfy, bg, 1.0, cam.world_view_transform, cam.full_proj_transform, 3,
                                       cam.camera_center, False, False)
    hs = forward_state(rs, ins["means3D"], ins["shs"], None, ins["opacities"], ins["scales"], ins["rotations"], None)
    _invariants(hs, H, W)
    hs2 = forward_state(rs, ins["means3D"], ins["shs"], None, ins["opacities"], ins["scales"], ins["rotations"], None)
    assert torch.equal(hs["color"], hs2["color"]) and torch.equal(hs["keys"], hs2["keys"])       # forward is deterministic
    # full-size oracle
    s = oracle.make_settings(H, W, tfx, tfy, [1, 1, 1], 1.0, _np(cam.world_view_transform), _np(cam.full_proj_transform), 3,
                             _np(cam.camera_center))
    a = {k: _np(v) for k, v in ins.items()}
    st = oracle.forward(s, a["means3D"], a["shs"], None, a["opacities"], a["scales"], a["rotations"], None)
    assert st.num_rendered == hs["num_rendered"]
    np.testing.assert_array_equal(_np(hs["radii"]), st.radii)
    np.testing.assert_array_equal(_np(hs["keys"]).view(np.uint64), st.keys)
    np.testing.assert_array_equal(_np(hs["point_list"]).astype(np.uint32), st.point_list)
    np.testing.assert_array_equal(_np(hs["n_contrib"]).astype(np.uint32), st.n_contrib)
    assert np.array_equal(_np(hs["color"]).view(np.uint32), st.color.view(np.uint32))
    # backward: L1 vs white (config 3) against the oracle
    t = {k: v.clone().requires_grad_(True) for k, v in ins.items()}
    m2 = torch.zeros_like(t["means3D"], requires_grad=True)
    color, _ = GaussianRasterizer(rs)(means3D=t["means3D"], means2D=m2, shs=t["shs"], opacities=t["opacities"], scales=t["scales"],
                                      rotations=t["rotations"])
    # production mode (tile_culling = 1, what render() and bench.py run): the image is the oracle's, bit for bit
    assert np.array_equal(_np(color).view(np.uint32), st.color.view(np.uint32))
    (color - 1.0).abs().mean().backward()
    gpix = (np.sign(st.color - 1.0) / st.color.size).astype(np.float32)
    ref = oracle.backward(s, st, gpix)
    for k, got in (("means3D", t["means3D"].grad), ("means2D", m2.grad), ("shs", t["shs"].grad), ("opacities", t["opacities"].grad),
                   ("scales", t["scales"].grad), ("rotations", t["rotations"].grad)):
        r = ref[k]
        err = np.abs(_np(got).reshape(r.shape) - r).max() / (np.abs(r).max() + 1e-30)
        assert err < 5e-4, f"{k}: rel err {err:.2e}"
    # linearity of the backward in dL/dpixel
    t2 = {k: v.clone().requires_grad_(True) for k, v in ins.items()}
    color2, _ = GaussianRasterizer(rs)(means3D=t2["means3D"], means2D=torch.zeros_like(m2, requires_grad=True), shs=t2["shs"],
                                       opacities=t2["opacities"], scales=t2["scales"], rotations=t2["rotations"])
    (2.0 * (color2 - 1.0).abs().mean()).backward()
    err = (t2["means3D"].grad - 2 * t["means3D"].grad).abs().max() / (2 * t["means3D"].grad.abs().max())
    assert float(err) < 1e-4


def _leaf_gradients(g, ts, ref, dtype=torch.float64):
    """The oracle's world-space gradients `ref` (means3D, scales, rotations, opacities, shs) carried to the model's leaves and to row `ts`
    of the FLAME tables by the composed-torch binding (gaussianavatars_amd/unfused.py: the reference's scene/gaussian_model.py:113-150,
    utils/graphics_utils.py:116-135, flame_model/lbs.py restated and pinned to the reference's classes by tests/test_model_pins.py) in `dtype`."""
    from gaussianavatars_amd import unfused as U

    dev = g._xyz.device
    d = lambda t: t.detach().to(dtype)
    fm = g.flame_model
    rig = {k: d(getattr(fm, k)) for k in ("v_template", "shapedirs", "posedirs", "J_regressor", "lbs_weights")}
    rig["parents"] = fm.parents
    fp = g.flame_param
    rows = {k: d(fp[k][[ts]]).requires_grad_(True) for k in ("expr", "rotation", "neck_pose", "jaw_pose", "eyes_pose", "translation")}
    leaves = {k: d(getattr(g, k)).requires_grad_(True) for k in ("_xyz", "_scaling", "_rotation", "_opacity")}
    verts, _ = U.flame_forward(rig, d(fp["shape"])[None], rows["expr"], rows["rotation"], rows["neck_pose"], rows["jaw_pose"], rows["eyes_pose"],
                               rows["translation"], d(fp["static_offset"]))
    c, R, sc, q = U.face_frames(verts[0], fm.faces)
    world = [U.bind_xyz(leaves["_xyz"], g.binding, R, sc, c), U.bind_scaling(leaves["_scaling"], g.binding, sc),
             U.bind_rotation(leaves["_rotation"], g.binding, q), torch.sigmoid(leaves["_opacity"])]
    grads = [torch.as_tensor(ref[k], dtype=dtype, device=dev).reshape(w.shape) for k, w in zip(("means3D", "scales", "rotations", "opacities"), world)]
    torch.autograd.backward(world, grads)
    out = {k: v.grad.cpu().numpy() for k, v in leaves.items()}
    out.update({"flame_" + k: v.grad[0].cpu().numpy() for k, v in rows.items()})
    shs = np.asarray(ref["shs"], np.float64)
    out["_features_dc"], out["_features_rest"] = shs[:, :1], shs[:, 1:]
    return out


def _leaf_gradients_fp64(g, ts, ref):
    return _leaf_gradients(g, ts, ref, torch.float64)


FLAME_ROWS = ("expr", "rotation", "neck_pose", "jaw_pose", "eyes_pose", "translation")
FLAME_ROW_FLOOR = 3e-5   # the floor of a FLAME-row bar (rounds 3 - 5 held these rows to a flat 2e-3).  Measured on the MI355X, round 6: errors 2e-7 .. 9e-6, the composed-torch binding's own fp32 deviation 1e-8 .. 2e-5
FLAME_ROW_FACTOR = 6.0   # ... and otherwise this multiple of the composed-torch binding's OWN fp32-vs-fp64 deviation on the same gradients (as tests/test_model_pins.py:88-113)


def _flame_row_bars(g, ts, ref, want64):
    """{row: (deviation, bar)}: a FLAME row's gradient is a sum over every splat of terms that largely cancel (the orientation and scale parts of a
    rigid motion), so ANY fp32 evaluation is a draw around the fp64 value.  The yardstick is how far the reference-shaped composed-torch binding lands from
    its own fp64 result when it is run in fp32 on the SAME world-space gradients; the bar is FLAME_ROW_FACTOR times that, never below FLAME_ROW_FLOOR."""
    w32 = _leaf_gradients(g, ts, ref, torch.float32)
    out = {}
    for k in FLAME_ROWS:
        r = np.asarray(want64["flame_" + k], np.float64)
        dev = float(np.abs(np.asarray(w32["flame_" + k], np.float64) - r).max() / (np.abs(r).max() + 1e-30))
        out[k] = (dev, max(FLAME_ROW_FLOOR, FLAME_ROW_FACTOR * dev))
    return out


def _check_flame_rows(g, ts, ref, want64, what):
    """The FLAME-row gradients of the model against `want64` (the oracle carried to row ts in fp64), each held to its own bar; the measured errors are printed."""
    bars = _flame_row_bars(g, ts, ref, want64)
    for k in FLAME_ROWS:
        r = np.asarray(want64["flame_" + k], np.float64)
        err = float(np.abs(_np(g.flame_param[k].grad[ts]).astype(np.float64) - r).max() / (np.abs(r).max() + 1e-30))
        dev, bar = bars[k]
        print(f"{what} d flame {k}: err {err:.2e}  composed-torch fp32 deviation {dev:.2e}  bar {bar:.2e}")
        assert err < bar, f"{what} d flame {k}: rel err {err:.2e} (bar {bar:.2e} = max({FLAME_ROW_FLOOR:g}, {FLAME_ROW_FACTOR:g} x {dev:.2e}))"


@pytest.mark.parametrize("head", ["ellipsoid", "template_like"])
def test_config3_benchmarked_step_in_the_benchmarked_mode(oracle, head):
    """(`head`: the two scenes bench.py reports -- the ellipsoid stand-in of the default line and the head with the reference template's face-area
    distribution, whose tiles are as deep as the avatar staged on the real template.)
    BASELINE configs[2], the step bench.py times, in the mode it times it: 100 000 Morton-ordered mesh-bound splats through select_mesh_by_timestep ->
    render() -> l1_loss -> backward() with the PRODUCT defaults -- fast blend, tile culling, the bound entry, the compiled host (asserted from
    last_forward_info) -- against the oracle directly: image within the fast blend's stated tolerance, radii equal, the six leaf gradients and the
    screen-space gradient within 5e-4 of the oracle's world-space gradients carried to the leaves in fp64, every FLAME row within its own bar."""
    import bench
    from gaussianavatars_amd import _host
    from gaussianavatars_amd import rasterizer as R
    from gaussianavatars_amd.gaussian_renderer import l1_loss, render
    from tests.test_fast_blend_gpu import check_image

    dev = _dev()
    H, W, N, T = 802, 550, 100_000, 8
    assert bench.SPATIAL_SORT
    prev_scene, bench.SCENE = bench.SCENE, head
    try:
        g, cam = bench.build_scene(dev, N, 3, W, H, T, "fused", True)
    finally:
        bench.SCENE = prev_scene
    bg = torch.ones(3, device=dev)
    target = torch.ones((3, H, W), device=dev)
    tfx, tfy = math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5)
    s = oracle.make_settings(H, W, tfx, tfy, [1, 1, 1], 1.0, _np(cam.world_view_transform), _np(cam.full_proj_transform), 3, _np(cam.camera_center))
    prev = R.set_fast_blend(True)
    try:
        for ts in ((0, 5) if head == "ellipsoid" else (3,)):
            g.select_mesh_by_timestep(ts)
            with torch.no_grad():
                a = {k: _np(v) for k, v in dict(means3D=g.get_xyz, opacities=g.get_opacity, scales=g.get_scaling, rotations=g.get_rotation).items()}
                shs = _np(g.get_features)
            st = oracle.forward(s, a["means3D"], shs, None, a["opacities"], a["scales"], a["rotations"], None)
            ref = oracle.backward(s, st, (np.sign(st.color - 1.0) / st.color.size).astype(np.float32))
            want = _leaf_gradients_fp64(g, ts, ref)
            want["means2D"] = ref["means2D"]
            # the benchmarked step (bench.one_step)
            bench.zero_grads(g)
            g.select_mesh_by_timestep(ts)
            pkg = render(cam, g, bench.Pipe, bg)
            l1_loss(pkg["render"], target).backward()
            info = R.last_forward_info()
            assert info.get("bound") and info.get("tile_culling") and not info.get("forward_only"), info
            assert bool(info.get("native_host")) == (_host.get() is not None), info
            np.testing.assert_array_equal(_np(pkg["radii"]), st.radii)
            n = check_image(_np(pkg["render"]), st.color, float(st.rgb[st.radii > 0].max()), f"cfg3 {head} t={ts}")
            got = dict(_xyz=g._xyz.grad, _scaling=g._scaling.grad, _rotation=g._rotation.grad, _opacity=g._opacity.grad,
                       _features_dc=g._features_dc.grad, _features_rest=g._features_rest.grad, means2D=pkg["viewspace_points"].grad)
            for k, v in got.items():
                r = np.asarray(want[k], np.float64).reshape(tuple(v.shape))
                err = np.abs(_np(v).astype(np.float64) - r).max() / (np.abs(r).max() + 1e-30)
                assert err < 5e-4, f"cfg3 {head} t={ts} benchmarked mode d{k}: rel err {err:.2e}"
            _check_flame_rows(g, ts, ref, want, f"cfg3 {head} t={ts} benchmarked mode")
            print(f"cfg3 {head} t={ts} benchmarked mode ({'compiled' if info.get('native_host') else 'python'} host): {n} threshold pixel(s), "
                  f"image max|diff| {np.abs(_np(pkg['render']) - st.color).max():.2e}")
    finally:
        R.set_fast_blend(prev)


def test_config4_200k_rigged_sequence_vs_oracle(oracle):
    """BASELINE configs[3]: 200 000 splats bound to the 5143-vertex synthetic FLAME rig, 300-frame expression sequence.
    Three timesteps through the model path the benchmark runs (select_mesh_by_timestep -> render -> L1 vs white -> backward,
    production mode): image bit-exact against the oracle fed the SAME world-space splats, screen-space and per-splat
    gradients within 5e-4 of each tensor's max (the oracle sums in double)."""
    import bench
    from gaussianavatars_amd.gaussian_renderer import l1_loss, render

    dev = _dev()
    H, W, N, T = 802, 550, 200_000, 300
    g, cam = bench.build_scene(dev, N, 3, W, H, T, "fused", True)
    bg = torch.ones(3, device=dev)
    target = torch.ones((3, H, W), device=dev)
    tfx, tfy = math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5)
    s = oracle.make_settings(H, W, tfx, tfy, [1, 1, 1], 1.0, _np(cam.world_view_transform), _np(cam.full_proj_transform), 3,
                             _np(cam.camera_center))
    for ts in (0, 137, 299):
        # what the benchmark runs: the rasterizer's bound entry on the model's leaves (no world-space tensors); same image bits
        g.bound_render = True
        g.select_mesh_by_timestep(ts)
        with torch.no_grad():
            img_bound = _np(render(cam, g, bench.Pipe, bg)["render"])
        # the world-space gradients the oracle is compared on only exist on the accessor path
        g.bound_render = False
        bench.zero_grads(g)
        g.select_mesh_by_timestep(ts)
        world = dict(means3D=g.get_xyz, opacities=g.get_opacity, scales=g.get_scaling, rotations=g.get_rotation)
        for v in world.values():
            v.retain_grad()                      # what the rasterizer's backward hands to the binding backward
        pkg = render(cam, g, bench.Pipe, bg)
        loss = l1_loss(pkg["render"], target)
        loss.backward()
        a = {k: _np(v) for k, v in world.items()}
        shs = _np(g.get_features)
        st = oracle.forward(s, a["means3D"], shs, None, a["opacities"], a["scales"], a["rotations"], None)
        img = _np(pkg["render"])
        assert np.array_equal(img.view(np.uint32), st.color.view(np.uint32)), f"t={ts}: image max abs diff {np.abs(img - st.color).max()}"
        assert np.array_equal(img_bound.view(np.uint32), st.color.view(np.uint32)), f"t={ts}: bound entry: image max abs diff {np.abs(img_bound - st.color).max()}"
        np.testing.assert_array_equal(_np(pkg["radii"]), st.radii)
        np.testing.assert_array_equal(_np(pkg["visibility_filter"]), st.radii > 0)
        gpix = (np.sign(st.color - 1.0) / st.color.size).astype(np.float32)
        ref = oracle.backward(s, st, gpix)
        got = dict(means3D=world["means3D"].grad, means2D=pkg["viewspace_points"].grad, opacities=world["opacities"].grad,
                   scales=world["scales"].grad, rotations=world["rotations"].grad,
                   shs=torch.cat([g._features_dc.grad, g._features_rest.grad], 1))
        for k, v in got.items():
            r = ref[k]
            err = np.abs(_np(v).reshape(r.shape) - r).max() / (np.abs(r).max() + 1e-30)
            assert err < 5e-4, f"t={ts} {k}: rel err {err:.2e}"
        for k in ("expr", "jaw_pose", "rotation", "translation"):
            gk = g.flame_param[k].grad
            other = torch.ones(gk.shape[0], dtype=torch.bool, device=gk.device)
            other[ts] = False
            assert float(gk[ts].abs().max()) > 0 and float(gk[other].abs().max()) == 0.0   # row ts only
        # ---- what the benchmark actually runs, gradients included: the BOUND entry (leaves in, leaf gradients out, no world-space tensors).
        # Anchor: the oracle's world-space gradients pushed through the binding in fp64 -- the composed-torch statement of
        # scene/gaussian_model.py:113-150, utils/graphics_utils.py:116-135 and flame_model/lbs.py (gaussianavatars_amd/unfused.py, itself
        # pinned to the reference's classes by tests/test_model_pins.py) under torch autograd.
        want = _leaf_gradients_fp64(g, ts, ref)
        g.bound_render = True
        bench.zero_grads(g)
        g.select_mesh_by_timestep(ts)
        pkg = render(cam, g, bench.Pipe, bg)
        l1_loss(pkg["render"], target).backward()
        got = dict(_xyz=g._xyz.grad, _scaling=g._scaling.grad, _rotation=g._rotation.grad, _opacity=g._opacity.grad,
                   _features_dc=g._features_dc.grad, _features_rest=g._features_rest.grad, means2D=pkg["viewspace_points"].grad)
        want["means2D"] = ref["means2D"]
        for k, v in got.items():
            r = np.asarray(want[k], np.float64).reshape(tuple(v.shape))
            err = np.abs(_np(v).astype(np.float64) - r).max() / (np.abs(r).max() + 1e-30)
            assert err < 5e-4, f"t={ts} bound entry d{k}: rel err {err:.2e}"
        _check_flame_rows(g, ts, ref, want, f"cfg4 t={ts} bound entry")   # (sums over 200 k splats of cancelling terms: each row's bar is a multiple of the composed-torch binding's own fp32 noise)


def test_config5_2m_stress_forward(oracle):
    from gaussianavatars_amd import synthetic as S
    from gaussianavatars_amd.debug import forward_state
    from gaussianavatars_amd.rasterizer import GaussianRasterizationSettings

    dev = _dev()
    H, W, N = 1100, 1600, 2_000_000
    cam = S.orbit_camera(W, H)
    sp = S.random_splats(N, 3, 5, xyz_sigma=0.08, log_scale_mean=math.log(0.0015), log_scale_sigma=0.4)
    sp["rotations"] /= np.linalg.norm(sp["rotations"], axis=1, keepdims=True)   # unit quaternions: I/N ~ 2-4 as the config asks
    tfx, tfy = math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    rs = GaussianRasterizationSettings(H, W, tfx, tfy, torch.ones(3, device=dev), 1.0, t(cam.world_view_transform),
                                       t(cam.full_proj_transform), 3, t(cam.camera_center), False, False)
    args = (t(sp["means3D"]), t(sp["shs"]), None, t(sp["opacities"]), t(sp["scales"]), t(sp["rotations"]), None)
    hs = forward_state(rs, *args)
    assert 1.0 < hs["num_rendered"] / N < 12.0
    _invariants(hs, H, W)
    hs2 = forward_state(rs, *args)
    assert torch.equal(hs["color"], hs2["color"]) and torch.equal(hs["point_list"], hs2["point_list"])
    s = oracle.make_settings(H, W, tfx, tfy, [1, 1, 1], 1.0, cam.world_view_transform, cam.full_proj_transform, 3, cam.camera_center)
    st = oracle.forward(s, sp["means3D"], sp["shs"], None, sp["opacities"], sp["scales"], sp["rotations"], None)
    assert st.num_rendered == hs["num_rendered"]
    np.testing.assert_array_equal(_np(hs["point_list"]).astype(np.uint32), st.point_list)
    assert np.array_equal(_np(hs["color"]).view(np.uint32), st.color.view(np.uint32))


def test_second_backward_through_the_same_graph():
    """retain_graph: the gradient accumulators live in the saved state (GsrGeomLayout.acc, the FLAME workspace) and each
    backward leaves them zeroed, so a second backward through the same forward reproduces the first one's gradients."""
    import bench
    from gaussianavatars_amd.gaussian_renderer import l1_loss, render

    dev = _dev()
    g, cam = bench.build_scene(dev, 30_000, 3, 320, 256, 2, "fused", True)
    bg = torch.ones(3, device=dev)
    target = torch.full((3, cam.image_height, cam.image_width), 0.3, device=dev)
    g.select_mesh_by_timestep(1)
    pkg = render(cam, g, bench.Pipe, bg)
    loss = l1_loss(pkg["render"], target)
    params = [g._xyz, g._features_dc, g._features_rest, g._scaling, g._rotation, g._opacity, g.flame_param["expr"],
              g.flame_param["jaw_pose"], g.flame_param["rotation"], g.flame_param["translation"]]
    first = torch.autograd.grad(loss, params, retain_graph=True)
    second = torch.autograd.grad(loss, params)
    for a, b, p in zip(first, second, params):
        scale = float(a.abs().max()) + 1e-30
        assert scale > 1e-12
        assert float((a - b).abs().max()) / scale < 2e-4   # float atomics reorder the sums between the two runs


def test_non_default_stream_and_interleaved_forwards():
    """Everything is enqueued on torch's CURRENT stream (no hidden default-stream work, no device-wide sync), and several
    forwards may precede a backward (state lives in the buffers the graph holds): two frames rendered on a side stream,
    backwarded in reverse order, must equal the same frames done one at a time on the default stream."""
    import bench
    from gaussianavatars_amd.gaussian_renderer import l1_loss, render

    dev = _dev()
    g, cam = bench.build_scene(dev, 20_000, 3, 256, 192, 3, "fused", True)
    bg = torch.ones(3, device=dev)
    target = torch.full((3, cam.image_height, cam.image_width), 0.25, device=dev)
    params = [g._xyz, g._features_dc, g._features_rest, g._scaling, g._rotation, g._opacity]

    def frame(t):
        g.select_mesh_by_timestep(t)
        pkg = render(cam, g, bench.Pipe, bg)
        return pkg["render"], l1_loss(pkg["render"], target)

    ref = []
    for t in (0, 2):
        img, loss = frame(t)
        ref.append((img.detach().clone(), [x.clone() for x in torch.autograd.grad(loss, params)]))
    torch.cuda.synchronize()

    if hasattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch"):
        torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)   # the leaves were first used on the default stream
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        img0, loss0 = frame(0)
        img2, loss2 = frame(2)              # second forward before any backward
        g2 = torch.autograd.grad(loss2, params)
        g0 = torch.autograd.grad(loss0, params)
    side.synchronize()
    for (img_ref, grads_ref), img, grads in ((ref[0], img0, g0), (ref[1], img2, g2)):
        assert torch.equal(img_ref, img)    # the forward is bit-reproducible
        for a, b in zip(grads_ref, grads):
            scale = float(a.abs().max()) + 1e-30
            assert float((a - b).abs().max()) / scale < 2e-4


# ---- the frames bench.py times, in the mode it times them (round 5) ---------------------------------------------------------------------
# Everything above runs the EXACT blend (tests/conftest.py) and, for the 2 M-splat frame, the parity binning of debug.forward_state on a
# landscape image.  bench.py --workload cfg5 / cfg2 run render() under torch.no_grad(): fast blend, forward_only, default tile culling (the
# 20-band rank path at 2 M splats), Morton-ordered leaves through the leaves / bound entry.  The two tests below compare exactly those frames
# with the oracle fed the SAME world-space splats, under the tolerance tests/test_fast_blend_gpu.py states (shared constants), and then the
# exact blend on the same frame bit for bit.
def _kernel_activations(g):
    """World-space (means3D, scales, rotations, opacities) of an UNBOUND model with the bits the rasterizer's first kernel computes in place:
    csrc/bind_math.h is shared by gab::k_bind and gsr::k_preprocess, and binding every splat to ONE identity face frame (R = I, s = 1, c = 0,
    q = (1,0,0,0)) makes k_bind's bound transform the unbound activations exactly (x*1+0, exp(ls)*1, 1 (x) normalize(q), sigmoid)."""
    from gaussianavatars_amd import binding as fused

    dev = g._xyz.device
    P = g._xyz.shape[0]
    eye = torch.eye(3, device=dev)[None].contiguous()
    one, zero3 = torch.ones((1, 1), device=dev), torch.zeros((1, 3), device=dev)
    quat = torch.tensor([[1.0, 0.0, 0.0, 0.0]], device=dev)
    with torch.no_grad():
        xyz, sc, rot, op = fused.bind_splats(g._xyz.detach(), g._scaling.detach(), g._rotation.detach(), torch.zeros(P, dtype=torch.int32, device=dev),
                                             eye, one, zero3, quat, opacity_logit=g._opacity.detach())
    assert torch.equal(xyz, g._xyz.detach())
    return xyz, sc, rot, op


@pytest.mark.fast_blend
def test_config5_benchmarked_frame_in_the_benchmarked_mode(oracle):
    """BASELINE configs[4] exactly as bench.py --workload cfg5 runs it: H = 1600, W = 1100 (portrait, 6900 tiles), 2 000 000 Morton-ordered
    SH-3 splats of an unbound GaussianModel, render() under torch.no_grad()."""
    import bench
    from gaussianavatars_amd import rasterizer as R
    from gaussianavatars_amd.gaussian_renderer import render
    from tests.test_fast_blend_gpu import check_image

    dev = _dev()
    H, W, N = 1600, 1100, 2_000_000
    assert bench.SPATIAL_SORT
    g, cam = bench.build_unbound_scene(dev, N, 3, W, H)
    bg = torch.ones(3, device=dev)
    assert R.get_tile_culling() == 1 and R._fast_blend == 1          # the product defaults, nothing selected by the test
    with torch.no_grad():
        pkg = render(cam, g, bench.Pipe, bg)
    info = R.last_forward_info()
    assert info["forward_only"] is True and info["bound"] is True and info["tile_culling"] is True
    assert info["binning_path"] == 0 and info["rank_bands"] == 20, info     # the rank path over 20 bands of tile rows: what the 2 M-splat line times
    xyz, sc, rot, op = _kernel_activations(g)
    tfx, tfy = math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5)
    s = oracle.make_settings(H, W, tfx, tfy, [1, 1, 1], 1.0, _np(cam.world_view_transform), _np(cam.full_proj_transform), 3, _np(cam.camera_center))
    shs = _np(g.get_features)
    st = oracle.forward(s, _np(xyz), shs, None, _np(op), _np(sc), _np(rot), None)
    assert info["rect_instances"] == st.num_rendered                  # the reference's instance count; the culled lists hold fewer
    assert 0 < info["num_rendered"] <= st.num_rendered
    np.testing.assert_array_equal(_np(pkg["radii"]), st.radii)
    np.testing.assert_array_equal(_np(pkg["visibility_filter"]), st.radii > 0)
    n = check_image(_np(pkg["render"]), st.color, float(shs.max()) + 0.5, "cfg5 portrait, fast blend")
    # the same frame, exact blend: the oracle's bits
    prev = R.set_fast_blend(False)
    try:
        with torch.no_grad():
            exact = render(cam, g, bench.Pipe, bg)
    finally:
        R.set_fast_blend(prev)
    assert R.last_forward_info()["rank_bands"] == 20
    assert np.array_equal(_np(exact["render"]).view(np.uint32), st.color.view(np.uint32)), f"exact blend: max |diff| {np.abs(_np(exact['render']) - st.color).max()}"
    assert torch.equal(exact["radii"], pkg["radii"])
    print(f"cfg5 product mode: {n} threshold pixel(s) of {H * W}; rect instances {st.num_rendered}, binned {info['num_rendered']}")


@pytest.mark.fast_blend
def test_config2_benchmarked_frame_in_the_benchmarked_mode(oracle):
    """BASELINE configs[1] as bench.py --workload cfg2 / fps_benchmark_demo.py:59-61 run it: 100 000 Morton-ordered mesh-bound SH-3 splats,
    802x550, select_mesh_by_timestep + render() under torch.no_grad() through the BOUND entry (fast blend, forward_only)."""
    import bench
    from gaussianavatars_amd import rasterizer as R
    from gaussianavatars_amd.gaussian_renderer import render
    from tests.test_fast_blend_gpu import check_image

    dev = _dev()
    H, W, N = 802, 550, 100_000
    g, cam = bench.build_scene(dev, N, 3, W, H, 300, "fused", False)
    bg = torch.ones(3, device=dev)
    tfx, tfy = math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5)
    s = oracle.make_settings(H, W, tfx, tfy, [1, 1, 1], 1.0, _np(cam.world_view_transform), _np(cam.full_proj_transform), 3, _np(cam.camera_center))
    shs = _np(g.get_features)
    assert R.get_tile_culling() == 1 and R._fast_blend == 1
    for ts in (0, 151):
        with torch.no_grad():
            g.bound_render = True
            g.select_mesh_by_timestep(ts)
            pkg = render(cam, g, bench.Pipe, bg)
            info = R.last_forward_info()
            assert info["forward_only"] is True and info["bound"] is True and info["tile_culling"] is True
            g.bound_render = False          # the accessor path: the world-space tensors the oracle is fed (k_bind: the bound entry's bits)
            world = dict(means3D=_np(g.get_xyz), opacities=_np(g.get_opacity), scales=_np(g.get_scaling), rotations=_np(g.get_rotation))
        st = oracle.forward(s, world["means3D"], shs, None, world["opacities"], world["scales"], world["rotations"], None)
        assert info["rect_instances"] == st.num_rendered
        np.testing.assert_array_equal(_np(pkg["radii"]), st.radii)
        check_image(_np(pkg["render"]), st.color, float(shs.max()) + 0.5, f"cfg2 t={ts}, fast blend")
        prev = R.set_fast_blend(False)
        try:
            with torch.no_grad():
                g.bound_render = True
                exact = render(cam, g, bench.Pipe, bg)
        finally:
            R.set_fast_blend(prev)
        assert np.array_equal(_np(exact["render"]).view(np.uint32), st.color.view(np.uint32)), f"t={ts} exact blend: max |diff| {np.abs(_np(exact['render']) - st.color).max()}"
