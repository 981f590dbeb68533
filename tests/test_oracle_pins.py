"""CPU tests: the oracle (and the torch restatements it is cross-checked with) against golden vectors
generated FROM THE REFERENCE's own in-tree code (tests/golden/make_golden.py), and the oracle's
analytic backward against autograd of an independent fp64 restatement + finite differences."""
import math
import os

import numpy as np
import pytest
import torch

from tests.scenes import scene, settings_args

G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def pins():
    return np.load(os.path.join(G, "raster_pins.npz"))


def _front_camera_settings(oracle, H, W, deg, campos, mod=1.0):
    # identity view looking down +z, generic projection: only colour / cov3D outputs are inspected
    view = np.eye(4, dtype=np.float32)
    proj = np.eye(4, dtype=np.float32)
    proj[2, 3] = 1.0  # w = z  (stored transposed: element [2,3] of M^T)
    proj[3, 3] = 0.0
    return oracle.make_settings(H, W, 0.5, 0.5, [0, 0, 0], mod, view, proj, deg, campos)


def test_sh_colour_matches_reference_eval_sh(oracle, pins):
    means = pins["sh_means"].copy()
    means[:, 2] += 2.0  # in front of the identity camera
    campos = pins["sh_campos"].copy()
    campos[2] += 2.0    # keeps (mean - campos) identical to the fixture's
    N = means.shape[0]
    for deg in range(4):
        s = _front_camera_settings(oracle, 64, 64, deg, campos)
        st = oracle.forward(s, means, pins["sh_shs"], None, np.full((N, 1), 0.5, np.float32),
                            np.full((N, 3), 0.05, np.float32), np.tile(np.array([1, 0, 0, 0], np.float32), (N, 1)), None)
        vis = st.radii > 0
        assert vis.sum() > N // 2
        exp = pins[f"sh_rgb{deg}"]
        np.testing.assert_allclose(st.rgb[vis], exp[vis], rtol=0, atol=2e-6)
        np.testing.assert_array_equal(st.clamped[vis], (exp[vis] <= 0) & (st.rgb[vis] == 0) & st.clamped[vis].astype(bool))


def test_cov3d_matches_reference_build_scaling_rotation(oracle, pins):
    N = pins["cov_scales"].shape[0]
    means = np.zeros((N, 3), np.float32)
    means[:, 2] = 2.0
    s = _front_camera_settings(oracle, 64, 64, 0, np.zeros(3, np.float32), mod=float(pins["cov_mod"]))
    st = oracle.forward(s, means, np.zeros((N, 1, 3), np.float32), None, np.full((N, 1), 0.5, np.float32),
                        pins["cov_scales"], pins["cov_quat"], None)
    assert (st.radii > 0).all()
    np.testing.assert_allclose(st.cov3D, pins["cov_expected"], rtol=2e-5, atol=1e-9)


def test_camera_convention_matches_reference_cameras(oracle, pins):
    W, H = int(pins["cam_W"]), int(pins["cam_H"])
    pts = pins["cam_points"]
    N = pts.shape[0]
    s = oracle.make_settings(H, W, math.tan(float(pins["cam_fovx"]) / 2), math.tan(float(pins["cam_fovy"]) / 2), [0, 0, 0], 1.0,
                             pins["cam_viewmatrix"], pins["cam_projmatrix"], 0, pins["cam_center"])
    st = oracle.forward(s, pts, np.zeros((N, 1, 3), np.float32), None, np.full((N, 1), 0.5, np.float32),
                        np.full((N, 3), 0.01, np.float32), np.tile(np.array([1, 0, 0, 0], np.float32), (N, 1)), None)
    vis = st.radii > 0
    assert vis.sum() > N // 2
    np.testing.assert_allclose(st.depths[vis], pins["cam_depth"][vis], rtol=1e-6)
    np.testing.assert_allclose(st.xy[vis], pins["cam_pix"][vis], rtol=0, atol=2e-3)
    assert ((pins["cam_depth"] > 0.2) >= vis).all()


def test_orbit_camera_matches_survey_numbers():
    """fps_benchmark_demo.py camera: the matrix values SURVEY.md 8(d) cfg 2 derives from
    utils/viewer_utils.py:59-66,142-170."""
    from gaussianavatars_amd import synthetic as S

    cam = S.orbit_camera(550, 802)
    assert abs(math.tan(cam.FoVy / 2) - 0.1763270) < 1e-6
    assert abs(math.tan(cam.FoVx / 2) - 0.1209225) < 1e-6
    PV = cam.full_proj_transform.T
    exp = np.array([[8.26976, 0, 0, 0], [0, -5.671282, 0, 0], [0, 0, -1.002002, 0.981982], [0, 0, -1, 1]], np.float32)
    np.testing.assert_allclose(PV, exp, atol=2e-5)
    np.testing.assert_allclose(cam.camera_center, [0, 0, 1], atol=1e-7)


def test_expf_accuracy(oracle):
    x = -np.random.default_rng(0).uniform(0, 30, 20000).astype(np.float32)
    got = oracle.expf(x).astype(np.float64)
    ref = np.exp(x.astype(np.float64))
    assert (np.abs(got - ref) / ref).max() < 2.0 * 2.0 ** -24
    assert oracle.expf(np.array([0.0], np.float32))[0] == 1.0
    assert oracle.expf(np.array([-200.0], np.float32))[0] == 0.0


def test_binning_invariants(oracle):
    """Sortedness / range consistency (size-independent properties used again at full size on the GPU)."""
    for name in ("cfg1", "culls", "dense_tile", "empty_view"):
        cam, sp, bg, deg, mod = scene(name)
        s = oracle.make_settings(**settings_args(cam, bg, deg, mod))
        st = oracle.forward(s, sp["means3D"], sp["shs"], None, sp["opacities"], sp["scales"], sp["rotations"], None)
        assert st.num_rendered == int(st.tiles_touched.sum())
        if st.num_rendered:
            assert (np.diff(st.keys.astype(np.uint64)) >= 0).all()
            # stable: equal keys keep ascending splat index
            same = np.diff(st.keys.astype(np.uint64)) == 0
            assert (np.diff(st.point_list.astype(np.int64))[same] > 0).all()
        cnt = (st.ranges[:, 1] - st.ranges[:, 0]).astype(np.int64)
        assert cnt.sum() == st.num_rendered
        assert (st.n_contrib <= cnt.reshape(-1)[_tile_of_pixels(cam)]).all()


def _tile_of_pixels(cam):
    H, W = cam.image_height, cam.image_width
    gx = (W + 15) // 16
    ys, xs = np.mgrid[0:H, 0:W]
    return (ys // 16) * gx + xs // 16


@pytest.mark.parametrize("deg,mod", [(3, 1.0), (1, 1.3)])
def test_oracle_backward_matches_fp64_autograd(oracle, deg, mod):
    from gaussianavatars_amd import synthetic as S
    from oracle import torch_ref as TR

    H, W, N = 48, 64, 80
    cam = S.orbit_camera(W, H)
    sp = S.random_splats(N, 3, 7, xyz_sigma=0.05, log_scale_mean=math.log(0.01))
    sp["means3D"][:3, 0] += 0.4  # outside the 1.3x frustum: clamp branch
    tfx, tfy = math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2)
    bg = [0.3, 0.9, 0.5]
    # exact_scale_grad: the chain rule through scale_modifier, which is what autograd of the restatement computes
    s = oracle.make_settings(H, W, tfx, tfy, bg, mod, cam.world_view_transform, cam.full_proj_transform, deg, cam.camera_center,
                             exact_scale_grad=True)
    st = oracle.forward(s, sp["means3D"], sp["shs"], None, sp["opacities"], sp["scales"], sp["rotations"], None)
    gpix = np.random.default_rng(3).normal(0, 1, (3, H, W)).astype(np.float32)
    g = oracle.backward(s, st, gpix)
    # default mode = upstream's computeCov3D backward: dL/dscale is the gradient w.r.t. (scale_modifier * scale), i.e. the
    # exact one divided by the modifier; every other gradient is the same
    s_up = oracle.make_settings(H, W, tfx, tfy, bg, mod, cam.world_view_transform, cam.full_proj_transform, deg, cam.camera_center)
    g_up = oracle.backward(s_up, st, gpix)
    np.testing.assert_allclose(g_up["scales"] * np.float32(mod), g["scales"], rtol=2e-6, atol=1e-30)
    for k in ("means3D", "means2D", "shs", "opacities", "rotations"):
        np.testing.assert_array_equal(g_up[k], g[k])
    t = lambda a: torch.tensor(a, dtype=torch.float64, requires_grad=True)
    m3, m2, sh, op, sc, ro = (t(sp["means3D"]), t(np.zeros((N, 3))), t(sp["shs"]), t(sp["opacities"]), t(sp["scales"]),
                              t(sp["rotations"]))
    img = TR.render(H, W, tfx, tfy, bg, mod, torch.tensor(cam.world_view_transform), torch.tensor(cam.full_proj_transform), deg,
                    torch.tensor(cam.camera_center), m3, m2, sh, None, op, sc, ro, None, st.radii, st.ranges, st.point_list,
                    st.n_contrib)
    assert float((img.detach().numpy() - st.color).__abs__().max()) < 5e-6
    (img * torch.tensor(gpix, dtype=torch.float64)).sum().backward()
    for name, tt in [("means3D", m3), ("means2D", m2), ("shs", sh), ("opacities", op), ("scales", sc), ("rotations", ro)]:
        a, b = g[name], tt.grad.numpy().reshape(g[name].shape)
        assert np.abs(a - b).max() / (np.abs(b).max() + 1e-30) < 1e-4, name


def test_fp64_restatement_gradient_is_a_gradient():
    """Finite differences on the fp64 restatement itself (so the chain oracle == autograd == FD closes)."""
    from gaussianavatars_amd import synthetic as S
    from oracle import gsr_oracle as O, torch_ref as TR

    H, W, N = 32, 32, 12
    cam = S.orbit_camera(W, H)
    sp = S.random_splats(N, 1, 9, xyz_sigma=0.03, log_scale_mean=math.log(0.02))
    tfx, tfy = math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2)
    s = O.make_settings(H, W, tfx, tfy, [1, 1, 1], 1.0, cam.world_view_transform, cam.full_proj_transform, 1, cam.camera_center)
    st = O.forward(s, sp["means3D"], sp["shs"], None, sp["opacities"], sp["scales"], sp["rotations"], None)
    wts = torch.tensor(np.random.default_rng(1).normal(0, 1, (3, H, W)))

    def f(m3, sc, ro, op):
        img = TR.render(H, W, tfx, tfy, [1, 1, 1], 1.0, torch.tensor(cam.world_view_transform), torch.tensor(cam.full_proj_transform),
                        1, torch.tensor(cam.camera_center), m3, torch.zeros(N, 3, dtype=torch.float64),
                        torch.tensor(sp["shs"], dtype=torch.float64), None, op, sc, ro, None, st.radii, st.ranges, st.point_list,
                        st.n_contrib)
        return (img * wts).sum()

    t = lambda a: torch.tensor(a, dtype=torch.float64, requires_grad=True)
    assert torch.autograd.gradcheck(f, (t(sp["means3D"]), t(sp["scales"]), t(sp["rotations"]), t(sp["opacities"])),
                                    eps=1e-7, atol=1e-5, rtol=1e-3, nondet_tol=0.0)


def test_oracle_results_do_not_depend_on_the_thread_count(oracle):
    """The oracle's sort (stable grouping by tile, per-tile radix sort) and backward (per-tile partial rows, folded per splat in list
    order) are shared by the host cores -- bench.py's cpu_baseline times them on every core and on one -- with the same bits either way."""
    import math

    from tests.scenes import scene, settings_args

    cam, sp, bg, deg, mod = scene("sh3_small")
    s = oracle.make_settings(**settings_args(cam, bg, deg, mod))
    gpix = np.random.default_rng(3).normal(0, 1, (3, cam.image_height, cam.image_width)).astype(np.float32)
    out = {}
    try:
        for th in (1, 4):
            oracle.set_threads(th)
            st = oracle.forward(s, sp["means3D"], sp["shs"], None, sp["opacities"], sp["scales"], sp["rotations"], None)
            g = oracle.backward(s, st, gpix)
            out[th] = [st.keys, st.point_list, st.ranges, st.color, st.n_contrib] + [g[k] for k in ("means3D", "means2D", "shs", "opacities", "scales", "rotations")]
    finally:
        oracle.set_threads(0)
    assert out[1][0].size > 4096                       # (the parallel grouping takes its multi-chunk path)
    for a, b in zip(out[1], out[4]):
        assert np.array_equal(a, b)
