"""The two optional Python paths of render() (gaussian_renderer/__init__.py:63-67,73-79): `pipe.compute_cov3D_python`
(3D covariance from GaussianModel.get_covariance, scene/gaussian_model.py:162-163) and `pipe.convert_SHs_python` (colours
from utils/sh_utils.eval_sh).  Both feed the rasterizer's precomputed-input entries; the result must agree with the default
path, where the kernel evaluates the same formulas -- up to fp32 evaluation-order differences (a handful of pixels may sit on
the other side of the 1/255 or 1e-4 thresholds, bounded below)."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


class _Pipe:
    debug = False
    compute_cov3D_python = False
    convert_SHs_python = False


def _scene(dev):
    from gaussianavatars_amd import synthetic as S
    from gaussianavatars_amd.gaussian_model import GaussianModel

    sp = S.random_splats(20000, 3, 31, xyz_sigma=0.05, log_scale_mean=math.log(0.004))
    op = np.clip(sp["opacities"], 1e-6, 1 - 1e-6)
    arrs = dict(_xyz=sp["means3D"], _scaling=np.log(sp["scales"]), _rotation=sp["rotations"], _opacity=np.log(op / (1 - op)),
                _features_dc=sp["shs"][:, :1], _features_rest=sp["shs"][:, 1:])
    g = GaussianModel(3)     # un-bound: get_covariance uses the local rotation, right only here (reference quirk, kept)
    g.load_arrays(arrs, device=dev, requires_grad=True)
    cam = S.orbit_camera(208, 176, yaw_deg=15, pitch_deg=-8)
    for k in ("world_view_transform", "full_proj_transform", "camera_center"):
        setattr(cam, k, torch.as_tensor(getattr(cam, k), device=dev))
    return g, cam


def _step(g, cam, pipe, dev):
    from gaussianavatars_amd.gaussian_renderer import render

    leaves = (g._xyz, g._features_dc, g._features_rest, g._scaling, g._rotation, g._opacity)
    for p in leaves:
        p.grad = None
    pkg = render(cam, g, pipe, torch.tensor([0.1, 0.3, 0.2], device=dev))
    w = torch.randn(pkg["render"].shape, generator=torch.Generator().manual_seed(2)).to(dev)
    (pkg["render"] * w).sum().backward()
    return pkg["render"].detach(), pkg["radii"], [p.grad.clone() for p in leaves], pkg["viewspace_points"].grad.clone()


@pytest.mark.parametrize("flag", ["compute_cov3D_python", "convert_SHs_python"])
def test_python_path_agrees_with_the_kernel_path(flag):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    dev = torch.device("cuda:0")
    g, cam = _scene(dev)
    with torch.no_grad():    # the rasterizer takes the quaternion raw, get_covariance normalises it: feed unit quaternions
        g._rotation.div_(g._rotation.norm(dim=1, keepdim=True))
    base = _step(g, cam, _Pipe, dev)

    class P(_Pipe):
        pass

    setattr(P, flag, True)
    alt = _step(g, cam, P, dev)
    diff = (alt[0] - base[0]).abs()
    assert float(diff.mean()) < 2e-6 and float(diff.max()) < 2.0 / 255.0, (float(diff.mean()), float(diff.max()))
    assert int((alt[1] != base[1]).sum()) <= 2                      # radii: ceil(3 sigma) may flip on a rounding boundary
    names = ("_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity")
    for n, a, b in zip(names, alt[2], base[2]):
        scale = float(b.abs().max()) + 1e-30
        # _rotation: the python path differentiates through the normalisation, the kernel path does not (raw quaternion);
        # for unit quaternions the two differ by the radial component only -- compare the tangential part
        if n == "_rotation":
            q = g._rotation.detach()
            a = a - (a * q).sum(1, keepdim=True) * q
            b = b - (b * q).sum(1, keepdim=True) * q
        assert float((a - b).abs().max()) / scale < 2e-3, n
    assert float((alt[3] - base[3]).abs().max()) / (float(base[3].abs().max()) + 1e-30) < 2e-3


@pytest.mark.parametrize("N", [30000, 30003])   # 30003: a splat count that is no multiple of four (the backward's scratch rows stay 16-byte aligned)
def test_bound_entry_equals_accessors_plus_rasterizer(N):
    """render() of a mesh-bound model through the rasterizer's bound entry (the leaves go to world space inside the
    first kernel, include/gsr.h: gsr_forward_bound) against the reference-shaped path (get_xyz / get_scaling / get_rotation /
    get_opacity from gab_bind_forward, then the world-space rasterizer).  The transform is the same arithmetic in both libraries
    (csrc/bind_math.h), so image and radii must be the same BITS; gradients of the leaves and of the FLAME rows agree to fp32
    summation order (the face gradients are reduced per face in both, from rows written by different kernels)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import bench
    from gaussianavatars_amd import rasterizer as R
    from gaussianavatars_amd.gaussian_renderer import render

    dev = torch.device("cuda:0")
    H, W, T = 208, 176, 6
    g, cam = bench.build_scene(dev, N, 3, W, H, T, "fused", True)
    bg = torch.tensor([0.2, 0.1, 0.3], device=dev)
    wimg = torch.randn((3, H, W), generator=torch.Generator().manual_seed(4)).to(dev)
    leaves = lambda: (g._xyz, g._features_dc, g._features_rest, g._scaling, g._rotation, g._opacity)
    rows = ("expr", "rotation", "neck_pose", "jaw_pose", "eyes_pose", "translation")
    out = {}
    for fast in (True, False):
        g.bound_render = fast
        bench.zero_grads(g)
        g.select_mesh_by_timestep(3)
        pkg = render(cam, g, bench.Pipe, bg)
        assert bool(R.last_forward_info().get("bound", False)) == fast
        (pkg["render"] * wimg).sum().backward()
        out[fast] = dict(img=pkg["render"].detach().clone(), radii=pkg["radii"].clone(), vis=pkg["visibility_filter"].clone(),
                         vsp=pkg["viewspace_points"].grad.clone(), leaves=[p.grad.clone() for p in leaves()],
                         flame={k: g.flame_param[k].grad.clone() for k in rows})
    a, b = out[True], out[False]
    assert torch.equal(a["img"].view(torch.int32), b["img"].view(torch.int32))
    assert torch.equal(a["radii"], b["radii"]) and torch.equal(a["vis"], b["vis"])

    def close(x, y, tol, what):
        err = float((x.double() - y.double()).abs().max()) / (float(y.double().abs().max()) + 1e-30)
        assert err < tol, f"{what}: rel err {err:.3e}"
    close(a["vsp"], b["vsp"], 1e-5, "viewspace gradient")
    for name, x, y in zip(("_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity"), a["leaves"], b["leaves"]):
        close(x, y, 2e-5, "d" + name)
    for k in rows:
        close(a["flame"][k], b["flame"][k], 2e-4, "d flame " + k)
    # the deterministic backward through the bound entry: same bits run to run, same values as the atomic mode
    prev = R.set_deterministic(True)
    try:
        det = []
        for _ in range(2):
            g.bound_render = True
            bench.zero_grads(g)
            g.select_mesh_by_timestep(3)
            pkg = render(cam, g, bench.Pipe, bg)
            (pkg["render"] * wimg).sum().backward()
            det.append([p.grad.clone() for p in leaves()] + [g.flame_param["expr"].grad.clone()])
    finally:
        R.set_deterministic(prev)
    for x, y in zip(det[0][:6], det[1][:6]):
        assert torch.equal(x, y)          # the leaves' gradients come straight from the fixed-point sums
    for name, x, y in zip(("_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity"), det[0][:6], a["leaves"]):
        close(x, y, 2e-5, "deterministic d" + name)


def test_bound_entry_on_a_frame_large_enough_for_band_ranks():
    """300 000 bound splats at 1600 x 1100: past 262144 splats the rank path ranks per band of tile rows (csrc/gsr_rank.hip), and the
    bound entry feeds it.  Same image / radii bits as the accessor path, gradients to summation order."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import bench
    from gaussianavatars_amd import rasterizer as R
    from gaussianavatars_amd.gaussian_renderer import render

    dev = torch.device("cuda:0")
    H, W, N = 1100, 1600, 300_000
    g, cam = bench.build_scene(dev, N, 3, W, H, 4, "fused", True)
    bg = torch.ones(3, device=dev)
    out = {}
    for fast in (True, False):
        g.bound_render = fast
        bench.zero_grads(g)
        g.select_mesh_by_timestep(1)
        pkg = render(cam, g, bench.Pipe, bg)
        info = R.last_forward_info()
        assert bool(info.get("bound", False)) == fast and info["binning_path"] == 0 and info["rank_bands"] > 1, info
        pkg["render"].mean().backward()
        out[fast] = (pkg["render"].detach().clone(), pkg["radii"].clone(), g._xyz.grad.clone(), g._features_rest.grad.clone(), info["num_rendered"])
    a, b = out[True], out[False]
    assert a[4] == b[4] and torch.equal(a[0].view(torch.int32), b[0].view(torch.int32)) and torch.equal(a[1], b[1])
    for x, y, what in ((a[2], b[2], "d_xyz"), (a[3], b[3], "d_features_rest")):
        err = float((x.double() - y.double()).abs().max()) / (float(y.double().abs().max()) + 1e-30)
        assert err < 5e-5, f"{what}: rel err {err:.3e}"


def test_unbound_leaves_entry_equals_torch_activations_plus_rasterizer():
    """An UNBOUND model through the same entry without faces (rasterizer.rasterize_leaves): exp / normalize / sigmoid of
    scene/gaussian_model.py:113-160 evaluated in the first kernel instead of three torch launches.  Against the reference-shaped path
    (torch activations, world-space rasterizer): the kernel's exp is its own fmaf polynomial (~1 ulp from torch's), so values agree to
    rounding -- image within 2e-6, a handful of radii may differ by one where a 3-sigma extent sits on an integer, gradients within
    5e-5 of each tensor's max."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from gaussianavatars_amd import rasterizer as R

    dev = torch.device("cuda:0")
    g, cam = _scene(dev)
    out = {}
    for fast in (True, False):
        g.bound_render = fast
        out[fast] = _step(g, cam, _Pipe, dev)
        assert bool(R.last_forward_info().get("bound", False)) == fast
    (img_a, rad_a, gr_a, vs_a), (img_b, rad_b, gr_b, vs_b) = out[True], out[False]
    assert float((img_a - img_b).abs().max()) < 2e-6
    assert int((rad_a != rad_b).sum()) <= 3 and int((rad_a - rad_b).abs().max()) <= 1
    for name, x, y in zip(("_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity"), gr_a, gr_b):
        err = float((x - y).abs().max()) / (float(y.abs().max()) + 1e-30)
        assert err < 5e-5, f"d{name}: rel err {err:.3e}"
    assert float((vs_a - vs_b).abs().max()) / float(vs_b.abs().max()) < 5e-5


def test_bound_entry_with_every_splat_pruned():
    """A mesh-bound model whose splats have all been pruned (P == 0): render() gives the background, backward gives empty leaf gradients and
    zero face gradients instead of an error from the native entries (they have nothing to point at)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import bench
    from gaussianavatars_amd.gaussian_renderer import render

    dev = torch.device("cuda:0")
    g, cam = bench.build_scene(dev, 12000, 3, 96, 80, 3, "fused", True)
    F = int(g.binding.max().item()) + 1
    with torch.no_grad():
        for k in ("_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity"):
            setattr(g, k, torch.nn.Parameter(getattr(g, k)[:0].clone().requires_grad_(True)))
        g.binding = g.binding[:0].clone()
    if hasattr(g, "_binding_csr"):
        g._binding_csr = None
    g.select_mesh_by_timestep(1)
    assert g.face_center.shape[0] == F
    bg = torch.tensor([0.2, 0.5, 0.3], device=dev)
    pkg = render(cam, g, bench.Pipe, bg)
    assert pkg["radii"].numel() == 0 and torch.equal(pkg["render"], bg[:, None, None].expand_as(pkg["render"]))
    (pkg["render"].sum() + g.face_center.sum() * 0).backward()
    assert g._xyz.grad is None or g._xyz.grad.numel() == 0
    for k in ("expr", "rotation", "translation"):
        gr = g.flame_param[k].grad
        assert gr is None or float(gr.abs().max()) == 0.0


def test_morton_ordered_model_renders_the_same_frame():
    """io.spatial_sort is a layout choice: the same splats in another order give the same frame -- the same radii per splat, the same image
    bits except where two splats share a depth to the last bit (the depth order is stable by index there, in the reference too: with 30 k
    splats in a 0.4-wide depth range a hundred-odd pairs do, by the birthday bound), and the same gradients per splat up to those pixels and
    the order of the float atomics."""
    from gaussianavatars_amd import io as gio
    from gaussianavatars_amd import synthetic as S
    from gaussianavatars_amd.gaussian_model import GaussianModel

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    dev = torch.device("cuda:0")
    sp = S.random_splats(30000, 3, 31, xyz_sigma=0.05, log_scale_mean=math.log(0.004))
    op = np.clip(sp["opacities"], 1e-6, 1 - 1e-6)
    arrs = dict(_xyz=sp["means3D"], _scaling=np.log(sp["scales"]), _rotation=sp["rotations"], _opacity=np.log(op / (1 - op)),
                _features_dc=sp["shs"][:, :1], _features_rest=sp["shs"][:, 1:], tag=np.arange(30000))
    cam = S.orbit_camera(208, 176, yaw_deg=15, pitch_deg=-8)
    for k in ("world_view_transform", "full_proj_transform", "camera_center"):
        setattr(cam, k, torch.as_tensor(getattr(cam, k), device=dev))
    res = []
    for a in (arrs, gio.spatial_sort(arrs)):
        g = GaussianModel(3)
        g.load_arrays(a, device=dev, requires_grad=True)
        res.append((_step(g, cam, _Pipe, dev), torch.as_tensor(a["tag"], device=dev)))
    (img0, rad0, gr0, vs0), _ = res[0]
    (img1, rad1, gr1, vs1), perm = res[1]
    assert not torch.equal(perm, torch.arange(30000, device=dev))
    assert torch.equal(rad0[perm], rad1)
    differ = (img0 != img1).any(0)
    # measured: 49 pixels of 36 608, max |diff| 1.8e-3; gradients 2e-4 .. 1.4e-3 of each tensor's maximum
    assert float(differ.float().mean()) < 0.005 and float((img0 - img1).abs().max()) < 0.01
    for a, b in zip(gr0 + [vs0], gr1 + [vs1]):
        err = float((a[perm] - b).abs().max()) / (float(a.abs().max()) + 1e-30)
        assert err <= 5e-3, err


@pytest.mark.fast_blend
@pytest.mark.parametrize("fast", [True, False])
def test_no_grad_render_of_trainable_parameters_takes_the_forward_only_path(fast):
    """render.py:68-76 / fps_benchmark_*.py render a model whose leaves are nn.Parameters under torch.no_grad(), and render() hands the rasterizer a
    screen-space leaf that requires a gradient: whether a backward can follow is the caller's grad mode, not ctx.needs_input_grad (which only
    repeats requires_grad).  Under no_grad every entry (bound, leaves, world-space module) must take GsrSettings.forward_only -- no accumulator
    zero-fill, no covariance / checkpoint / final-state stores -- and give the SAME BITS as the frame rendered with autograd on, whose backward
    still works afterwards; poisoned state buffers show that nothing the image depends on was skipped."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import bench
    from gaussianavatars_amd import rasterizer as R
    from gaussianavatars_amd.gaussian_renderer import render

    dev = torch.device("cuda:0")
    prev_fast, prev_poison = R.set_fast_blend(fast), R._poison_state
    R._poison_state = 1
    try:
        H, W = 208, 176
        gb, cam = bench.build_scene(dev, 20000, 3, W, H, 4, "fused", True)
        gb.select_mesh_by_timestep(2)
        gu, cam_u = _scene(dev)
        bg = torch.tensor([0.2, 0.1, 0.3], device=dev)
        for g, c, what in ((gb, cam, "bound"), (gu, cam_u, "leaves")):
            pkg = render(c, g, bench.Pipe, bg)
            info = R.last_forward_info()
            assert info["forward_only"] is False, what
            with torch.no_grad():
                ng = render(c, g, bench.Pipe, bg)
            info = R.last_forward_info()
            assert info["forward_only"] is True, what
            assert torch.equal(ng["render"].view(torch.int32), pkg["render"].detach().view(torch.int32)), what
            assert torch.equal(ng["radii"], pkg["radii"]) and torch.equal(ng["visibility_filter"], pkg["visibility_filter"]), what
            assert ng["render"].grad_fn is None
            pkg["render"].sum().backward()      # the grad-mode frame's state is its own: its backward is untouched by the frame in between
            assert float(g._xyz.grad.abs().max()) > 0 and bool(torch.isfinite(g._xyz.grad).all())
        # the reference-shaped module (world-space tensors, what an unpatched gaussian_renderer.render() calls)
        rs = R.GaussianRasterizationSettings(H, W, math.tan(cam_u.FoVx * 0.5), math.tan(cam_u.FoVy * 0.5), bg, 1.0, cam_u.world_view_transform,
                                             cam_u.full_proj_transform, 3, cam_u.camera_center, False, False)
        rast = R.GaussianRasterizer(rs)
        args = dict(means3D=gu.get_xyz, means2D=torch.zeros_like(gu.get_xyz, requires_grad=True), opacities=gu.get_opacity, shs=gu.get_features,
                    scales=gu.get_scaling, rotations=gu.get_rotation)
        img, radii = rast(**args)
        assert R.last_forward_info()["forward_only"] is False
        with torch.no_grad():
            img2, radii2 = rast(**args)
        assert R.last_forward_info()["forward_only"] is True
        assert torch.equal(img2.view(torch.int32), img.detach().view(torch.int32)) and torch.equal(radii, radii2)
        with torch.inference_mode():
            img3, _ = rast(**{k: v.detach() for k, v in args.items()})
        assert R.last_forward_info()["forward_only"] is True and torch.equal(img3.view(torch.int32), img.detach().view(torch.int32))
    finally:
        R.set_fast_blend(prev_fast)
        R._poison_state = prev_poison


@pytest.mark.fast_blend
def test_late_count_wait_equals_the_blocking_form_and_replays_a_frame_that_did_not_fit():
    """The leaves entries (render() of a bound or un-bound model) take the deferred form of gsr_forward and wait for the instance count behind
    the call's own host work (rasterizer._apply_leaves_entry); GSR_LATE_COUNT=0 / rasterizer._late_count = False is the blocking form.  Same image
    bits, radii, instance count and gradients either way; a frame that does not fit its binning buffer (capacity hint forced down to a
    fraction of the frame) is rendered again with a larger one -- the image the caller gets is the full frame, `replays` says so, and the
    backward of the returned node works."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import bench
    from gaussianavatars_amd import rasterizer as R
    from gaussianavatars_amd.gaussian_renderer import render

    dev = torch.device("cuda:0")
    H, W = 208, 176
    g, cam = bench.build_scene(dev, 20000, 3, W, H, 4, "fused", True)
    gu, cam_u = _scene(dev)
    bg = torch.tensor([0.2, 0.1, 0.3], device=dev)
    prev = R._late_count
    try:
        for model, c, what in ((g, cam, "bound"), (gu, cam_u, "leaves")):
            out = {}
            for late in (False, True):
                R._late_count = late
                if what == "bound":
                    bench.zero_grads(model)
                    model.select_mesh_by_timestep(1)
                else:
                    for p_ in (model._xyz, model._features_dc, model._features_rest, model._scaling, model._rotation, model._opacity):
                        p_.grad = None
                pkg = render(c, model, bench.Pipe, bg)
                info = R.last_forward_info()
                pkg["render"].sum().backward()
                out[late] = (pkg["render"].detach().clone(), pkg["radii"].clone(), int(info["num_rendered"]), model._xyz.grad.clone())
            assert out[True][2] == out[False][2] > 0, what
            assert torch.equal(out[True][0].view(torch.int32), out[False][0].view(torch.int32)) and torch.equal(out[True][1], out[False][1]), what
            err = float((out[True][3] - out[False][3]).abs().max()) / float(out[False][3].abs().max())
            assert err < 1e-5, (what, err)      # float atomics: two backwards of the same frame differ by rounding
            # a buffer far too small for the frame: rendered again, transparently
            R._late_count = True
            I = out[True][2]
            for key in list(R._capacity_hint):
                R._capacity_hint[key] = max(1024, I // 7)
            if what == "bound":
                model.select_mesh_by_timestep(1)
            pkg = render(c, model, bench.Pipe, bg)
            info = R.last_forward_info()
            assert info["replays"] >= 1 and int(info["num_rendered"]) == I and info["capacity"] >= I, (what, info)
            assert torch.equal(pkg["render"].detach().view(torch.int32), out[True][0].view(torch.int32)), what
            pkg["render"].sum().backward()
            assert bool(torch.isfinite(model._xyz.grad).all())
            pkg2 = render(c, model, bench.Pipe, bg)                      # the next frame fits at once
            assert R.last_forward_info()["replays"] == 0 and torch.equal(pkg2["render"].detach().view(torch.int32), out[True][0].view(torch.int32))
        # the reference-shaped module (what an unpatched gaussian_renderer.render() calls) takes the same route
        rs = R.GaussianRasterizationSettings(H, W, math.tan(cam_u.FoVx * 0.5), math.tan(cam_u.FoVy * 0.5), bg, 1.0, cam_u.world_view_transform,
                                             cam_u.full_proj_transform, 3, cam_u.camera_center, False, False)
        rast = R.GaussianRasterizer(rs)
        args = dict(means3D=gu.get_xyz, means2D=torch.zeros_like(gu.get_xyz, requires_grad=True), opacities=gu.get_opacity, shs=gu.get_features,
                    scales=gu.get_scaling, rotations=gu.get_rotation)
        R._late_count = False
        img0, radii0 = rast(**args)
        I = int(R.last_forward_info()["num_rendered"])
        R._late_count = True
        img1, radii1 = rast(**args)
        assert int(R.last_forward_info()["num_rendered"]) == I and torch.equal(img1.view(torch.int32), img0.view(torch.int32)) and torch.equal(radii0, radii1)
        for key in list(R._capacity_hint):
            R._capacity_hint[key] = max(1024, I // 5)
        img2, _ = rast(**args)
        info = R.last_forward_info()
        assert info["replays"] >= 1 and int(info["num_rendered"]) == I and torch.equal(img2.view(torch.int32), img0.view(torch.int32))
        img2.sum().backward()
        assert bool(torch.isfinite(gu._xyz.grad).all()) and float(gu._xyz.grad.abs().max()) > 0
    finally:
        R._late_count = prev
