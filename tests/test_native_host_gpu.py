"""The COMPILED host side (gaussianavatars_amd/gaa_host.so, csrc/gaa_host.cpp) against its Python twins (rasterizer._RasterizeBound,
binding._MeshFramesTimestep, loss._L1 / _L1Ssim): the same launches through the same C ABI, so forward results must be the same BITS and
gradients agree to the order of the float atomics.  Everything else of the GPU suite runs with the compiled host (the default); these tests are
the A/B and the host-logic corners: a frame that does not fit its binning buffer, no_grad, a second backward, an in-place update between forward
and backward, the unit-seed shortcut of the L1 node."""
import math

import numpy as np
import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.fast_blend]


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _params(g):
    return [g._xyz, g._features_dc, g._features_rest, g._scaling, g._rotation, g._opacity, g.flame_param["expr"], g.flame_param["rotation"],
            g.flame_param["neck_pose"], g.flame_param["jaw_pose"], g.flame_param["eyes_pose"], g.flame_param["translation"]]


def _step(g, cam, bg, target, t, ssim=False):
    import bench
    from gaussianavatars_amd import loss as L
    from gaussianavatars_amd.gaussian_renderer import render

    bench.zero_grads(g)
    g.select_mesh_by_timestep(t)
    mesh = [x.detach().clone() for x in (g.verts, g.verts_cano, g.face_center, g.face_orien_mat, g.face_scaling, g.face_orien_quat)]
    pkg = render(cam, g, bench.Pipe, bg)
    if ssim:
        l1, ss = L.l1_ssim(pkg["render"], target)
        loss = 0.8 * l1 + 0.2 * (1.0 - ss)
    else:
        loss = L.l1_loss(pkg["render"], target)
    loss.backward()
    grads = [p.grad.detach().clone() for p in _params(g)] + [pkg["viewspace_points"].grad.detach().clone()]
    return dict(mesh=mesh, image=pkg["render"].detach().clone(), radii=pkg["radii"].clone(), visible=pkg["visibility_filter"].clone(), loss=float(loss), grads=grads)


@pytest.mark.parametrize("ssim", [False, True])
def test_compiled_host_equals_python_twins(ssim):
    import bench
    from gaussianavatars_amd import _host
    from gaussianavatars_amd import loss as L
    from gaussianavatars_amd import rasterizer as R

    dev = _dev()
    H, W = 401, 275
    g, cam = bench.build_scene(dev, 40_000, 3, W, H, 6, "fused", True)
    bg = torch.ones(3, device=dev)
    target = torch.full((3, H, W), 0.35, device=dev)
    L.install_backward_seed()
    try:
        out = {}
        for native in (True, False):
            prev = _host.set_enabled(native)
            try:
                res = [_step(g, cam, bg, target, t, ssim) for t in (0, 3, 3)]
                assert R.last_forward_info()["native_host"] is native
                out[native] = res
            finally:
                _host.set_enabled(prev)
    finally:
        L.install_backward_seed(False)
    eq = lambda a, b: torch.equal(a.view(torch.int32), b.view(torch.int32))
    for a, b in zip(out[True], out[False]):
        for x, y in zip(a["mesh"], b["mesh"]):
            assert eq(x, y)
        assert eq(a["image"], b["image"]) and torch.equal(a["radii"], b["radii"]) and torch.equal(a["visible"], b["visible"])
        assert a["loss"] == b["loss"]
        for i, (x, y) in enumerate(zip(a["grads"], b["grads"])):
            err = float((x - y).abs().max()) / (float(y.abs().max()) + 1e-30)
            assert err < (2e-3 if 6 <= i < 12 else 3e-4), (i, err)      # FLAME rows: sums of cancelling terms (tests/test_fullsize_gpu.py states the same bar)
            assert float(y.abs().max()) > 0


def test_compiled_host_replays_a_frame_that_did_not_fit_and_renders_under_no_grad():
    import bench
    from gaussianavatars_amd import _host
    from gaussianavatars_amd import rasterizer as R
    from gaussianavatars_amd.gaussian_renderer import render

    dev = _dev()
    assert _host.enabled()
    H, W = 401, 275
    g, cam = bench.build_scene(dev, 40_000, 3, W, H, 4, "fused", True)
    bg = torch.ones(3, device=dev)
    g.select_mesh_by_timestep(1)
    ref = render(cam, g, bench.Pipe, bg)
    info = R.last_forward_info()
    assert info["native_host"] is True and info["replays"] == 0 and info["forward_only"] is False
    I = info["num_rendered"]
    # force the capacity hint far below the frame: the first attempt renders nothing, the entry raises the hint and calls again
    for k in list(R._capacity_hint):
        R._capacity_hint[k] = R._CAP_QUANTUM
    assert I > R._CAP_QUANTUM
    again = render(cam, g, bench.Pipe, bg)
    info = R.last_forward_info()
    assert info["replays"] >= 1 and info["num_rendered"] == I
    assert torch.equal(again["render"].view(torch.int32), ref["render"].view(torch.int32)) and torch.equal(again["radii"], ref["radii"])
    again["render"].sum().backward()
    assert float(g._xyz.grad.abs().max()) > 0 and bool(torch.isfinite(g._xyz.grad).all())
    nxt = render(cam, g, bench.Pipe, bg)
    assert R.last_forward_info()["replays"] == 0 and torch.equal(nxt["render"].view(torch.int32), ref["render"].view(torch.int32))
    # no_grad (render.py / fps_benchmark_*.py): forward_only, no node, the same bits
    with torch.no_grad():
        g.select_mesh_by_timestep(1)
        ng = render(cam, g, bench.Pipe, bg)
    assert R.last_forward_info()["forward_only"] is True and ng["render"].grad_fn is None and g.face_center.grad_fn is None
    assert torch.equal(ng["render"].view(torch.int32), ref["render"].view(torch.int32))
    assert torch.equal(ng["visibility_filter"], ref["radii"] > 0)


def test_compiled_nodes_second_backward_and_inplace_update():
    import bench
    from gaussianavatars_amd import _host
    from gaussianavatars_amd import loss as L
    from gaussianavatars_amd.gaussian_renderer import render

    dev = _dev()
    assert _host.enabled()
    g, cam = bench.build_scene(dev, 20_000, 3, 256, 192, 3, "fused", True)
    bg = torch.ones(3, device=dev)
    target = torch.full((3, cam.image_height, cam.image_width), 0.3, device=dev)
    params = _params(g)
    g.select_mesh_by_timestep(2)
    pkg = render(cam, g, bench.Pipe, bg)
    loss = L.l1_loss(pkg["render"], target)
    assert type(loss.grad_fn).__name__ != "_L1Backward"            # a C++ node, not the Python Function's
    first = torch.autograd.grad(loss, params, retain_graph=True)
    second = torch.autograd.grad(loss, params)
    for a, b in zip(first, second):
        scale = float(a.abs().max()) + 1e-30
        assert scale > 1e-12 and float((a - b).abs().max()) / scale < 2e-4
    with pytest.raises(RuntimeError, match="second time|released"):
        torch.autograd.grad(loss, params)                            # the rasterizer state went with the un-retained backward
    # one mesh update serves two render + backward passes (the mesh node keeps what it needs, like binding._Keep)
    g.select_mesh_by_timestep(1)
    for _ in range(2):
        bench.zero_grads(g)
        L.l1_loss(render(cam, g, bench.Pipe, bg)["render"], target).backward()
        assert float(g.flame_param["expr"].grad[1].abs().max()) > 0
    # an in-place update of a leaf between forward and backward raises, as in stock autograd
    g.select_mesh_by_timestep(0)
    pkg = render(cam, g, bench.Pipe, bg)
    with torch.no_grad():
        g._xyz.mul_(1.0)
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        pkg["render"].sum().backward()


def test_compiled_l1_node_unit_seed_and_other_gradients():
    from gaussianavatars_amd import _host
    from gaussianavatars_amd import loss as L

    dev = _dev()
    assert _host.enabled()
    H = _host.get()
    a0, b = torch.rand(3, 61, 47, device=dev), torch.rand(3, 61, 47, device=dev)
    ref = torch.sign(a0 - b) / a0.numel()
    L.install_backward_seed()
    try:
        a = a0.clone().requires_grad_(True)
        loss = L.l1_loss(a, b)
        assert abs(float(loss) - float((a0 - b).abs().mean())) < 2e-6
        loss.backward()                                              # seeded with the cached 1: the image the forward left behind
        assert torch.equal(a.grad, ref) and H.l1_emit_state() == 1
        a = a0.clone().requires_grad_(True)
        (0.8 * L.l1_loss(a, b)).backward()
        assert torch.allclose(a.grad, 0.8 * ref, rtol=1e-6, atol=0)
        a = a0.clone().requires_grad_(True)
        L.l1_loss(a, b).backward(torch.full((), 3.0, device=dev))
        assert torch.allclose(a.grad, 3.0 * ref, rtol=1e-6, atol=0)
        assert H.l1_emit_state() == 0                                # two backwards the precomputed image could not serve: switched off ...
        a = a0.clone().requires_grad_(True)
        L.l1_loss(a, b).backward()
        assert torch.equal(a.grad, ref)                              # ... and the seeded backward takes the kernel: same gradient
        a, b2 = a0.clone().requires_grad_(True), b.clone().requires_grad_(True)
        L.l1_loss(a, b2).backward()
        assert torch.equal(a.grad, ref) and torch.equal(b2.grad, -ref)
        # l1_ssim: the two scalars against the stand-alone functions
        prev = _host.set_enabled(False)
        try:
            a = a0.clone().requires_grad_(True)
            l1p, ssp = L.l1_ssim(a, b)
            (0.8 * l1p + 0.2 * (1 - ssp)).backward()
            gp = a.grad.clone()
        finally:
            _host.set_enabled(prev)
        a = a0.clone().requires_grad_(True)
        l1n, ssn = L.l1_ssim(a, b)
        (0.8 * l1n + 0.2 * (1 - ssn)).backward()
        assert float(l1n) == float(l1p) and float(ssn) == float(ssp) and torch.equal(a.grad, gp)
        with torch.no_grad():
            l1e, sse = L.l1_ssim(a0, b)
        assert float(l1e) == float(l1p) and float(sse) == float(ssp)
    finally:
        L.install_backward_seed(False)
