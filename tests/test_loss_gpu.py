"""GPU parity tests of the fused loss / statistics kernels (include/gls.h) through the Python mirror of
utils/loss_utils.py.  Bars (fp32 kernels, separable window vs the reference's 2-D window):
value of l1 within 2e-6 abs, ssim within 2e-5 abs; l1 gradient exact up to the 1/n scale (rel 1e-6);
ssim gradient within 3e-4 of the per-tensor max magnitude; statistics update: integer-valued outputs
exact, accumulated norm within 2 ulp."""
import os

import numpy as np
import pytest
import torch

from oracle import loss_oracle as LO

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
PINS = np.load(os.path.join(HERE, "golden", "loss_pins.npz"))


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _t(x, dev, grad=False):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev).requires_grad_(grad)


@pytest.mark.parametrize("case", ["chw", "bchw", "tiny"])
def test_matches_reference_pins(case):
    from gaussianavatars_amd import loss

    dev = _dev()
    a, b = PINS[f"{case}_a"], PINS[f"{case}_b"]
    ta, tb = _t(a, dev, True), _t(b, dev)
    l1 = loss.l1_loss(ta, tb)
    (g_l1,) = torch.autograd.grad(l1, ta)
    ss = loss.ssim(ta, tb)
    (g_ss,) = torch.autograd.grad(ss, ta)
    l1, ss = l1.detach(), ss.detach()
    assert abs(float(l1) - float(PINS[f"{case}_l1"])) < 2e-6
    assert abs(float(ss) - float(PINS[f"{case}_ssim"])) < 2e-5
    np.testing.assert_allclose(g_l1.cpu().numpy(), PINS[f"{case}_g_l1"], rtol=1e-6, atol=1e-10)
    ref = PINS[f"{case}_g_ssim"]
    assert np.abs(g_ss.cpu().numpy() - ref).max() < 3e-4 * np.abs(ref).max()
    # the fused pair equals the two separate calls, values and combined gradient
    ta2 = _t(a, dev, True)
    f1, fs = loss.l1_ssim(ta2, tb)
    # (SSIM: the same kernel, the same bits.  L1: the pair's comes out of the SSIM pass's per-tile partials, the stand-alone one out of the one-launch
    #  fixed-point reduction of round 6 -- two summation orders of the same mean: equal to two units in the last place)
    assert abs(float(f1.detach()) - float(l1)) <= 2.4e-7 * abs(float(l1)) and float(fs.detach()) == float(ss)
    (0.8 * f1 + 0.2 * (1.0 - fs)).backward()
    comb = 0.8 * g_l1 - 0.2 * g_ss
    assert float((ta2.grad - comb).abs().max()) < 1e-6 * float(comb.abs().max()) + 1e-12
    if a.ndim == 4:
        per = loss.ssim(ta, tb, size_average=False)
        np.testing.assert_allclose(per.detach().cpu().numpy(), PINS[f"{case}_ssim_per_image"], rtol=0, atol=2e-5)


def test_full_size_against_oracle():
    """BASELINE image size (3 x 802 x 550): ragged right/bottom tiles, value + gradient vs the fp64 oracle."""
    from gaussianavatars_amd import loss

    dev = _dev()
    g = np.random.default_rng(9)
    yy, xx = np.mgrid[0:802, 0:550]
    base = 0.5 + 0.4 * np.sin(xx / 37.0)[None] * np.cos(yy / 23.0)[None] * np.array([1.0, 0.7, 0.4])[:, None, None]
    a = np.clip(base + g.normal(0, 0.05, base.shape), 0, 1).astype(np.float32)
    b = np.clip(base + g.normal(0, 0.02, base.shape), 0, 1).astype(np.float32)
    ta, tb = _t(a, dev, True), _t(b, dev)
    l1, ss = loss.l1_ssim(ta, tb)
    (0.8 * l1 + 0.2 * (1.0 - ss)).backward()
    assert abs(float(l1) - LO.l1(a, b)) < 2e-6
    assert abs(float(ss) - LO.ssim(a, b)) < 2e-5
    ref = 0.8 * LO.l1_grad(a, b) - 0.2 * LO.ssim_grad(a, b)
    assert np.abs(ta.grad.cpu().numpy() - ref).max() < 3e-4 * np.abs(ref).max()
    # deterministic: a second evaluation is bit-identical
    l1b, ssb = loss.l1_ssim(_t(a, dev), tb)
    assert float(l1b) == float(l1) and float(ssb) == float(ss)


def test_l1_one_launch_reduction_is_deterministic():
    """Round 6: k_l1_fwd reduces over its workgroups with one returning fixed-point atomic each (no second launch): integer adds commute, so
    the loss is the same bits every time, on every stream, and a call leaves the stream's accumulator word zero for the next one."""
    from gaussianavatars_amd import loss

    dev = _dev()
    g = np.random.default_rng(11)
    a, b = g.random((3, 802, 550), dtype=np.float32), g.random((3, 802, 550), dtype=np.float32)
    ta, tb = _t(a, dev), _t(b, dev)
    first = float(loss.l1_loss(ta, tb))
    assert abs(first - LO.l1(a, b)) < 2e-6
    for _ in range(5):
        assert float(loss.l1_loss(ta, tb)) == first
    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        other = [loss.l1_loss(ta, tb) for _ in range(3)]
    side.synchronize()
    assert all(float(o) == first for o in other)
    # a small ragged size (fewer workgroups than 255, n % 4 != 0)
    sa, sb = _t(a[:, :7, :5].copy(), dev), _t(b[:, :7, :5].copy(), dev)
    assert abs(float(loss.l1_loss(sa, sb)) - LO.l1(a[:, :7, :5], b[:, :7, :5])) < 2e-6


def test_gradient_wrt_second_image_and_identity():
    from gaussianavatars_amd import loss

    dev = _dev()
    a, b = PINS["chw_a"], PINS["chw_b"]
    ta, tb = _t(a, dev), _t(b, dev, True)
    loss.ssim(ta, tb).backward()
    ref = LO.ssim_grad(b, a)
    assert np.abs(tb.grad.cpu().numpy() - ref).max() < 3e-4 * np.abs(ref).max()
    assert abs(float(loss.ssim(ta, ta)) - 1.0) < 1e-6 and float(loss.l1_loss(ta, ta)) == 0.0


def test_l1_odd_sizes():
    from gaussianavatars_amd import loss

    dev = _dev()
    g = np.random.default_rng(3)
    for n in (1, 3, 4, 5, 1027, 3 * 802 * 550 + 1):
        a, b = g.normal(size=n).astype(np.float32), g.normal(size=n).astype(np.float32)
        ta = _t(a, dev, True)
        l = loss.l1_loss(ta, _t(b, dev))
        l.backward()
        assert abs(float(l) - LO.l1(a, b)) < 2e-6 * max(1.0, LO.l1(a, b))
        np.testing.assert_allclose(ta.grad.cpu().numpy(), LO.l1_grad(a, b), rtol=1e-6, atol=0)


def test_densification_stats():
    from gaussianavatars_amd import loss

    dev = _dev()
    mr, acc, dn = _t(PINS["ds_max_in"], dev), _t(PINS["ds_acc_in"], dev), _t(PINS["ds_den_in"], dev)
    loss.densification_stats(_t(PINS["ds_radii"], dev), _t(PINS["ds_vgrad"], dev), mr, acc, dn)
    np.testing.assert_array_equal(mr.cpu().numpy(), PINS["ds_max_out"])
    np.testing.assert_array_equal(dn.cpu().numpy(), PINS["ds_den_out"])
    np.testing.assert_allclose(acc.cpu().numpy(), PINS["ds_acc_out"], rtol=3e-7, atol=0)


def test_training_iteration_statistics_and_loss():
    """Three iterations shaped like train.py:118-198 on a bound model: select -> render -> fused L1+SSIM -> backward ->
    fused statistics; the loss / image gradient / statistics are compared with the same lines written in composed torch
    ops (the reference's formulation) on the same rendered image."""
    import torch.nn.functional as F

    import bench
    from gaussianavatars_amd import loss
    from gaussianavatars_amd.gaussian_renderer import render

    dev = _dev()
    g, cam = bench.build_scene(dev, 20_000, 3, 320, 256, 3, "fused", True)
    bg = torch.ones(3, device=dev)
    gen = torch.Generator(device="cpu").manual_seed(5)
    gt = torch.rand(3, cam.image_height, cam.image_width, generator=gen).to(dev)
    w1 = torch.from_numpy(LO.window_1d()).to(dev)
    win = (w1[:, None] * w1[None, :]).expand(3, 1, 11, 11).contiguous()

    def torch_ssim(a, b):  # utils/loss_utils.py:43-63 written out
        conv = lambda x: F.conv2d(x[None], win, padding=5, groups=3)[0]
        mu1, mu2 = conv(a), conv(b)
        s1, s2, s12 = conv(a * a) - mu1 * mu1, conv(b * b) - mu2 * mu2, conv(a * b) - mu1 * mu2
        return (((2 * mu1 * mu2 + LO.C1) * (2 * s12 + LO.C2)) / ((mu1 * mu1 + mu2 * mu2 + LO.C1) * (s1 + s2 + LO.C2))).mean()

    P = g._xyz.shape[0]
    ref_max, ref_acc, ref_den = torch.zeros(P, device=dev), torch.zeros(P, 1, device=dev), torch.zeros(P, 1, device=dev)
    lam = 0.2
    for it in range(3):
        bench.zero_grads(g)
        g.select_mesh_by_timestep(it)
        pkg = render(cam, g, bench.Pipe, bg)
        image, vsp, radii = pkg["render"], pkg["viewspace_points"], pkg["radii"]
        assert pkg["visibility_filter"].dtype == torch.bool and torch.equal(pkg["visibility_filter"], radii > 0)
        l1, ss = loss.l1_ssim(image, gt)
        total = (1.0 - lam) * l1 + lam * (1.0 - ss)
        # the same loss in torch on a detached copy of the image
        img2 = image.detach().clone().requires_grad_(True)
        total_ref = (1.0 - lam) * (img2 - gt).abs().mean() + lam * (1.0 - torch_ssim(img2, gt))
        total_ref.backward()
        image.retain_grad()
        total.backward()
        assert abs(float(total) - float(total_ref)) < 2e-5
        assert float((image.grad - img2.grad).abs().max()) < 3e-4 * float(img2.grad.abs().max())
        assert g._xyz.grad is not None and float(g._xyz.grad.abs().sum()) > 0
        g.update_densification_stats(vsp, radii)
        vis = radii > 0
        ref_max[vis] = torch.max(ref_max[vis], radii[vis])
        ref_acc[vis] += torch.norm(vsp.grad[vis, :2], dim=-1, keepdim=True)
        ref_den[vis] += 1
    assert int(vis.sum()) > 1000
    assert torch.equal(g.max_radii2D, ref_max) and torch.equal(g.denom, ref_den)
    assert float((g.xyz_gradient_accum - ref_acc).abs().max()) <= 4e-7 * float(ref_acc.abs().max())


def test_seeded_backward_gives_the_gradients_of_the_engines_own_seed():
    """loss.install_backward_seed: `loss.backward()` as train.py:133 writes it, through the arithmetic of train.py:132, gives the gradients of
    torch's own method bit for bit; explicit seeds and retain_graph pass through; uninstalling restores torch's method."""
    from gaussianavatars_amd import loss as L

    DEV = _dev()
    a0 = torch.rand(3, 40, 52, device=DEV)
    b = torch.rand(3, 40, 52, device=DEV)
    L.install_backward_seed(False)
    torch_backward = torch.Tensor.backward

    def run():
        a = a0.clone().requires_grad_(True)
        l1, ss = L.l1_ssim(a, b)
        loss = (1.0 - 0.2) * l1 + 0.2 * (1.0 - ss)
        loss.backward(retain_graph=True)
        g1 = a.grad.clone()
        a.grad = None
        loss.backward(torch.full((), 2.0, device=DEV))
        return loss.detach().clone(), g1, a.grad.clone()

    try:
        v0, g0, h0 = run()
        assert L.install_backward_seed() is True and L.install_backward_seed() is False
        assert torch.Tensor.backward is not torch_backward
        v1, g1, h1 = run()
        assert torch.equal(v0, v1) and torch.equal(g0, g1) and torch.equal(h0, h1) and torch.equal(h1, 2.0 * g1)
        assert (DEV, torch.float32) in L._UNIT_SEEDS
        a = a0.clone().requires_grad_(True)
        L.l1_loss(a, b).backward()
        assert torch.equal(a.grad, torch.sign(a0 - b) / a0.numel())
        c = torch.ones(3, requires_grad=True)                      # CPU tensors and non-scalars are torch's business
        (c * 2).sum().backward()
        assert torch.equal(c.grad, torch.full((3,), 2.0))
        with pytest.raises(RuntimeError):
            (a * 2).backward()
    finally:
        assert L.install_backward_seed(False) is True
    assert torch.Tensor.backward is torch_backward


def test_l1_backward_on_the_unit_seed_launches_nothing_and_equals_the_kernel():
    """`l1_loss(image, gt).backward()` under install_backward_seed: the forward (gls_l1_forward_grad) leaves sign(a - b) / n behind and the backward
    returns it without a launch; the same bits as gls_l1_backward gives.  Any other upstream gradient (a scaled loss, an explicit seed, a second
    backward over a retained graph, no seed installed) takes the kernel.  (The Python host side: the test watches its ctypes calls; the compiled
    host's L1 node is held to the same behaviour by tests/test_native_host_gpu.py.)"""
    from gaussianavatars_amd import _host
    from gaussianavatars_amd import _lib
    from gaussianavatars_amd import loss as L

    DEV = _dev()
    prev_host = _host.set_enabled(False)
    try:
        _l1_unit_seed_body(DEV, _lib, L)
    finally:
        _host.set_enabled(prev_host)


def _l1_unit_seed_body(DEV, _lib, L):
    lib = _lib.gls()
    calls = []
    real_bwd, real_fwd_grad = lib.gls_l1_backward, lib.gls_l1_forward_grad

    class Spy:
        def __init__(self, fn, name):
            self.fn, self.name = fn, name

        def __call__(self, *a):
            calls.append(self.name)
            return self.fn(*a)

    for n in ((3, 802, 550), (3, 37, 41), (5,)):   # BASELINE image; n % 4 != 0 tails
        a0, b = torch.rand(*n, device=DEV), torch.rand(*n, device=DEV)
        a0.view(-1)[:2] = b.view(-1)[:2]            # exact ties: gradient 0
        ref = torch.sign(a0 - b) / a0.numel()
        L.install_backward_seed(False)
        a = a0.clone().requires_grad_(True)
        L.l1_loss(a, b).backward()
        g_kernel = a.grad.clone()
        assert torch.equal(g_kernel, ref)
        assert L.install_backward_seed() is True
        lib.gls_l1_backward, lib.gls_l1_forward_grad = Spy(real_bwd, "bwd"), Spy(real_fwd_grad, "fwd_grad")
        try:
            calls.clear()
            a = a0.clone().requires_grad_(True)
            loss = L.l1_loss(a, b)
            loss.backward(retain_graph=True)
            assert calls == ["fwd_grad"] and torch.equal(a.grad, g_kernel)
            assert abs(float(loss) - float((a0 - b).abs().mean())) < 2e-6
            a.grad = None
            loss.backward()                                        # the image was handed out once: the kernel this time
            assert calls == ["fwd_grad", "bwd"] and torch.equal(a.grad, g_kernel)
            calls.clear()
            a = a0.clone().requires_grad_(True)
            (0.8 * L.l1_loss(a, b)).backward()                     # train.py:132's weighting: not the unit seed
            assert calls == ["fwd_grad", "bwd"] and torch.equal(a.grad, 0.8 * g_kernel)
            calls.clear()
            a = a0.clone().requires_grad_(True)
            L.l1_loss(a, b).backward(torch.full((), 3.0, device=DEV))
            assert calls == ["fwd_grad", "bwd"] and torch.equal(a.grad, 3.0 * g_kernel)
            calls.clear()
            a, b2 = a0.clone().requires_grad_(True), b.clone().requires_grad_(True)
            L.l1_loss(a, b2).backward()                            # both images differentiable: the plain forward
            assert calls == ["bwd", "bwd"] and torch.equal(a.grad, g_kernel) and torch.equal(b2.grad, -g_kernel)
        finally:
            lib.gls_l1_backward, lib.gls_l1_forward_grad = real_bwd, real_fwd_grad
            L.install_backward_seed(False)


def test_zero_edit_pair_l1_then_ssim_becomes_one_fused_pass():
    """What patch_reference() installs on utils.loss_utils (loss.l1_loss_paired / ssim_paired, SURVEY.md 8(f) N3 behind the zero-edit boundary): train.py:131-132
    calls l1_loss(image, gt) and then ssim(image, gt).  First iteration: two stand-alone passes (and the pattern is noted); from the second on l1_loss runs the
    FUSED pass and parks the SSIM scalar for the ssim() call that follows.  Values and the gradient of train.py's weighted sum equal the stand-alone functions';
    a different pair, another window or a CPU of the pattern (three l1_loss calls without an ssim) fall back to the stand-alone kernels."""
    from gaussianavatars_amd import loss as L

    DEV = _dev()
    L._PAIR.update(fused=False, last=None, parked=None, unclaimed=0)
    gt = torch.rand(3, 97, 61, device=DEV)
    want = None
    for it in range(4):
        img = (gt + 0.1 * torch.rand(3, 97, 61, device=DEV)).requires_grad_(True)
        ref_img = img.detach().clone().requires_grad_(True)
        ref = 0.8 * L.l1_loss(ref_img, gt) + 0.2 * (1.0 - L.ssim(ref_img, gt))
        ref.backward()
        l1 = L.l1_loss_paired(img, gt)
        assert L._PAIR["fused"] is (it >= 1)
        if it >= 1:
            assert L._PAIR["parked"] is not None
        ss = L.ssim_paired(img, gt)
        assert L._PAIR["parked"] is None
        loss = 0.8 * l1 + 0.2 * (1.0 - ss)
        loss.backward()
        assert abs(float(loss) - float(ref)) < 1e-6
        err = float((img.grad - ref_img.grad).abs().max()) / float(ref_img.grad.abs().max())
        assert err < 1e-5, (it, err)
    # another image in between: the parked scalar is not handed out for it
    img2 = torch.rand(3, 97, 61, device=DEV)
    l1 = L.l1_loss_paired(img, gt)
    assert L._PAIR["parked"] is not None
    assert abs(float(L.ssim_paired(img2, gt)) - float(L.ssim(img2, gt))) < 1e-7 and L._PAIR["parked"] is None
    # three fused results nobody collects: back to the plain L1 kernel
    for _ in range(4):
        L.l1_loss_paired(img, gt)
    assert L._PAIR["fused"] is False
    L._PAIR.update(fused=False, last=None, parked=None, unclaimed=0)
