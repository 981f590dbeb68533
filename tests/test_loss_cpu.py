"""CPU tests of the loss / statistics row: the numpy oracle against vectors produced by the
reference's own utils/loss_utils.py and GaussianModel.add_densification_stats (tests/golden/make_golden.py), and the
C-ABI library's exports / host-side argument checks.  No compute calls on a device."""
import os

import numpy as np
import pytest

from oracle import loss_oracle as LO

HERE = os.path.dirname(os.path.abspath(__file__))
PINS = np.load(os.path.join(HERE, "golden", "loss_pins.npz"))


def test_window_matches_reference():
    np.testing.assert_allclose(LO.window_2d(), PINS["window_2d"], rtol=0, atol=3e-8)  # <= 2 ulp: fp32 summation order of the normaliser
    assert abs(float(LO.window_1d().sum()) - 1.0) < 1e-6


@pytest.mark.parametrize("case", ["chw", "bchw", "tiny"])
def test_oracle_values_and_gradients_match_reference(case):
    a, b = PINS[f"{case}_a"], PINS[f"{case}_b"]
    # the pins are fp32 torch results; the oracle is fp64: tolerances are fp32 round-off of the reference
    assert abs(LO.l1(a, b) - float(PINS[f"{case}_l1"])) < 2e-6
    assert abs(LO.ssim(a, b) - float(PINS[f"{case}_ssim"])) < 2e-5
    np.testing.assert_allclose(LO.l1_grad(a, b), PINS[f"{case}_g_l1"], rtol=1e-5, atol=1e-9)
    g, ref = LO.ssim_grad(a, b), PINS[f"{case}_g_ssim"]
    assert np.abs(g - ref).max() < 2e-4 * np.abs(ref).max()
    if a.ndim == 4:
        np.testing.assert_allclose(LO.ssim(a, b, size_average=False), PINS[f"{case}_ssim_per_image"], rtol=0, atol=2e-5)


def test_ssim_gradient_is_the_derivative():
    """finite differences in fp64 on the oracle itself (independent of the pins)"""
    g = np.random.default_rng(1)
    a, b = g.uniform(0, 1, (1, 13, 14)), g.uniform(0, 1, (1, 13, 14))
    an = LO.ssim_grad(a, b)
    for (c, y, x) in [(0, 0, 0), (0, 6, 7), (0, 12, 13), (0, 3, 11)]:
        e = np.zeros_like(a)
        e[c, y, x] = 1e-6
        fd = (LO.ssim(a + e, b) - LO.ssim(a - e, b)) / 2e-6
        assert abs(fd - an[c, y, x]) < 1e-6 * max(1.0, abs(fd) * 1e3)


def test_densification_stats_match_reference():
    mr, acc, dn = LO.densification_stats(PINS["ds_radii"], PINS["ds_vgrad"], PINS["ds_max_in"], PINS["ds_acc_in"].ravel(),
                                         PINS["ds_den_in"].ravel())
    np.testing.assert_array_equal(mr, PINS["ds_max_out"])
    np.testing.assert_array_equal(dn, PINS["ds_den_out"].ravel())
    np.testing.assert_allclose(acc, PINS["ds_acc_out"].ravel(), rtol=2e-7, atol=0)


def test_gls_library_exports_every_declared_symbol():
    import re

    from gaussianavatars_amd import _lib

    root = os.path.dirname(HERE)
    txt = re.sub(r"/\*.*?\*/", "", open(os.path.join(root, "include", "gls.h")).read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(gls_[a-z0-9_]+)\s*\(", txt)))
    lib = _lib.gls()
    assert len(names) == 15
    for n in names:
        assert hasattr(lib, n), f"include/gls.h declares {n} but libgls_hip.so does not export it"
        assert n in _lib.GLS_SYMBOLS
    assert lib.gls_abi_version() == 4
    assert lib.gls_partial_floats(1, 3, 802, 550) == 2 * 51 * 18 * 3   # (32 x 16 tiles since round 4)
    # argument errors are reported before any device work
    assert lib.gls_l1_ssim_forward(1, 3, 0, 5, None, None, 1.0, None, None, None, None) < 0
    assert b"bad image shape" in lib.gls_last_error()
    assert lib.gls_l1_forward_grad(16, None, None, 1.0, None, None, None, None) < 0   # (ABI 2: the forward that also leaves the gradient image)
    assert lib.gls_l1_ssim_forward(1, 3, 8, 8, None, None, 1.0, None, None, None, None) < 0 and b"null" in lib.gls_last_error()


def test_product_path_refuses_cpu_tensors():
    import torch

    from gaussianavatars_amd import loss

    with pytest.raises(RuntimeError, match="device tensor"):
        loss.l1_loss(torch.zeros(3, 4, 4), torch.zeros(3, 4, 4))
    with pytest.raises(NotImplementedError):
        loss.ssim(torch.zeros(3, 4, 4), torch.zeros(3, 4, 4), window_size=7)
