"""CPU oracle (TEST INFRASTRUCTURE ONLY -- never imported by the product path) for the training-step
neighbours of SURVEY.md 8(f) N3: numpy fp64 restatements of

    l1_loss                   /root/reference/utils/loss_utils.py:17-18
    gaussian / create_window  utils/loss_utils.py:23-31   (fp32 1-D Gaussian, fp32 outer product)
    ssim / _ssim              utils/loss_utils.py:36-63   (zero-padded depthwise 11x11 correlation)
    max_radii2D update        train.py:197
    add_densification_stats   scene/gaussian_model.py:517-519

Parity status: PINNED.  Unlike the rasterizer, the reference source of these functions is in the
snapshot and runs on torch-CPU; tests/golden/make_golden.py imports it, evaluates values and autograd
gradients on seeded images and commits them as tests/golden/loss_pins.npz; tests/test_loss_cpu.py holds
this restatement to those vectors.
"""
from __future__ import annotations

import math

import numpy as np
from scipy.signal import correlate2d

C1 = 0.01 ** 2
C2 = 0.03 ** 2


def window_1d(window_size: int = 11, sigma: float = 1.5) -> np.ndarray:
    """loss_utils.py:23-25 -- the values live in a float32 tensor and are normalised in float32."""
    g = np.array([math.exp(-(x - window_size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(window_size)], np.float32)
    return (g / g.sum(dtype=np.float32)).astype(np.float32)


def window_2d(window_size: int = 11) -> np.ndarray:
    """loss_utils.py:27-31 -- fp32 outer product."""
    g = window_1d(window_size)
    return (g[:, None] * g[None, :]).astype(np.float32)


def _conv(planes: np.ndarray, w: np.ndarray) -> np.ndarray:
    """F.conv2d(x, window, padding=5, groups=C): per-plane correlation with zero fill."""
    out = np.empty_like(planes)
    for idx in np.ndindex(planes.shape[:-2]):
        out[idx] = correlate2d(planes[idx], w, mode="same", boundary="fill", fillvalue=0.0)
    return out


def l1(a: np.ndarray, b: np.ndarray) -> float:
    return float(np.abs(a.astype(np.float64) - b.astype(np.float64)).mean())


def l1_grad(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """d mean|a-b| / d a  (torch: sign(0) = 0)"""
    return np.sign(a.astype(np.float64) - b.astype(np.float64)) / a.size


def ssim_terms(img1: np.ndarray, img2: np.ndarray):
    x, y = img1.astype(np.float64), img2.astype(np.float64)
    w = window_2d().astype(np.float64)
    mu1, mu2 = _conv(x, w), _conv(y, w)
    e11, e22, e12 = _conv(x * x, w), _conv(y * y, w), _conv(x * y, w)
    s1, s2, s12 = e11 - mu1 * mu1, e22 - mu2 * mu2, e12 - mu1 * mu2
    A, B = 2 * mu1 * mu2 + C1, 2 * s12 + C2
    Cd, D = mu1 * mu1 + mu2 * mu2 + C1, s1 + s2 + C2
    return dict(x=x, y=y, w=w, mu1=mu1, mu2=mu2, A=A, B=B, C=Cd, D=D, map=A * B / (Cd * D))


def ssim(img1: np.ndarray, img2: np.ndarray, size_average: bool = True):
    """loss_utils.py:43-63.  Inputs (C,H,W) or (B,C,H,W)."""
    m = ssim_terms(img1, img2)["map"]
    if size_average or m.ndim == 3:
        return float(m.mean())
    return m.mean(axis=(1, 2, 3))


def ssim_grad(img1: np.ndarray, img2: np.ndarray) -> np.ndarray:
    """d ssim(img1, img2).mean() / d img1, by the chain rule through the five windowed moments (the window is
    symmetric, so the adjoint of the correlation is the same correlation)."""
    t = ssim_terms(img1, img2)
    x, y, w, mu1, mu2, A, B, Cd, D, m = (t[k] for k in ("x", "y", "w", "mu1", "mu2", "A", "B", "C", "D", "map"))
    inv = 1.0 / (Cd * D)
    dm_dmu1 = (2 * mu2 * (B - A) - m * 2 * mu1 * (D - Cd)) * inv
    dm_de11 = -m / D
    dm_de12 = 2 * A * inv
    g = _conv(dm_dmu1, w) + 2 * x * _conv(dm_de11, w) + y * _conv(dm_de12, w)
    return g / m.size


def densification_stats(radii, viewspace_grad, max_radii2D, xyz_gradient_accum, denom):
    """train.py:197 + gaussian_model.py:517-519 with update_filter = radii > 0; returns the three updated arrays."""
    vis = radii > 0
    mr = max_radii2D.astype(np.float32).copy()
    acc = xyz_gradient_accum.astype(np.float32).copy()
    dn = denom.astype(np.float32).copy()
    mr[vis] = np.maximum(mr[vis], radii[vis].astype(np.float32))
    g = viewspace_grad[vis, :2].astype(np.float64)
    acc[vis] = (acc[vis].astype(np.float64) + np.sqrt((g * g).sum(-1))).astype(np.float32)
    dn[vis] += 1.0
    return mr, acc, dn
