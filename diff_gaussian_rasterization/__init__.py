"""Import-compatible stand-in for the reference's `diff_gaussian_rasterization` package
(gaussian_renderer/__init__.py:15 does `from diff_gaussian_rasterization import
GaussianRasterizationSettings, GaussianRasterizer`).  Put the repository root on PYTHONPATH and the
reference's renderer runs on the MI355X-native kernels unchanged."""
from gaussianavatars_amd.rasterizer import (  # noqa: F401
    GaussianRasterizationSettings,
    GaussianRasterizer,
    rasterize_gaussians,
)
