"""The four `roma` functions the reference calls (scene/flame_gaussian_model.py:147, scene/gaussian_model.py:137), with
roma's conventions: quaternions are XYZW, batch shape free, differentiable.  (roma is a PyPI package that is not installed
here; semantics per SURVEY.md Appendix B, `rotmat_to_unitquat` cross-checked against SciPy in tests/test_binding_cpu.py.)
The arithmetic lives in gaussianavatars_amd.unfused (the composed-torch statement of the binding half)."""
from ..unfused import quat_product, quat_wxyz_to_xyzw, quat_xyzw_to_wxyz, rotmat_to_unitquat  # noqa: F401

__all__ = ["rotmat_to_unitquat", "quat_xyzw_to_wxyz", "quat_wxyz_to_xyzw", "quat_product"]
