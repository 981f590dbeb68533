"""`simple_knn` stand-in (un-vendored submodule, .gitmodules:1-3): only `_C.distCUDA2` is used, once, at the init of an
un-bound model (scene/gaussian_model.py:23,190-192).  Mesh-bound avatars never reach it (:193-194)."""
from . import _C  # noqa: F401
