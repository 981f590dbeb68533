"""`nvdiffrast` stand-in: the reference imports it eagerly (mesh_renderer/__init__.py:10, instantiated at train.py:40 and
render.py:33) but uses it only for the `--render_mesh` debug overlay, which is off the splat hot path."""
from . import torch  # noqa: F401
