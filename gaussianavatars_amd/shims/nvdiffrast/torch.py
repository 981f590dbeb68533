"""Contexts construct (NVDiffRenderer() is built at import time of train.py / render.py); rasterising raises."""


class _Ctx:
    def __init__(self, *a, **k):
        pass


class RasterizeCudaContext(_Ctx):
    pass


class RasterizeGLContext(_Ctx):
    pass


def _unavailable(name):
    def f(*a, **k):
        raise RuntimeError(f"nvdiffrast.torch.{name}: the mesh overlay (--render_mesh) needs the real nvdiffrast, which is CUDA-only; "
                           "the splat rendering path does not use it")
    f.__name__ = name
    return f


rasterize = _unavailable("rasterize")
interpolate = _unavailable("interpolate")
antialias = _unavailable("antialias")
texture = _unavailable("texture")
