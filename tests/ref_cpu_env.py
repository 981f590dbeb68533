"""TEST INFRASTRUCTURE for running the reference's own entry scripts on a box without a GPU (tests/test_reference_entry_cpu.py).

install() does two things inside the subprocess that runs the script:
  * neutralises CUDA: `.cuda()`, `device="cuda"`, torch.cuda.Event / synchronize / set_device become their host equivalents
    (the reference hard-codes the device, utils/general_utils.py:133, scene/gaussian_model.py:177-211);
  * replaces the rasterizer's autograd Function by the CPU ORACLE (oracle/gsr_oracle.c) -- the checker, used here as a stand-in
    so that the script's frame loop produces real images.  Never importable from the product package (tests/test_abi.py enforces it).
Nothing here is reachable from gaussianavatars_amd/."""
import time

import numpy as np
import torch


def _no_cuda():
    def to_cpu(dev):
        if dev is None:
            return None
        if isinstance(dev, torch.device):
            return torch.device("cpu") if dev.type == "cuda" else dev
        if isinstance(dev, str) and dev.startswith("cuda"):
            return "cpu"
        return dev

    def wrap(fn):
        def inner(*a, **k):
            if "device" in k:
                k["device"] = to_cpu(k["device"])
            return fn(*a, **k)
        inner.__name__ = getattr(fn, "__name__", "wrapped")
        return inner

    for name in ("tensor", "zeros", "ones", "empty", "full", "rand", "randn", "arange", "zeros_like", "ones_like", "empty_like", "full_like",
                 "eye", "linspace", "as_tensor", "randint"):
        setattr(torch, name, wrap(getattr(torch, name)))
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    orig_to = torch.Tensor.to

    def tensor_to(self, *a, **k):
        a = tuple(to_cpu(x) if isinstance(x, (str, torch.device)) else x for x in a)
        if "device" in k:
            k["device"] = to_cpu(k["device"])
        return orig_to(self, *a, **k)
    torch.Tensor.to = tensor_to
    torch.cuda.set_device = lambda *a, **k: None
    torch.cuda.synchronize = lambda *a, **k: None
    torch.cuda.manual_seed_all = lambda *a, **k: None
    torch.cuda.manual_seed = lambda *a, **k: None

    class Event:
        def __init__(self, enable_timing=False):
            self.t = None

        def record(self, stream=None):
            self.t = time.perf_counter()

        def elapsed_time(self, other):
            return 1e3 * (other.t - self.t)
    torch.cuda.Event = Event


class _OracleRasterize(torch.autograd.Function):
    """Same ten arguments and three outputs as gaussianavatars_amd.rasterizer._RasterizeGaussians, computed by the CPU oracle."""

    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, rs, sh_rest=None):
        from oracle import gsr_oracle as O

        n = lambda t: None if t is None or t.numel() == 0 else t.detach().cpu().numpy()
        shs = sh
        if sh_rest is not None and sh_rest.numel():
            shs = torch.cat([sh, sh_rest], 1)
        s = O.make_settings(rs.image_height, rs.image_width, rs.tanfovx, rs.tanfovy, n(rs.bg), rs.scale_modifier, n(rs.viewmatrix), n(rs.projmatrix),
                            rs.sh_degree, n(rs.campos))
        st = O.forward(s, n(means3D), n(shs), n(colors_precomp), n(opacities), n(scales), n(rotations), n(cov3Ds_precomp))
        ctx.mark_non_differentiable
        radii = torch.from_numpy(st.radii.copy())
        _OracleRasterize.calls += 1
        _OracleRasterize.last = st
        return torch.from_numpy(st.color.copy()), radii, radii > 0

    @staticmethod
    def backward(ctx, *g):
        raise RuntimeError("the CPU stand-in of the rasterizer is forward-only")


_OracleRasterize.calls = 0
_OracleRasterize.last = None


def install():
    _no_cuda()
    from gaussianavatars_amd import rasterizer as R

    R._RasterizeGaussians.apply = staticmethod(_OracleRasterize.apply)
    return _OracleRasterize
