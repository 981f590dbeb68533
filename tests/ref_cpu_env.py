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
    """Same ten arguments and three outputs as gaussianavatars_amd.rasterizer._RasterizeGaussians, computed by the CPU oracle -- forward
    (oracle.forward) and, since round 4, backward (oracle.backward: what lets the reference's train.py run its loss.backward(), the
    densification statistics and the optimiser step on this box)."""

    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, rs, sh_rest=None):
        from oracle import gsr_oracle as O

        n = lambda t: None if t is None or t.numel() == 0 else t.detach().cpu().numpy()
        shs = sh
        split = sh_rest is not None and sh_rest.numel() > 0
        if split:
            shs = torch.cat([sh, sh_rest], 1)
        s = O.make_settings(rs.image_height, rs.image_width, rs.tanfovx, rs.tanfovy, n(rs.bg), rs.scale_modifier, n(rs.viewmatrix), n(rs.projmatrix),
                            rs.sh_degree, n(rs.campos))
        st = O.forward(s, n(means3D), n(shs), n(colors_precomp), n(opacities), n(scales), n(rotations), n(cov3Ds_precomp))
        radii = torch.from_numpy(st.radii.copy())
        visible = radii > 0
        ctx.mark_non_differentiable(radii, visible)
        ctx.set_materialize_grads(False)
        ctx.oracle = (O, s, st, split, int(sh.shape[1]) if sh.numel() else 0)
        _OracleRasterize.calls += 1
        _OracleRasterize.last = st
        return torch.from_numpy(st.color.copy()), radii, visible

    @staticmethod
    def backward(ctx, g_color, _g_radii=None, _g_visible=None):
        if g_color is None:
            return (None,) * 10
        O, s, st, split, m_dc = ctx.oracle
        g = O.backward(s, st, g_color.detach().cpu().contiguous().numpy())
        t = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a))
        g_sh, g_rest = t(g["shs"]), None
        if split and g_sh is not None:
            g_sh, g_rest = g_sh[:, :m_dc].contiguous(), g_sh[:, m_dc:].contiguous()
        _OracleRasterize.backwards += 1
        return (t(g["means3D"]), t(g["means2D"]), g_sh, t(g["colors_precomp"]), t(g["opacities"]), t(g["scales"]), t(g["rotations"]),
                t(g["cov3D_precomp"]), None, g_rest)


_OracleRasterize.calls = 0
_OracleRasterize.backwards = 0
_OracleRasterize.last = None


def install():
    _no_cuda()
    from gaussianavatars_amd import rasterizer as R

    R._RasterizeGaussians.apply = staticmethod(_OracleRasterize.apply)
    return _OracleRasterize
