"""Views into the opaque state buffers of one forward pass (layouts of include/gsr.h), so tests and
the benchmark can look at every intermediate the reference keeps in geomBuffer / binningBuffer /
imgBuffer.  Not used by the product path."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from .rasterizer import GaussianRasterizationSettings, _RasterizeGaussians


def _view(buf: torch.Tensor, off: int, count: int, dtype: torch.dtype) -> torch.Tensor:
    nbytes = count * torch.empty((), dtype=dtype).element_size()
    return buf[off: off + nbytes].view(dtype)


def forward_state(rs: GaussianRasterizationSettings, means3D, shs, colors_precomp, opacities, scales, rotations,
                  cov3D_precomp, tile_culling: bool = False, fast_blend: bool = False) -> dict:
    """Runs the forward through the autograd Function (same code path as render()) and unpacks the
    saved state.  Returns device tensors.  tile_culling=False (default here) keeps the reference's rect-based
    instance lists so that keys / point_list / ranges / n_contrib compare with the oracle index for index."""
    from . import rasterizer as _R

    prev = _R.set_tile_culling(2 if tile_culling else 0)   # 2: culled, but the sorted lists are still written for inspection
    prev_fast = _R.set_fast_blend(fast_blend)               # default: the exact blend, whose every intermediate equals the oracle's bits
    try:
        return _forward_state(rs, means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp)
    finally:
        _R.set_tile_culling(prev)
        _R.set_fast_blend(prev_fast)


def _forward_state(rs, means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp) -> dict:
    lib = _lib.gsr()
    e = torch.Tensor([])
    args = [means3D, torch.zeros_like(means3D), e if shs is None else shs, e if colors_precomp is None else colors_precomp,
            opacities, e if scales is None else scales, e if rotations is None else rotations,
            e if cov3D_precomp is None else cov3D_precomp]

    class _Ctx:
        needs_input_grad = (True,) * 10   # lay the state out as a training forward does

        def save_for_backward(self, *t):
            self.saved = t

        def mark_non_differentiable(self, *t):
            pass

        def set_materialize_grads(self, flag):
            pass

    ctx = _Ctx()
    with torch.no_grad():
        color, radii, _visible = _RasterizeGaussians.forward(ctx, *args, rs)
    geom, binning, img = ctx.saved[7], ctx.saved[8], ctx.saved[9]
    P = means3D.shape[0]
    H, W = int(rs.image_height), int(rs.image_width)
    I, cap = ctx.num_rendered, ctx.capacity
    gl, bl, il = _lib.GsrGeomLayout(), _lib.GsrBinningLayout(), _lib.GsrImageLayout()
    lib.gsr_geom_layout(P, C.byref(gl))
    from . import rasterizer as _R

    lib.gsr_binning_layout(cap, W, H, P, int(_R.get_tile_culling()), C.byref(bl))
    prod = bl.path == 1
    lists = int(_R.get_tile_culling()) in (0, 2) or bl.path == 2   # the reference-format key / point lists were written
    lib.gsr_image_layout(W, H, C.byref(il))
    tiles = ((W + 15) // 16) * ((H + 15) // 16)
    f32, u32, i16 = torch.float32, torch.int32, torch.int16
    rec = _view(geom, gl.grec, 12 * P, f32).view(P, 12)
    fast = bool(ctx.fast_blend) and not ctx.deterministic and bl.path != 2   # gsr_api.hip: fast_effective
    out = dict(
        color=color, radii=radii, num_rendered=I, capacity=cap,
        rect_instances=int(_view(binning, 8, 1, torch.int64).item()),
        depths=_view(geom, gl.depths, P, f32),
        xy=_view(geom, gl.grec, 12 * P, f32).view(P, 12)[:, 0:2],
        conic_opacity=rec[:, 2:6] if not fast else torch.cat([rec[:, 2:5], rec[:, 10:11]], 1),   # fast-blend record: slot 5 is log2(opacity), the opacity sits in slot 10
        rgb=_view(geom, gl.grec, 12 * P, f32).view(P, 12)[:, 6:9],
        grec=rec,   # the raw 12-float records
        cov3D=_view(geom, gl.cov3D, 6 * P, f32).view(P, 6),
        rect=_view(geom, gl.rect, 4 * P, i16).view(P, 4),
        tiles_touched=_view(geom, gl.tiles_touched, P, u32),
        clamped=_view(geom, gl.clamped, P, torch.uint8),
        production_binning=prod,
        binning_path=int(bl.path),
        nbands=int(bl.nbands), band_rows=int(bl.band_rows),
        band_total=_view(binning, 40, int(bl.nbands) if bl.path == 0 and bl.nbands > 1 else 0, u32),   # binned splats per band of tile rows
        keys=_view(binning, bl.keys, I if lists else 0, torch.int64),
        point_list=_view(binning, bl.point_list, I if lists else 0, u32),
        qlist=_view(binning, bl.qlist, 4 * cap if lists and not prod else 0, u32),           # parity modes: stream entries' positions in the tile list
        qpos=_view(binning, bl.qpos, cap if prod else 4 * cap, u32),           # the quadrant streams of splat indices (from qstart)
        qcount=_view(binning, bl.qcount, 4 * tiles, u32).view(tiles, 4),
        qstart=_view(binning, bl.qstart, 4 * tiles, u32).view(tiles, 4),
        ranges=_view(binning, bl.ranges, 0 if prod else 2 * tiles, u32).view(-1, 2),
        tile_count=_view(binning, bl.tile_count, 0 if prod else tiles, u32),
        final_T=_view(img, il.final_T, H * W, f32).view(H, W),
        n_contrib=_view(img, il.n_contrib, H * W, u32).view(H, W),
        n_contrib_q=_view(img, il.n_contrib_q, H * W, u32).view(H, W),
    )
    return out
