"""ctypes/numpy driver for oracle/libgsr_oracle.so (the CPU restatement of the rasterizer).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg, never by the product package.  See gsr_oracle.c for the parity status
("parity unpinned" against the CUDA rasterizer; pinned pieces listed there).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libgsr_oracle.so")


class OraSettings(C.Structure):
    _fields_ = [
        ("image_height", C.c_int32),
        ("image_width", C.c_int32),
        ("tanfovx", C.c_float),
        ("tanfovy", C.c_float),
        ("bg", C.c_float * 3),
        ("scale_modifier", C.c_float),
        ("viewmatrix", C.c_float * 16),
        ("projmatrix", C.c_float * 16),
        ("sh_degree", C.c_int32),
        ("campos", C.c_float * 3),
        ("prefiltered", C.c_int32),
        ("debug", C.c_int32),
        ("exact_scale_grad", C.c_int32),
    ]


def build(force: bool = False) -> str:
    """Compile the oracle with its own Makefile (gcc); returns the .so path."""
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(
        os.path.join(_HERE, "gsr_oracle.c")
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.ora_expf.restype = C.c_float
        _lib.ora_expf.argtypes = [C.c_float]
        _lib.ora_scan.restype = C.c_int64
        _lib.ora_set_threads.restype = C.c_int
        _lib.ora_set_threads.argtypes = [C.c_int]
    return _lib


def _p(a: Optional[np.ndarray]):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


def make_settings(H, W, tanfovx, tanfovy, bg, scale_modifier, viewmatrix, projmatrix, sh_degree, campos,
                  prefiltered=False, debug=False, exact_scale_grad=False) -> OraSettings:
    s = OraSettings()
    s.image_height, s.image_width = int(H), int(W)
    s.tanfovx, s.tanfovy = float(tanfovx), float(tanfovy)
    s.bg[:] = [float(v) for v in np.asarray(bg, np.float32).reshape(3)]
    s.scale_modifier = float(scale_modifier)
    s.viewmatrix[:] = [float(v) for v in np.asarray(viewmatrix, np.float32).reshape(16)]
    s.projmatrix[:] = [float(v) for v in np.asarray(projmatrix, np.float32).reshape(16)]
    s.sh_degree = int(sh_degree)
    s.campos[:] = [float(v) for v in np.asarray(campos, np.float32).reshape(3)]
    s.prefiltered, s.debug = int(prefiltered), int(debug)
    s.exact_scale_grad = int(exact_scale_grad)   # default: upstream's dL/dscale (no scale_modifier factor)
    return s


@dataclass
class OracleState:
    """Every intermediate of one forward pass (what the HIP path is compared against)."""
    P: int = 0
    M: int = 0
    depths: np.ndarray = None
    xy: np.ndarray = None
    cov3D: np.ndarray = None
    conic_opacity: np.ndarray = None
    rgb: np.ndarray = None
    radii: np.ndarray = None
    tiles_touched: np.ndarray = None
    rect: np.ndarray = None
    clamped: np.ndarray = None
    offsets: np.ndarray = None
    num_rendered: int = 0
    keys: np.ndarray = None
    point_list: np.ndarray = None
    ranges: np.ndarray = None
    final_T: np.ndarray = None
    n_contrib: np.ndarray = None
    color: np.ndarray = None
    inputs: dict = field(default_factory=dict)


def _f32(a):
    return None if a is None else np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def forward(settings: OraSettings, means3D, shs, colors_precomp, opacities, scales, rotations,
            cov3D_precomp) -> OracleState:
    """A.1-A.3 of SURVEY.md Appendix A.  `shs` is (P, M, 3)."""
    L = lib()
    means3D = _f32(means3D)
    P = means3D.shape[0]
    shs = _f32(shs)
    colors_precomp = _f32(colors_precomp)
    opacities = _f32(opacities).reshape(-1)
    scales, rotations, cov3D_precomp = _f32(scales), _f32(rotations), _f32(cov3D_precomp)
    if (shs is None) == (colors_precomp is None):
        raise Exception("Please provide excatly one of either SHs or precomputed colors!")
    if ((scales is None or rotations is None) and cov3D_precomp is None) or (
        (scales is not None or rotations is not None) and cov3D_precomp is not None
    ):
        raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
    M = 0 if shs is None else shs.shape[1]
    H, W = settings.image_height, settings.image_width
    st = OracleState(P=P, M=M)
    st.inputs = dict(means3D=means3D, shs=shs, colors_precomp=colors_precomp, opacities=opacities,
                     scales=scales, rotations=rotations, cov3D_precomp=cov3D_precomp)
    st.depths = np.zeros(P, np.float32)
    st.xy = np.zeros((P, 2), np.float32)
    st.cov3D = np.zeros((P, 6), np.float32)
    st.conic_opacity = np.zeros((P, 4), np.float32)
    st.rgb = np.zeros((P, 3), np.float32)
    st.radii = np.zeros(P, np.int32)
    st.tiles_touched = np.zeros(P, np.uint32)
    st.rect = np.zeros((P, 4), np.int32)
    st.clamped = np.zeros((P, 3), np.uint8)
    L.ora_preprocess(C.byref(settings), C.c_int(P), C.c_int(M), _p(means3D), _p(shs), _p(colors_precomp),
                     _p(opacities), _p(scales), _p(rotations), _p(cov3D_precomp), _p(st.depths), _p(st.xy),
                     _p(st.cov3D), _p(st.conic_opacity), _p(st.rgb), _p(st.radii), _p(st.tiles_touched),
                     _p(st.rect), _p(st.clamped))
    st.offsets = np.zeros(max(P, 1), np.uint32)
    st.num_rendered = int(L.ora_scan(C.c_int(P), _p(st.tiles_touched), _p(st.offsets)))
    I = st.num_rendered
    gx, gy = (W + 15) // 16, (H + 15) // 16
    st.keys = np.zeros(max(I, 1), np.uint64)
    st.point_list = np.zeros(max(I, 1), np.uint32)
    st.ranges = np.zeros((gx * gy, 2), np.uint32)
    L.ora_bin(C.byref(settings), C.c_int(P), _p(st.depths), _p(st.radii), _p(st.rect), _p(st.offsets),
              C.c_int64(I), _p(st.keys), _p(st.point_list), _p(st.ranges))
    st.keys, st.point_list = st.keys[:I], st.point_list[:I]
    st.final_T = np.zeros((H, W), np.float32)
    st.n_contrib = np.zeros((H, W), np.uint32)
    st.color = np.zeros((3, H, W), np.float32)
    pl = st.point_list if I > 0 else np.zeros(1, np.uint32)
    L.ora_render_forward(C.byref(settings), _p(st.ranges), _p(pl), _p(st.xy), _p(st.rgb),
                         _p(st.conic_opacity), _p(st.final_T), _p(st.n_contrib), _p(st.color))
    return st


def backward(settings: OraSettings, st: OracleState, dL_dpix) -> dict:
    """A.4: returns the gradient dict in the layout the reference's backward returns."""
    L = lib()
    P, M = st.P, st.M
    dL_dpix = _f32(dL_dpix)
    g_mean2D = np.zeros((P, 2), np.float64)
    g_conic = np.zeros((P, 3), np.float64)
    g_opac = np.zeros(P, np.float64)
    g_color = np.zeros((P, 3), np.float64)
    pl = st.point_list if st.num_rendered > 0 else np.zeros(1, np.uint32)
    L.ora_render_backward(C.byref(settings), C.c_int(P), _p(st.ranges), _p(pl), _p(st.xy),
                          _p(st.conic_opacity), _p(st.rgb), _p(st.final_T), _p(st.n_contrib), _p(dL_dpix),
                          _p(g_mean2D), _p(g_conic), _p(g_opac), _p(g_color))
    inp = st.inputs
    use_pre_cov = inp["cov3D_precomp"] is not None
    use_pre_col = inp["colors_precomp"] is not None
    g_mean2D32, g_conic32, g_color32 = (g_mean2D.astype(np.float32), g_conic.astype(np.float32),
                                         g_color.astype(np.float32))
    g_means3D = np.zeros((P, 3), np.float32)
    g_cov3D = np.zeros((P, 6), np.float32)
    g_sh = np.zeros((P, max(M, 1), 3), np.float32)
    g_scale = np.zeros((P, 3), np.float32)
    g_rot = np.zeros((P, 4), np.float32)
    L.ora_preprocess_backward(C.byref(settings), C.c_int(P), C.c_int(M), _p(inp["means3D"]), _p(inp["shs"]),
                              _p(inp["scales"]), _p(inp["rotations"]), _p(st.cov3D), C.c_int(use_pre_cov),
                              C.c_int(use_pre_col), _p(st.radii), _p(st.clamped), _p(g_mean2D32), _p(g_conic32),
                              _p(g_color32), _p(g_means3D), _p(g_cov3D), _p(g_sh) if M > 0 else None,
                              _p(g_scale), _p(g_rot))
    means2D = np.zeros((P, 3), np.float32)
    means2D[:, :2] = g_mean2D32
    return dict(
        means3D=g_means3D,
        means2D=means2D,
        shs=g_sh[:, :M] if M > 0 else None,
        colors_precomp=g_color32 if use_pre_col else None,
        opacities=g_opac.astype(np.float32).reshape(P, 1),
        scales=None if use_pre_cov else g_scale,
        rotations=None if use_pre_cov else g_rot,
        cov3D_precomp=g_cov3D if use_pre_cov else None,
        _conic=g_conic32,
        _color=g_color32,
        _cov3D=g_cov3D,
    )


def set_threads(n: int) -> int:
    """OpenMP threads for the oracle's parallel loops from now on (0: keep the default); returns the count in use."""
    return int(lib().ora_set_threads(int(n)))


def expf(x: np.ndarray) -> np.ndarray:
    L = lib()
    x = np.asarray(x, np.float32)
    return np.array([L.ora_expf(C.c_float(float(v))) for v in x.reshape(-1)], np.float32).reshape(x.shape)
