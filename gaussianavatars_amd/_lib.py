"""ctypes loader for the HIP shared libraries (C ABI of include/*.h).

There is deliberately NO fallback: if the library is missing or does not export a symbol the
import-time error says so.  The CPU oracle under oracle/ is test infrastructure and is never
reachable from here.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))


# ---- host-side fast paths (the frame loop is ~0.7 ms of Python; these run a dozen times per frame) -----------------
def raw_stream(dev):
    """torch's current stream on `dev` as a c_void_p, without building a torch.cuda.Stream object."""
    import torch

    try:
        return torch._C._cuda_getCurrentRawStream(dev.index if dev.index is not None else torch.cuda.current_device())
    except AttributeError:   # very old / very new torch: the public route
        return torch.cuda.current_stream(dev).cuda_stream


class _NoCtx:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


_NOCTX = _NoCtx()


def on_device(dev):
    """Context that makes `dev` the current device for the native call; free when it already is (the usual case)."""
    import torch

    global _ONE_DEVICE
    if _ONE_DEVICE is None:
        _ONE_DEVICE = torch.cuda.device_count() == 1
    if _ONE_DEVICE or dev.index is None or torch.cuda.current_device() == dev.index:
        return _NOCTX
    return torch.cuda.device(dev)


_ONE_DEVICE = None   # a process that sees one GPU never switches devices
# GSR_LIB overrides the library file (kernel experiments build variants side by side: tools/exp_build.sh); the product default is
# the in-tree build.  An override that does not exist is an error, never a fallback.
GSR_LIB_PATH = os.environ.get("GSR_LIB") or os.path.join(_HERE, "libgsr_hip.so")

GSR_OK = 0
GSR_ABI_VERSION = 11
GSR_E_CAPACITY = 1
GSR_COUNT_SLOTS = 128   # include/gsr.h: persistent instance-count slots of the deferred forwards


class GsrSettings(C.Structure):
    """include/gsr.h: GsrSettings"""
    _fields_ = [
        ("image_height", C.c_int32),
        ("image_width", C.c_int32),
        ("tanfovx", C.c_float),
        ("tanfovy", C.c_float),
        ("bg", C.c_void_p),
        ("scale_modifier", C.c_float),
        ("viewmatrix", C.c_void_p),
        ("projmatrix", C.c_void_p),
        ("sh_degree", C.c_int32),
        ("campos", C.c_void_p),
        ("prefiltered", C.c_int32),
        ("debug", C.c_int32),
        ("tile_culling", C.c_int32),
        ("forward_only", C.c_int32),
        ("deterministic", C.c_int32),
        ("exact_scale_grad", C.c_int32),
        ("deferred_count", C.c_int32),
        ("fast_blend", C.c_int32),
    ]


class GsrGeomLayout(C.Structure):
    _fields_ = [(n, C.c_size_t) for n in
                ("depths", "grec", "cov3D", "rect", "tiles_touched", "clamped", "visible", "brec", "acc64", "acc", "total")]


class GsrBinningLayout(C.Structure):
    _fields_ = [(n, C.c_size_t) for n in
                ("keys", "point_list", "qlist", "qpos", "qcount", "qstart", "ranges", "tile_count", "tile_start", "tile_cursor", "tile_order",
                 "block_hist", "dkeys", "dtmp", "order", "bcount", "bstart", "bcursor", "border", "bhist", "qhist", "qprefix", "qmask", "ranks", "rank", "rank_over", "srect", "sspan", "pstat", "tdesc", "obs", "bandcnt", "path", "chunks", "nb", "nbands", "band_rows",
                 "total")]


class GsrBound(C.Structure):
    """include/gsr.h: GsrBound -- the per-face frames of a mesh-bound model for the rasterizer's bound entry"""
    _fields_ = [("binding", C.c_void_p), ("binding_is_i64", C.c_int32), ("F", C.c_int32), ("face_R", C.c_void_p), ("face_scale", C.c_void_p),
                ("face_center", C.c_void_p), ("face_quat", C.c_void_p), ("slot", C.c_void_p), ("rows", C.c_void_p)]


class GsrImageLayout(C.Structure):
    _fields_ = [(n, C.c_size_t) for n in ("final_T", "n_contrib", "n_contrib_q", "c_final", "ck", "gmax", "units", "total")]


#: every symbol include/gsr.h declares -> (restype, argtypes)
GSR_SYMBOLS = {
    "gsr_abi_version": (C.c_int, []),
    "gsr_last_error": (C.c_char_p, []),
    "gsr_count_slot_read": (C.c_int, [C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "gsr_count_slot_overflow": (C.c_int, [C.c_int32, C.POINTER(C.c_int64), C.c_int32]),
    "gsr_last_forward_seq": (C.c_int64, []),
    "gsr_count_slot_wait": (C.c_int, [C.c_int32, C.c_int64, C.c_void_p, C.POINTER(C.c_int64)]),
    "gsr_geom_layout": (C.c_int, [C.c_int32, C.POINTER(GsrGeomLayout)]),
    "gsr_binning_layout": (C.c_int, [C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(GsrBinningLayout)]),
    "gsr_image_layout": (C.c_int, [C.c_int32, C.c_int32, C.POINTER(GsrImageLayout)]),
    "gsr_forward": (C.c_int, [C.POINTER(GsrSettings), C.c_int32, C.c_int32] + [C.c_void_p] * 7 +
                    [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                     C.POINTER(C.c_int64), C.c_void_p]),
    "gsr_backward": (C.c_int, [C.POINTER(GsrSettings), C.c_int32, C.c_int32] + [C.c_void_p] * 6 +
                     [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p] +
                     [C.c_void_p] * 8 + [C.c_void_p]),
    "gsr_forward_ex": (C.c_int, [C.POINTER(GsrSettings), C.c_int32, C.c_int32] + [C.c_void_p] * 8 +
                       [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                        C.POINTER(C.c_int64), C.c_void_p]),
    "gsr_backward_ex": (C.c_int, [C.POINTER(GsrSettings), C.c_int32, C.c_int32] + [C.c_void_p] * 7 +
                        [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p] +
                        [C.c_void_p] * 9 + [C.c_void_p]),
    "gsr_forward_bound": (C.c_int, [C.POINTER(GsrSettings), C.c_int32, C.c_int32, C.POINTER(GsrBound)] + [C.c_void_p] * 6 +
                          [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.POINTER(C.c_int64), C.c_void_p]),
    "gsr_backward_bound": (C.c_int, [C.POINTER(GsrSettings), C.c_int32, C.c_int32, C.POINTER(GsrBound)] + [C.c_void_p] * 6 +
                           [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p] +
                           [C.c_void_p] * 8 + [C.c_void_p]),
    "gsr_mark_visible": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gsr_profile_enable": (C.c_int, [C.c_int]),
    "gsr_profile_read": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "gsr_kernel_name": (C.c_char_p, [C.c_int]),
    "gsr_wait_stats": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
}
GSR_NUM_KERNELS = 12


def gsr_profile_enable(on: bool) -> None:
    gsr().gsr_profile_enable(1 if on else 0)


def gsr_profile_read() -> dict:
    """{kernel name: (total_ms, launches)} accumulated since the last read."""
    ms = (C.c_double * GSR_NUM_KERNELS)()
    n = (C.c_int64 * GSR_NUM_KERNELS)()
    rc = gsr().gsr_profile_read(ms, n)
    if rc != GSR_OK:
        raise RuntimeError(f"gsr_profile_read failed: {gsr().gsr_last_error().decode()}")
    return {gsr().gsr_kernel_name(i).decode(): (ms[i], n[i]) for i in range(GSR_NUM_KERNELS)}

def gsr_wait_stats():
    """(total host wait in ms, number of waits) since the last call -- see include/gsr.h."""
    ms, n = C.c_double(), C.c_int64()
    gsr().gsr_wait_stats(C.byref(ms), C.byref(n))
    return ms.value, n.value


_gsr = None


def _torch_first():
    """The libraries take raw device pointers of torch tensors and launch on torch's streams, so they have to share torch's HIP
    runtime: torch is imported before the first library is mapped.  (Mapped first -- e.g. build() followed by smoke() in one
    process -- they pull in /opt/rocm's libamdhip64 ahead of the copy torch ships, and the first runtime call that needs the
    device fails with 'no ROCm-capable device is detected'.)"""
    import torch  # noqa: F401


def gsr():
    """The rasterizer library; raises (never falls back) when it is not built."""
    global _gsr
    if _gsr is None:
        if not os.path.exists(GSR_LIB_PATH):
            raise RuntimeError(
                f"{GSR_LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback."
            )
        _torch_first()
        lib = C.CDLL(GSR_LIB_PATH)
        for name, (res, args) in GSR_SYMBOLS.items():
            fn = getattr(lib, name)  # AttributeError if the ABI is incomplete
            fn.restype = res
            fn.argtypes = args
        if lib.gsr_abi_version() != GSR_ABI_VERSION:
            raise RuntimeError(f"gsr ABI version {lib.gsr_abi_version()} != {GSR_ABI_VERSION}")
        _gsr = lib
    return _gsr


# ------------------------------------------------------------------------------------------------
# libgab_hip.so : FLAME / face-frame / splat binding kernels (include/gab.h)
# ------------------------------------------------------------------------------------------------
GAB_LIB_PATH = os.environ.get("GAB_LIB") or os.path.join(_HERE, "libgab_hip.so")   # (GAB_LIB: experiment builds, as GSR_LIB)
GAB_FLAME_WS_FLOATS = 512
GAB_BIND_ROW_FLOATS = 20   # include/gab.h: floats per splat of the two-pass CSR backward's scratch
_P = C.c_void_p


class GabRig(C.Structure):
    """include/gab.h: GabRig"""
    _fields_ = [("V", C.c_int32), ("n_shape", C.c_int32), ("n_expr", C.c_int32), ("v_template", _P), ("shapedirs", _P),
                ("posedirs", _P), ("J_regressor", _P), ("lbs_weights", _P), ("parents", C.c_int32 * 5)]


GAB_SYMBOLS = {
    "gab_abi_version": (C.c_int, []),
    "gab_last_error": (C.c_char_p, []),
    "gab_flame_forward": (C.c_int, [C.POINTER(GabRig)] + [_P] * 8 + [_P, _P, _P, _P]),
    "gab_flame_backward": (C.c_int, [C.POINTER(GabRig)] + [_P] * 8 + [_P, _P, _P, _P] + [_P] * 8 + [_P] +
                           [C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_int32), _P]),
    "gab_flame_prepared_floats": (C.c_int64, [C.POINTER(GabRig)]),
    "gab_flame_prepare": (C.c_int, [C.POINTER(GabRig), _P, _P, _P, _P]),
    "gab_flame_forward_prepared": (C.c_int, [C.POINTER(GabRig), _P] + [_P] * 6 + [_P, _P, _P, _P]),
    "gab_blend_sequence": (C.c_int, [C.POINTER(GabRig), _P, _P, C.c_int32, _P, _P]),
    "gab_flame_forward_sequence": (C.c_int, [C.POINTER(GabRig), _P, _P] + [_P] * 6 + [_P, _P, _P, _P]),
    "gab_flame_backward_prepared": (C.c_int, [C.POINTER(GabRig), _P] + [_P] * 4 + [_P, _P, _P] + [_P] * 6 + [_P] +
                                    [C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_int32), _P]),
    "gab_mesh_backward_prepared": (C.c_int, [C.POINTER(GabRig), _P] + [_P] * 4 + [_P, _P, _P, _P, _P] + [_P] * 5 + [_P] * 6 + [_P] +
                                   [C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_int32), _P]),
    "gab_face_frames_forward": (C.c_int, [C.c_int32, C.c_int32, _P, _P, C.c_int32, _P, _P, _P, _P, _P, _P]),
    "gab_face_frames_backward": (C.c_int, [C.c_int32, C.c_int32, _P, _P, C.c_int32, _P, _P, _P, _P, _P, C.c_int32, _P]),
    "gab_bind_forward": (C.c_int, [C.c_int32, C.c_int32, _P, _P, _P, _P, C.c_int32, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "gab_bind_backward": (C.c_int, [C.c_int32, C.c_int32, _P, _P, _P, _P, C.c_int32, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "gab_bind_backward_csr": (C.c_int, [C.c_int32, C.c_int32] + [_P] * 22),
    "gab_bind_backward_faces": (C.c_int, [C.c_int32, _P, _P, _P, _P]),
    "gab_zero_buffers": (C.c_int, [C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_int32), _P]),
    "gab_feed_row": (C.c_int, [_P, C.c_int32, C.c_int32, _P, C.c_int32, _P, _P, _P]),
    "gab_profile_enable": (C.c_int, [C.c_int]),
    "gab_profile_collect": (C.c_int, []),
    "gab_profile_entry": (C.c_int, [C.c_int32, C.POINTER(C.c_char_p), C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "gab_profile_reset": (C.c_int, []),
}
GAB_ABI_VERSION = 5

_gab = None


def gab():
    """The binding library; raises (never falls back) when it is not built."""
    global _gab
    if _gab is None:
        if not os.path.exists(GAB_LIB_PATH):
            raise RuntimeError(f"{GAB_LIB_PATH} is missing: run __graft_entry__.build() (hipcc, gfx950).  There is no CPU fallback.")
        _torch_first()
        lib = C.CDLL(GAB_LIB_PATH)
        for name, (res, args) in GAB_SYMBOLS.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        if lib.gab_abi_version() != GAB_ABI_VERSION:
            raise RuntimeError(f"gab ABI version {lib.gab_abi_version()} != {GAB_ABI_VERSION}")
        _gab = lib
    return _gab


# ------------------------------------------------------------------------------------------------
# libgls_hip.so : fused L1 + SSIM loss and densification statistics (include/gls.h)
# ------------------------------------------------------------------------------------------------
GLS_LIB_PATH = os.path.join(_HERE, "libgls_hip.so")
GLS_SYMBOLS = {
    "gls_abi_version": (C.c_int, []),
    "gls_last_error": (C.c_char_p, []),
    "gls_partial_floats": (C.c_int64, [C.c_int32] * 4),
    "gls_l1_ssim_forward": (C.c_int, [C.c_int32] * 4 + [_P, _P, C.c_float] + [_P] * 4),
    "gls_l1_ssim_backward": (C.c_int, [C.c_int32] * 4 + [_P] * 4 + [C.c_float, _P, _P]),
    "gls_l1_ssim_backward_split": (C.c_int, [C.c_int32] * 4 + [_P] * 5 + [C.c_int32, C.c_float, _P, _P]),
    "gls_l1_forward": (C.c_int, [C.c_int64, _P, _P, C.c_float, _P, _P, _P]),
    "gls_l1_forward_grad": (C.c_int, [C.c_int64, _P, _P, C.c_float, _P, _P, _P, _P]),
    "gls_l1_backward": (C.c_int, [C.c_int64, _P, _P, _P, C.c_float, _P, _P]),
    "gls_densification_stats": (C.c_int, [C.c_int32] + [_P] * 6),
    "gls_add_densification_stats": (C.c_int, [C.c_int32, _P, _P, C.c_int32, _P, _P, _P]),
    "gls_profile_enable": (C.c_int, [C.c_int]),
    "gls_profile_collect": (C.c_int, []),
    "gls_profile_entry": (C.c_int, [C.c_int32, C.POINTER(C.c_char_p), C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "gls_profile_reset": (C.c_int, []),
}
GLS_ABI_VERSION = 4

_gls = None


def gls():
    """The loss / statistics library; raises (never falls back) when it is not built."""
    global _gls
    if _gls is None:
        if not os.path.exists(GLS_LIB_PATH):
            raise RuntimeError(f"{GLS_LIB_PATH} is missing: run __graft_entry__.build() (hipcc, gfx950).  There is no CPU fallback.")
        _torch_first()
        lib = C.CDLL(GLS_LIB_PATH)
        for name, (res, args) in GLS_SYMBOLS.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        if lib.gls_abi_version() != GLS_ABI_VERSION:
            raise RuntimeError(f"gls ABI version {lib.gls_abi_version()} != {GLS_ABI_VERSION}")
        _gls = lib
    return _gls


def launch_profile_enable(on: bool) -> None:
    """Event pairs around every launch of libgab / libgls (include/gab.h, gls.h: *_profile_*); libgsr's twin is gsr_profile_enable."""
    for lib, tag in ((gab(), "gab"), (gls(), "gls")):
        getattr(lib, tag + "_profile_enable")(1 if on else 0)
        if on:
            getattr(lib, tag + "_profile_reset")()


def launch_profile_read() -> dict:
    """{kernel name: (total_ms, launches)} of libgab's and libgls's launches since launch_profile_enable(True)."""
    out = {}
    for lib, tag in ((gab(), "gab"), (gls(), "gls")):
        n = getattr(lib, tag + "_profile_collect")()
        for i in range(n):
            name, ms, k = C.c_char_p(), C.c_double(), C.c_int64()
            if getattr(lib, tag + "_profile_entry")(i, C.byref(name), C.byref(ms), C.byref(k)) == 0:
                out[name.value.decode()] = (ms.value, k.value)
    return out


def gls_error() -> str:
    return gls().gls_last_error().decode("utf-8", "replace")


def gab_error() -> str:
    return gab().gab_last_error().decode("utf-8", "replace")


def gsr_error() -> str:
    return gsr().gsr_last_error().decode("utf-8", "replace")
