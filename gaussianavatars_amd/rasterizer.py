"""Drop-in for the Python surface of the reference's `diff_gaussian_rasterization` package.

The reference imports `GaussianRasterizationSettings, GaussianRasterizer` at
gaussian_renderer/__init__.py:15, builds the 12-field settings at :37-50 and calls the rasterizer
with keyword arguments at :86-94, expecting `(color (3,H,W) f32, radii (P,) int32)` and gradients
for means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp.
(The package itself is an un-vendored submodule -- .gitmodules:4-6 -- so names, argument meaning
and error behaviour below restate its public contract)

Everything numeric happens in libgsr_hip.so through the C ABI of include/gsr.h; torch is used for
device memory, the current stream and autograd plumbing only.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import NamedTuple, Optional

import torch
from torch import nn

from . import _host, _lib

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians", "rasterize_bound", "rasterize_leaves", "last_forward_info", "set_tile_culling", "deferred_count",
           "get_tile_culling", "set_exact_scale_grad", "set_deterministic", "set_fast_blend", "set_poison_state"]


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


# ---- exact tile culling (include/gsr.h: GsrSettings.tile_culling) --------------------------------
# On by default: instances that cannot reach a tile are not binned (same image / radii / gradients, about a third
# fewer keys to sort).  GSR_TILE_CULLING=0 or set_tile_culling(False) keeps the reference's rect-based lists, which is
# what the parity tests compare with the oracle index for index.
_tile_culling = int(os.environ.get("GSR_TILE_CULLING", "1"))


def set_tile_culling(enabled: bool) -> bool:
    """Process-wide switch; returns the previous value."""
    global _tile_culling
    prev, _tile_culling = _tile_culling, int(enabled)   # 0 off, 1 on (production), 2 on + sorted lists kept (debug)
    return prev


def get_tile_culling() -> int:
    return _tile_culling


# ---- dL/dscales convention (include/gsr.h: GsrSettings.exact_scale_grad) -------------------------
# Default = upstream: the gradient w.r.t. (scale_modifier * scale), no factor for the modifier itself.
_exact_scale_grad = int(os.environ.get("GSR_EXACT_SCALE_GRAD", "0"))


# ---- bit-reproducible backward (include/gsr.h: GsrSettings.deterministic) -------------------------
_deterministic = int(os.environ.get("GSR_DETERMINISTIC", "0"))


def set_deterministic(enabled: bool) -> bool:
    """Process-wide switch; True makes the rasterizer's backward bit-reproducible run to run (fixed-point accumulation instead of fp32
    atomics; ~the same speed).  Returns the previous value."""
    global _deterministic
    prev, _deterministic = bool(_deterministic), int(bool(enabled))
    return prev


# ---- fast blend (include/gsr.h: GsrSettings.fast_blend) --------------------------------------------
# On by default: the two blend kernels work in the 2^x domain (pre-scaled conic, hardware exp2, fused products) -- a third fewer
# instructions per record.  Integer outputs are unchanged; the image agrees with the exact mode to ~1e-6 except at the (counted, rare)
# pixels where a record sits within an ulp of one of the two blend thresholds.  GSR_FAST_BLEND=0 or set_fast_blend(False) selects
# the kernels whose image is bit-identical to the oracle's (what the bit-exact parity tests run).
_fast_blend = int(os.environ.get("GSR_FAST_BLEND", "1"))


def set_fast_blend(enabled) -> int:
    """Process-wide switch; returns the previous value.  True / False (1 / 0) select the fast / exact kernels; the raw integer 2 (the A/B
    mode of include/gsr.h: fast forward with the pixel-parallel fast backward) is kept as it is, so a save / restore pair
    `prev = set_fast_blend(x); ...; set_fast_blend(prev)` restores every mode."""
    global _fast_blend
    prev, _fast_blend = _fast_blend, int(enabled)
    return prev


# ---- poisoned state buffers (debugging aid, GSR_POISON_STATE=1) --------------------------------------
# The three state buffers of a forward are fresh torch.empty allocations: whatever a kernel reads from them, an earlier kernel of the
# same frame has to have written.  With this switch on they are filled with 0xFF bytes first (NaN as floats, 2^32 - 1 as counters), so
# a word that is read before it is written shows up in the tests instead of depending on what the caching allocator recycled.
_poison_state = int(os.environ.get("GSR_POISON_STATE", "0"))

# Whether a backward can follow a forward is decided by the caller's GRAD MODE as well as by its inputs: inside Function.forward autograd is
# always off and ctx.needs_input_grad only repeats the inputs' requires_grad flags -- render.py / fps_benchmark_*.py render nn.Parameters under
# torch.no_grad(), and render() hands every call a screen-space leaf that requires a gradient.  The entry points below note the mode around
# .apply (thread-local); a direct .apply from elsewhere finds None and is treated as "a backward may follow".
import threading as _threading

_call_state = _threading.local()


def _apply_noting_grad_mode(fn, *args):
    prev = getattr(_call_state, "grad", None)
    _call_state.grad = torch.is_grad_enabled()
    try:
        return fn.apply(*args)
    finally:
        _call_state.grad = prev


def _backward_may_follow(needs) -> bool:
    grad = getattr(_call_state, "grad", None)
    return bool(needs) and (grad is None or grad)



def set_poison_state(enabled: bool) -> bool:
    """Process-wide debugging switch; returns the previous value."""
    global _poison_state
    prev, _poison_state = bool(_poison_state), int(bool(enabled))
    return prev


def set_exact_scale_grad(enabled: bool) -> bool:
    """Process-wide switch; True multiplies dL/dscales by scale_modifier (the exact chain rule).  Returns the previous value."""
    global _exact_scale_grad
    prev, _exact_scale_grad = bool(_exact_scale_grad), int(bool(enabled))
    return prev


# ---- per-device running estimate of the binning capacity (instances per frame) -----------------
_CAP_QUANTUM = 1 << 16
_capacity_hint: dict = {}
_last_info: dict = {}


_last_binning: list = [None]
_forward_peak: list = [0]   # largest instance count of any forward since the caller last zeroed it (GraphedStep's warm-up sizes its capacity from it)


def last_forward_info() -> dict:
    """{'num_rendered', 'capacity', 'replays', 'tile_culling', 'rect_instances'} of the most recent forward on this
    process.  num_rendered counts the instances actually binned; rect_instances is the reference's rect-based count
    (sum of tiles_touched; equal to num_rendered without tile culling) -- read back from the device on request."""
    info = dict(_last_info)
    b = _last_binning[0]
    if isinstance(b, tuple):   # (state allocation, byte offset of the binning buffer inside it): the compiled host's frame, sliced on request
        b = b[0][b[1]:]
    if b is not None:
        info["rect_instances"] = int(b[8:16].view(torch.int64).item())
        if info.get("production_binning"):   # header of the production path (csrc/gsr_device.h: BinHeader)
            info["binned_splats"] = int(b[16:20].view(torch.int32).item())
            info["tile_instances"] = int(b[32:40].view(torch.int64).item())   # after snug-rect culling: what the sort path would bin
    return info


def release_last_state() -> None:
    """Drops the reference last_forward_info() keeps to the most recent frame's state buffers (they are otherwise freed when the next frame is rendered)."""
    _last_binning[0] = None


def _round_cap(n: int) -> int:
    return max(_CAP_QUANTUM, (int(n) + _CAP_QUANTUM - 1) // _CAP_QUANTUM * _CAP_QUANTUM)


# ---- deferred instance count (include/gsr.h: GsrSettings.deferred_count) -------------------------
# A forward normally waits for the frame's instance count (to grow the binning buffer and replay the frame when it does not
# fit).  Inside a `deferred_count` context it does not: the binning buffer has the context's fixed capacity, the count goes
# to a persistent slot the caller reads after the fact, and the call contains no host synchronisation at all -- which is what
# lets a whole frame step be stream-captured and replayed as a hipGraph (gaussianavatars_amd.graphs.GraphedStep).
class _Deferred:
    def __init__(self, capacity: int):
        self.capacity = _round_cap(capacity)
        self.slots: list = []      # one persistent slot per rasterizer forward issued inside the context

    def take(self) -> int:
        if not _free_slots:
            raise RuntimeError(f"all {_lib.GSR_COUNT_SLOTS} deferred-count slots are in use (release GraphedStep objects that are no longer needed)")
        slot = _free_slots.pop()
        self.slots.append(slot)
        return slot

    def release(self):
        """Returns the slots to the pool, their sticky overflow marks cleared (call with none of their frames in flight)."""
        lib = _lib.gsr()
        for slot in self.slots:
            lib.gsr_count_slot_overflow(slot, None, 1)
        _free_slots.extend(self.slots)
        self.slots = []

    def overflow(self, reset: bool = False) -> int:
        """The STICKY overflow mark of the slots: the instance count of the most recent frame, since the last reset, that did not fit the
        capacity (0: every frame fitted).  counts() only shows the newest frame of each slot; a replayed recording overwrites it."""
        lib = _lib.gsr()
        worst = 0
        for slot in self.slots:
            n = C.c_int64(0)
            if lib.gsr_count_slot_overflow(slot, C.byref(n), int(reset)) != _lib.GSR_OK:
                raise RuntimeError(_lib.gsr_error())
            worst = max(worst, int(n.value))
        return worst

    def counts(self) -> list:
        """The newest instance count posted to each slot (-1: nothing yet)."""
        lib = _lib.gsr()
        out = []
        for slot in self.slots:
            n = C.c_int64(0)
            if lib.gsr_count_slot_read(slot, C.byref(n), None) != _lib.GSR_OK:
                raise RuntimeError(_lib.gsr_error())
            out.append(int(n.value))
        return out


_free_slots = list(range(_lib.GSR_COUNT_SLOTS - 1, -1, -1))

# ---- the count wait of an eager frame, moved behind the rest of the call's host work (round 4) ---------------------------------------------
# gsr_forward launches the frame's eight kernels and then spins until the fifth has posted the instance count -- ~18 us per step during which
# the Python side of a step that is bound by its host work does nothing.  The entry points of the leaves entries
# (rasterize_bound / rasterize_leaves) take the deferred form of the call instead -- the count goes to a persistent slot of this (device,
# stream) -- let Function.forward and autograd's wrapping of its outputs finish, and wait THEN (gsr_count_slot_wait), still before the image
# is handed to code that cannot be replayed.  A frame that did not fit its buffer rendered nothing: the call is made again with a larger one,
# exactly as the loop inside forward() does for the blocking form.  GSR_LATE_COUNT=0 keeps the blocking form.
_late_count = os.environ.get("GSR_LATE_COUNT", "1") != "0"
_late_slots: dict = {}   # (device index, raw stream) -> persistent slot


_LATE_SLOTS_MAX = 8   # persistent slots the late waits may hold at once (of GSR_COUNT_SLOTS = 128: the rest stays with the recordings)


def _late_slot(dev, stream) -> int:
    """The persistent count slot of this (device, stream)'s late waits.  At most _LATE_SLOTS_MAX are held: code that keeps creating streams
    (lanes re-created, per-iteration streams) recycles the least recently used one -- its frames have been waited for, a late wait never leaves a
    frame in flight -- instead of draining the pool the recordings draw from; a recycled stream address simply gets a freshly reset slot."""
    k = (dev.index, stream)
    slot = _late_slots.pop(k, None)
    if slot is None:
        if len(_late_slots) >= _LATE_SLOTS_MAX:
            slot = _late_slots.pop(next(iter(_late_slots)))      # least recently used (dicts keep insertion order; a hit re-inserts below)
        elif _free_slots:
            slot = _free_slots.pop()
        elif _late_slots:
            slot = _late_slots.pop(next(iter(_late_slots)))
        else:
            return -1   # (every slot handed to recordings: this call takes the blocking form)
        _lib.gsr().gsr_count_slot_overflow(slot, None, 1)
    _late_slots[k] = slot
    return slot


def _finish_late(pend) -> bool:
    """Waits for the count of the frame `pend` describes; True when the frame fitted its binning buffer (then the bookkeeping the blocking form
    does inside forward() happens here), False when it has to be rendered again (the capacity hint is raised first)."""
    slot, seq, cap, key, stream, dev = pend
    lib = _lib.gsr()
    n = C.c_int64(0)
    with _lib.on_device(dev):
        rc = lib.gsr_count_slot_wait(slot, seq, stream, C.byref(n))
    if rc != _lib.GSR_OK:
        raise RuntimeError(f"gsr_count_slot_wait failed ({rc}): {_lib.gsr_error()}")
    I = int(n.value)
    if I > cap:
        lib.gsr_count_slot_overflow(slot, None, 1)   # (the device left its sticky mark: this slot's only reader is this function)
        _capacity_hint[key] = _round_cap(int(I * 1.25) + 1)
        return False
    _capacity_hint[key] = max(_round_cap(int(I * 1.25) + 1), min(cap, _round_cap(2 * I + 1)))
    _forward_peak[0] = max(_forward_peak[0], I)
    _last_info["num_rendered"] = I
    return True


def _is_plain(t) -> bool:
    return t.is_cuda and t.dtype is torch.float32 and t.is_contiguous()


_camera_copies: dict = {}   # id(tensor) -> (the tensor, its version, a contiguous fp32 copy)


def _camera_tensor(t):
    """A contiguous fp32 device tensor with `t`'s values, or None (a host tensor: not this entry's case).  fps_benchmark_demo.py:30-31 and scene/cameras.py:44-46
    hand the rasterizer TRANSPOSED VIEWS (`.T`, `.transpose(0, 1)`) that live as long as the camera: copying them in every frame is a launch per matrix, so the copy
    is made once per (object, version) and kept beside the camera's tensor."""
    if t.is_cuda and t.dtype is torch.float32 and t.is_contiguous():
        return t
    if not t.is_cuda:
        return None
    hit = _camera_copies.get(id(t))
    if hit is not None and hit[0] is t and hit[1] == t._version:
        return hit[2]
    if len(_camera_copies) > 256:
        _camera_copies.clear()
    c = t.detach().to(torch.float32).contiguous()
    _camera_copies[id(t)] = (t, t._version, c)
    return c


def _native_leaves_entry(H, xyz, means2D, sh_dc, sh_rest, opacity_logit, log_scaling, rotation, face_R, face_scale, face_center, face_quat, binding, csr, rs):
    """The same frame through the COMPILED host (csrc/gaa_host.cpp: rasterize_bound): layouts, ONE state allocation (geom | img | binning),
    gsr_forward_bound in its deferred form, the autograd node, and the late count wait, in one native call; the backward runs on autograd's
    device thread without the interpreter.  The capacity bookkeeping stays here (shared with the Python twin: one running estimate per path).
    Returns None when the call is outside what the native entry takes (the caller then uses _RasterizeBound)."""
    if not (_is_plain(xyz) and _is_plain(sh_dc) and _is_plain(sh_rest) and _is_plain(opacity_logit) and _is_plain(log_scaling) and _is_plain(rotation)):
        return None
    P = xyz.shape[0]
    if P == 0 or sh_rest.dim() != 3 or sh_rest.shape[1] < 1 or rs.debug:
        return None
    bg, vm, pm, cp = _camera_tensor(rs.bg), _camera_tensor(rs.viewmatrix), _camera_tensor(rs.projmatrix), _camera_tensor(rs.campos)
    if bg is None or vm is None or pm is None or cp is None:
        return None
    if binding is not None:
        if not (_is_plain(face_R) and _is_plain(face_scale) and _is_plain(face_center) and _is_plain(face_quat)) or csr is None:
            return None
        face_begin, slot = csr[1], csr[3]
    else:
        face_begin = slot = None
    dev = xyz.device
    Hh, Ww = int(rs.image_height), int(rs.image_width)
    lib = _lib.gsr()
    tc = int(_tile_culling)
    prod = _binning_layout(lib, 0, Ww, Hh, P, tc).path == 1
    key = (dev.index, Hh, Ww, prod)
    late = _late_slot(dev, _lib.raw_stream(dev))
    replays = 0
    while True:
        cap = _capacity_hint.get(key) or _round_cap((24 if prod else 8) * P)
        r = H.rasterize_bound(xyz, means2D, sh_dc, sh_rest, opacity_logit, log_scaling, rotation, face_R, face_scale, face_center, face_quat, binding, slot, face_begin,
                              bg, vm, pm, cp, Hh, Ww, float(rs.tanfovx), float(rs.tanfovy), float(rs.scale_modifier), int(rs.sh_degree), 0, tc,
                              int(_exact_scale_grad), int(_deterministic), int(_fast_blend), cap, late)
        I = r.num_rendered
        if r.fitted:
            break
        _capacity_hint[key] = _round_cap(int(I * 1.25) + 1)   # the frame did not fit: its kernels did nothing, the node just made is dropped un-walked
        replays += 1
    _capacity_hint[key] = max(_round_cap(int(I * 1.25) + 1), min(cap, _round_cap(2 * I + 1)))
    if I > _forward_peak[0]:
        _forward_peak[0] = I
    _last_info.update(num_rendered=I, capacity=cap, replays=replays, tile_culling=bool(tc), production_binning=prod, forward_only=bool(r.forward_only),
                      binning_path=r.path, rank_bands=r.nbands, bound=True, native_host=True)
    # (state allocation, byte offset of its binning part): last_forward_info() reads the header on request.  This keeps the last frame's ONE state
    # allocation alive until the next frame replaces it (a view would too, a copy of the header would cost a launch per frame): release_last_state() drops it.
    _last_binning[0] = (r.state, r.off_binning)
    return r.color, r.radii, r.visible


def _apply_leaves_entry(*args):
    if _late_count and _deferred is None and not _poison_state:
        H = _host.get()
        if H is not None:
            out = _native_leaves_entry(H, *args)
            if out is not None:
                return out
    _last_info["native_host"] = False
    return _apply_late(_RasterizeBound, *args)


def _apply_late(fn, *args):
    """fn.apply with the caller's grad mode noted and, unless a recording or GSR_LATE_COUNT=0 says otherwise, the count awaited after the
    call's own host work instead of inside it."""
    if not _late_count or _deferred is not None:
        return _apply_noting_grad_mode(fn, *args)
    replays = 0
    while True:
        _call_state.late, _call_state.pending = True, None
        try:
            out = _apply_noting_grad_mode(fn, *args)
        finally:
            _call_state.late = False
        pend, _call_state.pending = _call_state.pending, None
        if pend is None:
            return out
        if _finish_late(pend):
            if replays:
                _last_info["replays"] = replays
            return out
        replays += 1   # the frame did not fit: its kernels did nothing, the node just made is dropped un-walked
_deferred: Optional[_Deferred] = None


class deferred_count:
    """with deferred_count(capacity) as d: ... -- every rasterizer forward inside enqueues without waiting for its instance count;
    d.counts() afterwards (once the work has run) returns them; a count above d.capacity means that frame skipped its binning and
    blend kernels: its image and gradients are not valid.  d.release() returns the slots."""

    def __init__(self, capacity_or_state):
        self.state = capacity_or_state if isinstance(capacity_or_state, _Deferred) else _Deferred(int(capacity_or_state))

    def __enter__(self) -> _Deferred:
        global _deferred
        self.prev, _deferred = _deferred, self.state
        return self.state

    def __exit__(self, *exc):
        global _deferred
        _deferred = self.prev
        return False


def _debug_dump(path: str, args) -> None:
    """Upstream's `debug=True` behaviour (diff_gaussian_rasterization/__init__.py: on an exception of the native call the arguments are
    written with torch.save to snapshot_fw.dump / snapshot_bw.dump in the working directory, a note is printed, the error is re-raised)."""
    try:
        cpu_args = tuple(a.detach().cpu().clone() if isinstance(a, torch.Tensor) else a for a in args)
        torch.save(cpu_args, path)
        print(f"\nAn error occured in {'forward' if 'fw' in path else 'backward'}. Please forward {path} for debugging.")
    except Exception as e:   # noqa: BLE001 -- the dump must never replace the error it documents
        print(f"\n(debug snapshot {path} could not be written: {e})")


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None or t.numel() == 0 else t.data_ptr()   # a plain int converts to the c_void_p argument


def _f32c(t: torch.Tensor, name: str) -> torch.Tensor:
    if t.is_cuda and t.dtype is torch.float32 and t.is_contiguous():
        return t
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a device tensor (got {t.device}); the MI355X rasterizer has no CPU path")
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _make_settings(rs: GaussianRasterizationSettings, keep: list) -> _lib.GsrSettings:
    s = _lib.GsrSettings()
    s.image_height, s.image_width = int(rs.image_height), int(rs.image_width)
    s.tanfovx, s.tanfovy = float(rs.tanfovx), float(rs.tanfovy)
    s.scale_modifier = float(rs.scale_modifier)
    s.sh_degree = int(rs.sh_degree)
    s.prefiltered, s.debug = int(bool(rs.prefiltered)), int(bool(rs.debug))
    s.tile_culling = int(_tile_culling)
    s.exact_scale_grad = int(_exact_scale_grad)
    s.deterministic = int(_deterministic)
    s.fast_blend = int(_fast_blend)
    for field in ("bg", "viewmatrix", "projmatrix", "campos"):
        t = _f32c(getattr(rs, field), field)
        keep.append(t)
        setattr(s, field, t.data_ptr())
    return s


_layout_cache: dict = {}


def _layouts(lib, P, W, H):
    """(geom, image) layouts: pure host arithmetic of the C ABI, cached per shape."""
    key = (P, W, H)
    hit = _layout_cache.get(key)
    if hit is None:
        gl, il = _lib.GsrGeomLayout(), _lib.GsrImageLayout()
        lib.gsr_geom_layout(P, C.byref(gl))
        lib.gsr_image_layout(W, H, C.byref(il))
        if len(_layout_cache) > 64:
            _layout_cache.clear()
        hit = _layout_cache[key] = (gl, il)
    return hit


def _binning_layout(lib, cap, W, H, P, mode):
    key = ("b", cap, W, H, P, mode)
    hit = _layout_cache.get(key)
    if hit is None:
        hit = _lib.GsrBinningLayout()
        lib.gsr_binning_layout(cap, W, H, P, mode, C.byref(hit))
        if len(_layout_cache) > 64:
            _layout_cache.clear()
        _layout_cache[key] = hit
    return hit


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings, sh_rest=None):
        ctx.set_materialize_grads(False)   # no zeros tensor for the (integer, non-differentiable) radii output
        lib = _lib.gsr()
        dev = means3D.device
        keep: list = []
        s = _make_settings(raster_settings, keep)
        # no differentiable input (torch.no_grad, inference): no backward can follow, the forward skips preparing for one
        s.forward_only = int(not _backward_may_follow(any(ctx.needs_input_grad[:8]) or (len(ctx.needs_input_grad) > 9 and ctx.needs_input_grad[9])))
        means3D = _f32c(means3D, "means3D")
        if means3D.dim() != 2 or means3D.shape[1] != 3:
            raise RuntimeError("means3D must have dimensions (num_points, 3)")
        P = means3D.shape[0]
        H, W = s.image_height, s.image_width
        sh = _f32c(sh, "shs") if sh.numel() else sh
        colors_precomp = _f32c(colors_precomp, "colors_precomp") if colors_precomp.numel() else colors_precomp
        opacities = _f32c(opacities, "opacities")
        scales = _f32c(scales, "scales") if scales.numel() else scales
        rotations = _f32c(rotations, "rotations") if rotations.numel() else rotations
        cov3Ds_precomp = _f32c(cov3Ds_precomp, "cov3D_precomp") if cov3Ds_precomp.numel() else cov3Ds_precomp
        split = sh_rest is not None and sh_rest.numel() > 0
        if split:   # the model's two leaf tensors, read in place (include/gsr.h: gsr_forward_ex)
            sh_rest = _f32c(sh_rest, "shs_rest")
            if sh.dim() != 3 or sh.shape[1] != 1 or sh_rest.shape[0] != sh.shape[0]:
                raise RuntimeError("split SH: shs must be (P,1,3) and shs_rest (P,M-1,3)")
        else:
            sh_rest = torch.empty(0, device=dev)
        M = (int(sh.shape[1]) + (int(sh_rest.shape[1]) if split else 0)) if sh.numel() else 0

        gl, il = _layouts(lib, P, W, H)
        u8 = dict(dtype=torch.uint8, device=dev)
        color = torch.empty((3, H, W), dtype=torch.float32, device=dev)
        radii = torch.empty((P,), dtype=torch.int32, device=dev)
        geom = torch.empty(gl.total, **u8)
        img = torch.empty(il.total, **u8)
        if _poison_state:
            geom.fill_(0xFF), img.fill_(0xFF)

        # the capacity counts what the binning path produces: tile instances on the per-tile sort path, quadrant-stream entries
        # on the production path (include/gsr.h: gsr_binning_layout) -- one running estimate per path
        prod = _binning_layout(lib, 0, W, H, P, int(s.tile_culling)).path == 1
        key = (dev.index, H, W, prod)
        cap = _capacity_hint.get(key) or _round_cap((24 if prod else 8) * P)
        defer = _deferred
        stream = _lib.raw_stream(dev)
        late_slot = -1
        if defer is not None:   # fixed capacity, count posted to a persistent slot, nothing waits (see deferred_count)
            cap, s.deferred_count = defer.capacity, defer.take() + 1
        elif getattr(_call_state, "late", False):   # (_apply_late) the count is awaited after this call's host work, not inside it
            late_slot = _late_slot(dev, stream)
            if late_slot >= 0:
                s.deferred_count = late_slot + 1
        n_host = C.c_int64(0)
        replays = 0
        with _lib.on_device(dev):
            while True:
                bl = _binning_layout(lib, cap, W, H, P, int(s.tile_culling))
                binning = torch.empty(bl.total, **u8)
                if _poison_state:
                    binning.fill_(0xFF)
                rc = lib.gsr_forward_ex(C.byref(s), P, M, _ptr(means3D), _ptr(sh), _ptr(sh_rest), _ptr(colors_precomp), _ptr(opacities),
                                     _ptr(scales), _ptr(rotations), _ptr(cov3Ds_precomp), _ptr(color), _ptr(radii),
                                     _ptr(geom), _ptr(binning), cap, _ptr(img), C.byref(n_host), stream)
                if rc == _lib.GSR_E_CAPACITY:
                    cap = _round_cap(int(n_host.value * 1.25) + 1)
                    replays += 1
                    continue
                if rc != _lib.GSR_OK:
                    msg = _lib.gsr_error()
                    if "provide" in msg:  # the two argument-contract errors are plain Exceptions upstream
                        raise Exception(msg)
                    if raster_settings.debug:   # upstream's debug mode: the failing call's arguments go to a file before the error is raised
                        _debug_dump("snapshot_fw.dump", (raster_settings.bg, means3D, colors_precomp, opacities, scales, rotations, raster_settings.scale_modifier,
                                                         cov3Ds_precomp, raster_settings.viewmatrix, raster_settings.projmatrix, raster_settings.tanfovx,
                                                         raster_settings.tanfovy, raster_settings.image_height, raster_settings.image_width, sh,
                                                         raster_settings.sh_degree, raster_settings.campos, raster_settings.prefiltered))
                    raise RuntimeError(f"gsr_forward failed ({rc}): {msg}")
                break
        I = int(n_host.value)
        if late_slot >= 0:
            _call_state.pending = (late_slot, int(lib.gsr_last_forward_seq()), cap, key, stream, dev)
            I = cap      # as a recording's frame: the backward takes the capacity as the bound; _finish_late does the bookkeeping
        elif defer is not None:
            I = cap      # unknown until the kernels have run: the backward takes the capacity as the bound
        else:   # next frame: 25 % headroom over what this one needed, never shrinking below it
            _capacity_hint[key] = max(_round_cap(int(I * 1.25) + 1), min(cap, _round_cap(2 * I + 1)))
            _forward_peak[0] = max(_forward_peak[0], I)
        _last_info.update(num_rendered=I if (defer is None and late_slot < 0) else -1, capacity=cap, replays=replays, tile_culling=bool(s.tile_culling), production_binning=prod, forward_only=bool(s.forward_only),
                          binning_path=int(bl.path), rank_bands=int(bl.nbands), bound=False)   # 0 rank path, 1 depth-ordered scatter, 2 per-tile sort (include/gsr.h)
        _last_binning[0] = binning

        ctx.raster_settings = raster_settings
        ctx.tile_culling = int(s.tile_culling)    # the state buffers are laid out for this mode
        ctx.deterministic = int(s.deterministic)  # and the accumulators zero-filled for this one
        ctx.fast_blend = int(s.fast_blend)        # and the per-splat records written for this one
        ctx.num_rendered = I
        ctx.capacity = cap
        ctx.M = M
        ctx.split = split
        ctx.save_for_backward(colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geom, binning, img, sh_rest)
        # render()'s visibility_filter (radii > 0) as the forward wrote it: a bool view of the state buffer, no comparison launch
        visible = geom[gl.visible: gl.visible + P].view(torch.bool)
        ctx.mark_non_differentiable(radii, visible)
        return color, radii, visible

    @staticmethod
    def backward(ctx, grad_out_color, _grad_radii, _grad_visible=None):
        if grad_out_color is None:
            return (None,) * 10
        lib = _lib.gsr()
        colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geom, binning, img, sh_rest = ctx.saved_tensors
        rs = ctx.raster_settings
        dev = means3D.device
        keep: list = []
        s = _make_settings(rs, keep)
        s.tile_culling = ctx.tile_culling
        s.deterministic = ctx.deterministic
        s.fast_blend = ctx.fast_blend
        s.forward_only = 0
        P, M = means3D.shape[0], ctx.M
        f32 = dict(dtype=torch.float32, device=dev)
        grad_out_color = _f32c(grad_out_color, "grad_out_color")
        g_means3D = torch.empty((P, 3), **f32)
        g_means2D = torch.empty((P, 3), **f32)
        g_colors = torch.empty((P, 3), **f32)
        g_opacity = torch.empty((P, 1), **f32)
        g_cov3D = torch.empty((P, 6), **f32)
        use_sh, use_sr = sh.numel() > 0, scales.numel() > 0
        g_sh = torch.empty((P, 1 if ctx.split else M, 3), **f32) if use_sh else None
        g_sh_rest = torch.empty((P, M - 1, 3), **f32) if ctx.split else None
        g_scales = torch.empty((P, 3), **f32) if use_sr else None
        g_rot = torch.empty((P, 4), **f32) if use_sr else None
        stream = _lib.raw_stream(dev)
        with _lib.on_device(dev):
            rc = lib.gsr_backward_ex(C.byref(s), P, M, _ptr(means3D), _ptr(sh), _ptr(sh_rest), _ptr(colors_precomp), _ptr(scales),
                                  _ptr(rotations), _ptr(cov3Ds_precomp), _ptr(radii), _ptr(geom), _ptr(binning),
                                  ctx.capacity, _ptr(img), ctx.num_rendered, _ptr(grad_out_color),
                                  _ptr(g_means3D), _ptr(g_means2D), _ptr(g_sh), _ptr(g_sh_rest), _ptr(g_colors), _ptr(g_opacity),
                                  _ptr(g_scales), _ptr(g_rot), _ptr(g_cov3D), stream)
        if rc != _lib.GSR_OK:
            msg = _lib.gsr_error()
            if rs.debug:
                _debug_dump("snapshot_bw.dump", (rs.bg, means3D, radii, colors_precomp, scales, rotations, rs.scale_modifier, cov3Ds_precomp, rs.viewmatrix,
                                                 rs.projmatrix, rs.tanfovx, rs.tanfovy, grad_out_color, sh, rs.sh_degree, rs.campos, ctx.num_rendered))
            raise RuntimeError(f"gsr_backward failed ({rc}): {msg}")
        return (g_means3D, g_means2D, g_sh, g_colors if not use_sh else None, g_opacity, g_scales, g_rot,
                g_cov3D if not use_sr else None, None, g_sh_rest)


class _RasterizeBound(torch.autograd.Function):
    """The rasterizer on a mesh-BOUND model's own leaves (include/gsr.h: gsr_forward_bound / gsr_backward_bound): get_xyz / get_scaling /
    get_rotation / get_opacity (scene/gaussian_model.py:113-160) are evaluated inside the first kernel, the world-space tensors never
    exist, and one autograd node replaces the accessor node + the rasterizer node.  Inputs: the leaves (_xyz, _scaling, _rotation,
    _opacity, _features_dc, _features_rest), the screen-space dummy, the four per-face frame tensors, binding and its CSR."""

    @staticmethod
    def forward(ctx, xyz, means2D, sh_dc, sh_rest, opacity_logit, log_scaling, rotation, face_R, face_scale, face_center, face_quat,
                binding, csr, raster_settings):
        ctx.set_materialize_grads(False)
        lib = _lib.gsr()
        dev = xyz.device
        keep: list = []
        s = _make_settings(raster_settings, keep)
        need = ctx.needs_input_grad
        s.forward_only = int(not _backward_may_follow(any(need[:11])))
        xyz, sh_dc, sh_rest = _f32c(xyz, "_xyz"), _f32c(sh_dc, "_features_dc"), _f32c(sh_rest, "_features_rest")
        opacity_logit, log_scaling, rotation = _f32c(opacity_logit, "_opacity"), _f32c(log_scaling, "_scaling"), _f32c(rotation, "_rotation")
        P = xyz.shape[0]
        b = _lib.GsrBound()
        if binding is None:   # an unbound model's leaves: the activations only (exp, normalize, sigmoid)
            fR = fs = fc = fq = None
            F = 0
        else:
            fR, fs, fc, fq = (_f32c(t, n) for t, n in ((face_R, "face_orien_mat"), (face_scale, "face_scaling"), (face_center, "face_center"),
                                                        (face_quat, "face_orien_quat")))
            F = fc.shape[0]
            if binding.dtype not in (torch.int32, torch.int64) or not binding.is_contiguous() or binding.numel() != P:
                raise RuntimeError("binding must be a contiguous int32 / int64 tensor with one face per splat")
            if csr is None or csr[1].numel() != F + 1 or csr[3].numel() != P:
                raise RuntimeError("the bound rasterizer needs the binding's per-face CSR (binding.binding_csr) for this binding and mesh")
            b.binding, b.binding_is_i64, b.F = binding.data_ptr(), int(binding.dtype == torch.int64), F
            b.face_R, b.face_scale, b.face_center, b.face_quat = fR.data_ptr(), fs.data_ptr(), fc.data_ptr(), fq.data_ptr()
        H, W = s.image_height, s.image_width
        M = 1 + int(sh_rest.shape[1])
        gl, il = _layouts(lib, P, W, H)
        u8 = dict(dtype=torch.uint8, device=dev)
        color = torch.empty((3, H, W), dtype=torch.float32, device=dev)
        radii = torch.empty((P,), dtype=torch.int32, device=dev)
        geom = torch.empty(gl.total, **u8)
        img = torch.empty(il.total, **u8)
        if _poison_state:
            geom.fill_(0xFF), img.fill_(0xFF)
        prod = _binning_layout(lib, 0, W, H, P, int(s.tile_culling)).path == 1
        key = (dev.index, H, W, prod)
        cap = _capacity_hint.get(key) or _round_cap((24 if prod else 8) * P)
        defer = _deferred
        stream = _lib.raw_stream(dev)
        late_slot = -1
        if defer is not None:   # fixed capacity, count posted to a persistent slot, nothing waits (see deferred_count)
            cap, s.deferred_count = defer.capacity, defer.take() + 1
        elif getattr(_call_state, "late", False):   # (_apply_leaves_entry) the count is awaited after this call's host work, not inside it
            late_slot = _late_slot(dev, stream)
            if late_slot >= 0:
                s.deferred_count = late_slot + 1
        n_host = C.c_int64(0)
        replays = 0
        with _lib.on_device(dev):
            while True:
                bl = _binning_layout(lib, cap, W, H, P, int(s.tile_culling))
                binning = torch.empty(bl.total, **u8)
                if _poison_state:
                    binning.fill_(0xFF)
                rc = lib.gsr_forward_bound(C.byref(s), P, M, C.byref(b), _ptr(xyz), _ptr(sh_dc), _ptr(sh_rest), _ptr(opacity_logit),
                                           _ptr(log_scaling), _ptr(rotation), _ptr(color), _ptr(radii), _ptr(geom), _ptr(binning), cap,
                                           _ptr(img), C.byref(n_host), stream)
                if rc == _lib.GSR_E_CAPACITY:
                    cap = _round_cap(int(n_host.value * 1.25) + 1)
                    replays += 1
                    continue
                if rc != _lib.GSR_OK:
                    raise RuntimeError(f"gsr_forward_bound failed ({rc}): {_lib.gsr_error()}")
                break
        I = int(n_host.value)
        if late_slot >= 0:
            _call_state.pending = (late_slot, int(lib.gsr_last_forward_seq()), cap, key, stream, dev)
            I = cap      # as a recording's frame: the backward takes the capacity as the bound; _finish_late does the bookkeeping
        elif defer is not None:
            I = cap      # unknown until the kernels have run: the backward takes the capacity as the bound
        else:
            _capacity_hint[key] = max(_round_cap(int(I * 1.25) + 1), min(cap, _round_cap(2 * I + 1)))
            _forward_peak[0] = max(_forward_peak[0], I)
        _last_info.update(num_rendered=I if (defer is None and late_slot < 0) else -1, capacity=cap, replays=replays, tile_culling=bool(s.tile_culling), production_binning=prod, forward_only=bool(s.forward_only),
                          binning_path=int(bl.path), rank_bands=int(bl.nbands), bound=True)
        _last_binning[0] = binning
        ctx.raster_settings = raster_settings
        ctx.tile_culling, ctx.deterministic, ctx.fast_blend = int(s.tile_culling), int(s.deterministic), int(s.fast_blend)
        ctx.num_rendered, ctx.capacity, ctx.M, ctx.F = I, cap, M, F
        ctx.is64 = b.binding_is_i64
        ctx.csr = csr
        ctx.unbound = binding is None
        ctx.save_for_backward(xyz, sh_dc, sh_rest, opacity_logit, log_scaling, rotation, fR, fs, fc, fq, binding, radii, geom, binning, img)
        visible = geom[gl.visible: gl.visible + P].view(torch.bool)
        ctx.mark_non_differentiable(radii, visible)
        return color, radii, visible

    @staticmethod
    def backward(ctx, grad_out_color, _grad_radii, _grad_visible=None):
        if grad_out_color is None:
            return (None,) * 14
        lib = _lib.gsr()
        gab = None if ctx.unbound else _lib.gab()
        xyz, sh_dc, sh_rest, opacity_logit, log_scaling, rotation, fR, fs, fc, fq, binding, radii, geom, binning, img = ctx.saved_tensors
        dev = xyz.device
        keep: list = []
        s = _make_settings(ctx.raster_settings, keep)
        s.tile_culling, s.deterministic, s.fast_blend, s.forward_only = ctx.tile_culling, ctx.deterministic, ctx.fast_blend, 0
        P, M, F = xyz.shape[0], ctx.M, ctx.F
        f32 = dict(dtype=torch.float32, device=dev)
        if P == 0:   # everything pruned: no splat, no gradient -- zeros of the right shapes (the native entries have nothing to point at)
            z = lambda *shape: torch.zeros(shape, **f32)
            leaves = (z(0, 3), z(0, 3), z(0, 1, 3), z(0, M - 1, 3), z(0, 1), z(0, 3), z(0, 4))
            if ctx.unbound:
                return leaves + (None,) * 7
            return leaves + (z(F, 3, 3), z(F, 1), z(F, 3), z(F, 4), None, None, None)
        grad_out_color = _f32c(grad_out_color, "grad_out_color")
        g_xyz, g_means2D = torch.empty((P, 3), **f32), torch.empty((P, 3), **f32)
        g_dc, g_rest = torch.empty((P, 1, 3), **f32), torch.empty((P, M - 1, 3), **f32)
        g_op, g_ls, g_rot = torch.empty((P, 1), **f32), torch.empty((P, 3), **f32), torch.empty((P, 4), **f32)
        b = _lib.GsrBound()
        if ctx.unbound:
            scratch = torch.empty(9 * P, **f32)                                  # colour / covariance gradients (internal)
            d_face = face_begin = None
        else:
            rows_at = (9 * P + 3) // 4 * 4                                       # the CSR rows are read and written as float4: 16-byte aligned whatever P is
            scratch = torch.empty(rows_at + _lib.GAB_BIND_ROW_FLOATS * P, **f32)  # ... then the CSR rows
            d_face = torch.empty(17 * F, **f32)                                  # four contiguous blocks: center | orien_mat | scaling | orien_quat
            order, face_begin, _splat_face, slot = ctx.csr[:4]
            b.binding, b.binding_is_i64, b.F = binding.data_ptr(), ctx.is64, F
            b.face_R, b.face_scale, b.face_center, b.face_quat = fR.data_ptr(), fs.data_ptr(), fc.data_ptr(), fq.data_ptr()
            b.slot, b.rows = slot.data_ptr(), scratch.data_ptr() + 4 * rows_at
        stream = _lib.raw_stream(dev)
        with _lib.on_device(dev):
            rc = lib.gsr_backward_bound(C.byref(s), P, M, C.byref(b), _ptr(xyz), _ptr(sh_dc), _ptr(sh_rest), _ptr(opacity_logit), _ptr(log_scaling),
                                        _ptr(rotation), _ptr(radii), _ptr(geom), _ptr(binning), ctx.capacity, _ptr(img), ctx.num_rendered,
                                        _ptr(grad_out_color), _ptr(g_xyz), _ptr(g_means2D), _ptr(g_dc), _ptr(g_rest), _ptr(g_op), _ptr(g_ls),
                                        _ptr(g_rot), _ptr(scratch), stream)
            if rc != _lib.GSR_OK:
                raise RuntimeError(f"gsr_backward_bound failed ({rc}): {_lib.gsr_error()}")
            if not ctx.unbound:
                rc = gab.gab_bind_backward_faces(F, face_begin.data_ptr(), b.rows, d_face.data_ptr(), stream)
                if rc != 0:
                    raise RuntimeError(f"gab_bind_backward_faces failed ({rc}): {_lib.gab_error()}")
        if ctx.unbound:
            return (g_xyz, g_means2D, g_dc, g_rest, g_op, g_ls, g_rot, None, None, None, None, None, None, None)
        # (one as_strided per block: a slice + a view each were eight dispatches of ~1.5 us on a step whose host side is the critical path)
        return (g_xyz, g_means2D, g_dc, g_rest, g_op, g_ls, g_rot, d_face.as_strided((F, 3, 3), (9, 3, 1), 3 * F), d_face.as_strided((F, 1), (1, 1), 12 * F),
                d_face.as_strided((F, 3), (3, 1), 0), d_face.as_strided((F, 4), (4, 1), 13 * F), None, None, None)


def rasterize_leaves(xyz, means2D, sh_dc, sh_rest, opacity_logit, log_scaling, rotation, raster_settings):
    """-> (color, radii, visibility_filter) of an UNBOUND model straight from its leaves: get_scaling = exp, get_rotation = normalize,
    get_opacity = sigmoid (scene/gaussian_model.py:113-160) are evaluated inside the rasterizer's first kernel and their chain rule in
    its last one -- no activation launches, no activated tensors."""
    return _apply_leaves_entry(xyz, means2D, sh_dc, sh_rest, opacity_logit, log_scaling, rotation, None, None, None, None, None, None,
                                 raster_settings)


def rasterize_bound(xyz, means2D, sh_dc, sh_rest, opacity_logit, log_scaling, rotation, face_R, face_scale, face_center, face_quat, binding, csr,
                    raster_settings):
    """-> (color, radii, visibility_filter) of a mesh-bound model straight from its leaves and face frames (see _RasterizeBound)."""
    return _apply_leaves_entry(xyz, means2D, sh_dc, sh_rest, opacity_logit, log_scaling, rotation, face_R, face_scale, face_center, face_quat,
                                 binding, csr, raster_settings)


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings,
                        sh_rest=None):
    """-> (color, radii), the reference's pair."""
    return _apply_late(_RasterizeGaussians, means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                     cov3Ds_precomp, raster_settings, sh_rest)[:2]


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings: GaussianRasterizationSettings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions: torch.Tensor) -> torch.Tensor:
        """bool (P,): view-space depth > 0.2 (the only active test of upstream's frustum check)."""
        rs = self.raster_settings
        with torch.no_grad():
            pos = _f32c(positions, "positions")
            vm, pm = _f32c(rs.viewmatrix, "viewmatrix"), _f32c(rs.projmatrix, "projmatrix")
            out = torch.empty((pos.shape[0],), dtype=torch.uint8, device=pos.device)
            stream = _lib.raw_stream(pos.device)
            with _lib.on_device(pos.device):
                rc = _lib.gsr().gsr_mark_visible(pos.shape[0], _ptr(pos), _ptr(vm), _ptr(pm), _ptr(out), stream)
            if rc != _lib.GSR_OK:
                raise RuntimeError(f"gsr_mark_visible failed ({rc}): {_lib.gsr_error()}")
        return out.bool()

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None, shs_rest=None):
        """Same keywords as the reference.  Extension: `shs_rest` -- pass the model's `_features_dc` as `shs` and
        `_features_rest` here to skip the per-frame concatenation (get_features)."""
        rs = self.raster_settings
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception("Please provide excatly one of either SHs or precomputed colors!")
        if ((scales is None or rotations is None) and cov3D_precomp is None) or (
            (scales is not None or rotations is not None) and cov3D_precomp is not None
        ):
            raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        empty = torch.Tensor([])
        shs = empty if shs is None else shs
        colors_precomp = empty if colors_precomp is None else colors_precomp
        scales = empty if scales is None else scales
        rotations = empty if rotations is None else rotations
        cov3D_precomp = empty if cov3D_precomp is None else cov3D_precomp
        color, radii, visible = _apply_late(_RasterizeGaussians, means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
                                                          cov3D_precomp, rs, shs_rest)
        # extension: `radii > 0` of THIS call as the forward kernel wrote it (what render() returns as visibility_filter)
        self.visibility_filter = visible
        return color, radii
