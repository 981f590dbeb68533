"""Fused binding path: torch.autograd Functions over the C ABI of include/gab.h.

Each function replaces one composed-torch stage of the reference (see gab.h for file:line) with
one native call forward and one backward; torch supplies device memory, the current stream and the
autograd graph.  `gaussianavatars_amd.unfused` holds the same math as composed torch ops (the fp32
reference these kernels are tested against); nothing here falls back to it.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import _host, _lib


def _p(t):
    return None if t is None else t.data_ptr()   # a plain int converts to the c_void_p argument without an intermediate object


def _stream(dev):
    return _lib.raw_stream(dev)


def _chk(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed ({rc}): {_lib.gab_error()}")


def _need_cuda(t, name):
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a device tensor (got {t.device}); the fused binding path has no CPU implementation")


def _f32(t):
    if t.dtype is torch.float32 and t.is_contiguous():
        return t          # inside an autograd Function the inputs already are plain tensors: no detach / copy needed
    return t.detach().float().contiguous()


def _idx(t):
    t = t.detach()
    if t.dtype not in (torch.int32, torch.int64):
        t = t.long()
    return t.contiguous(), int(t.dtype == torch.int64)


class _Keep:
    """Tensors a node needs in its backward, kept for the node's whole life (not freed by the first backward).

    The accessors of a bound model hand the SAME node to every consumer of one mesh update (get_xyz / get_scaling /
    get_rotation / get_opacity share one launch), so the node must survive being walked more than once -- two
    render()+backward() passes at one timestep work in the reference, which recomputes the accessors per call.  What
    autograd's saved-tensor machinery would still give us is done by hand: every entry is a detached alias (never an
    output object, so no ctx -> output -> grad_fn -> ctx cycle keeps device memory until the cyclic GC runs) and the
    version counter of each is recorded and re-checked, so an in-place update between forward and backward raises
    like it does in stock autograd instead of yielding silently wrong gradients."""

    __slots__ = ("tensors", "versions")

    def __init__(self, *tensors):
        self.tensors = tuple(None if t is None else t.detach() for t in tensors)
        self.versions = tuple(None if t is None else t._version for t in tensors)

    def get(self):
        for t, v in zip(self.tensors, self.versions):
            if t is not None and t._version != v:
                raise RuntimeError("one of the variables needed for gradient computation has been modified by an inplace operation "
                                   f"(a {tuple(t.shape)} tensor saved by the fused binding path is at version {t._version}, expected {v})")
        return self.tensors


# -------------------------------------------------------------------------------------------------
# FLAME forward
# -------------------------------------------------------------------------------------------------
def _rig_struct(head) -> "_lib.GabRig":
    r = getattr(head, "_gab_rig", None)
    key = (head.v_template.data_ptr(), head.shapedirs.data_ptr(), head.posedirs.data_ptr(), head.J_regressor.data_ptr(),
           head.lbs_weights.data_ptr())
    if r is None or r[0] != key:
        for name in ("v_template", "shapedirs", "posedirs", "J_regressor", "lbs_weights"):
            b = getattr(head, name)
            _need_cuda(b, name)
            assert b.dtype == torch.float32 and b.is_contiguous(), name
        s = _lib.GabRig()
        s.V = head.v_template.shape[0]
        s.n_shape = int(head.n_shape_params)
        s.n_expr = int(head.shapedirs.shape[2]) - int(head.n_shape_params)
        s.v_template, s.shapedirs, s.posedirs = key[0], key[1], key[2]
        s.J_regressor, s.lbs_weights = key[3], key[4]
        par = [int(x) for x in head.parents.tolist()]
        if len(par) != 5:
            raise RuntimeError("the fused FLAME kernels are built for the 5-joint FLAME skeleton")
        s.parents[:] = par
        head._gab_rig = (key, s)
        r = head._gab_rig
    return r[1]


def _prepared_rig(head, rig, shape_flat, so_flat):
    """The per-(rig, shape, static_offset) part of the FLAME forward (include/gab.h: gab_flame_prepare), cached on the head and
    re-made when any of them changes (identity + in-place version of the two tensors; the rig's own buffers key _rig_struct)."""
    key = (id(rig), shape_flat.data_ptr(), shape_flat._version, None if so_flat is None else (so_flat.data_ptr(), so_flat._version))
    c = getattr(head, "_gab_prepared", None)
    if c is None or c[0] != key:
        lib = _lib.gab()
        dev = head.v_template.device
        buf = torch.empty(int(lib.gab_flame_prepared_floats(C.byref(rig))), dtype=torch.float32, device=dev)
        with _lib.on_device(dev):
            _chk(lib.gab_flame_prepare(C.byref(rig), _p(shape_flat), _p(so_flat), _p(buf), _stream(dev)), "gab_flame_prepare")
        head._gab_prepared = c = (key, buf, shape_flat, so_flat)   # the sources stay alive: their data_ptr is part of the key
    return c[1]


def _sequence_mode() -> str:
    return os.environ.get("GAA_MESH_SEQUENCE", "off")


def _sequence_table(head, rig, prepared, expr_tab):
    """Render mode over a timestep sequence (render.py:68-76, fps_benchmark_dataset.py:19-32: select_mesh_by_timestep under no_grad, frame
    after frame of one avatar): `v_shaped` of ALL T frames as one fp32-MFMA product (include/gab.h: gab_blend_sequence, the one GEMM-shaped
    product of the path), cached on the head and keyed on the prepared rig and on the expression table's identity and in-place version;
    a frame then takes its row (gab_flame_forward_sequence) instead of re-reading the 6 MB expression block.  Built on the SECOND
    gradient-free call that sees the same key (a training loop whose optimiser rewrites `expr` between two evaluation frames never pays
    for a table it would use once); tables beyond 1 GiB are not built.  OFF by default -- measured, it does not pay: the per-frame kernel
    already hides its GEMV behind the joint chain (cfg2: 6444 frames/s without the table, 6375 with it) --;
    GAA_MESH_SEQUENCE=auto enables the behaviour above, =eager builds on the first call.  Returns the (T, 3V) table or None."""
    mode = _sequence_mode()
    if mode == "off" or getattr(head, "flame_sequence", True) is False or rig.n_expr <= 0:
        return None
    T = int(expr_tab.shape[0])
    if T < 2 or T * 3 * rig.V * 4 > (1 << 30):
        return None
    key = (prepared.data_ptr(), id(prepared), expr_tab.data_ptr(), expr_tab._version, T)
    c = getattr(head, "_gab_sequence", None)
    if c is not None and c[0] == key:
        if c[1] is not None:
            return c[1]
    elif mode != "eager":
        head._gab_sequence = (key, None, None, None)   # seen once: the next gradient-free frame of this table builds it
        return None
    lib = _lib.gab()
    dev = expr_tab.device
    seq = torch.empty((T, 3 * rig.V), dtype=torch.float32, device=dev)
    with _lib.on_device(dev):
        _chk(lib.gab_blend_sequence(C.byref(rig), _p(prepared), _p(expr_tab), T, _p(seq), _stream(dev)), "gab_blend_sequence")
    head._gab_sequence = (key, seq, prepared, expr_tab)   # (the sources stay alive: their addresses are part of the key)
    return seq


def _use_prepared(head, shape_needs_grad, so_needs_grad) -> bool:
    """One launch forward / two backward (prepared rig) when neither shape nor static_offset is being optimised -- the
    reference's training setup (scene/flame_gaussian_model.py:155-178 puts only the per-timestep parameters in the optimiser).
    head.flame_impl = "classic" keeps the three-kernel forward and backward."""
    return not shape_needs_grad and not so_needs_grad and getattr(head, "flame_impl", "prepared") != "classic"


class _FlameForward(torch.autograd.Function):
    @staticmethod
    def forward(ctx, head, shape, expr, rotation, neck, jaw, eyes, translation, static_offset):
        ctx.set_materialize_grads(False)   # the backward takes NULL for an unused output's gradient
        lib = _lib.gab()
        rig = _rig_struct(head)
        dev = head.v_template.device
        V = rig.V
        ins = [_f32(t).reshape(-1) for t in (shape, expr, rotation, neck, jaw, eyes, translation)]
        if ins[0].numel() != rig.n_shape or ins[1].numel() != rig.n_expr or ins[5].numel() != 6:
            raise RuntimeError("fused FLAME forward is batch-1: shape (1,n_shape), expr (1,n_expr), eyes (1,6)")
        so = None if static_offset is None else _f32(static_offset).reshape(-1)
        if so is not None and so.numel() != 3 * V:
            raise RuntimeError("static_offset must be (1,V,3)")
        verts = torch.empty((1, V, 3), dtype=torch.float32, device=dev)
        v_shaped = torch.empty((1, V, 3), dtype=torch.float32, device=dev)
        ws = torch.empty(_lib.GAB_FLAME_WS_FLOATS, dtype=torch.float32, device=dev)
        need = ctx.needs_input_grad
        ctx.prepared = _prepared_rig(head, rig, ins[0], so) if _use_prepared(head, need[1], static_offset is not None and need[8]) else None
        with _lib.on_device(dev):
            if ctx.prepared is not None:
                _chk(lib.gab_flame_forward_prepared(C.byref(rig), _p(ctx.prepared), *[_p(t) for t in ins[1:]], _p(verts), _p(v_shaped),
                                                    _p(ws), _stream(dev)), "gab_flame_forward_prepared")
            else:
                _chk(lib.gab_flame_forward(C.byref(rig), *[_p(t) for t in ins], _p(so), _p(verts), _p(v_shaped), _p(ws), _stream(dev)),
                     "gab_flame_forward")
        ctx.head = head
        ctx.has_so = static_offset is not None
        ctx.shapes = [t.shape for t in (shape, expr, rotation, neck, jaw, eyes, translation)]
        ctx.so_shape = None if static_offset is None else static_offset.shape
        ctx.save_for_backward(*ins, *( [so] if so is not None else []), v_shaped, ws)
        return verts, v_shaped

    @staticmethod
    def backward(ctx, g_verts, g_vshaped):
        lib = _lib.gab()
        head = ctx.head
        rig = _rig_struct(head)
        saved = ctx.saved_tensors
        ins = list(saved[:7])
        so = saved[7] if ctx.has_so else None
        v_shaped, ws = saved[-2], saved[-1]
        dev = v_shaped.device
        V = rig.V
        need = ctx.needs_input_grad  # (head, shape, expr, rotation, neck, jaw, eyes, translation, static_offset)
        f32 = dict(dtype=torch.float32, device=dev)
        d_shape = torch.empty(rig.n_shape, **f32) if need[1] else None
        d_expr = torch.empty(rig.n_expr, **f32)
        d_rot, d_neck, d_jaw = torch.empty(3, **f32), torch.empty(3, **f32), torch.empty(3, **f32)
        d_eyes, d_trans = torch.empty(6, **f32), torch.empty(3, **f32)
        d_so = torch.empty(3 * V, **f32) if (ctx.has_so and need[8]) else None
        scratch = torch.empty(3 * V, **f32)
        gv = torch.zeros((V, 3), **f32) if g_verts is None else _f32(g_verts)
        gvs = None if g_vshaped is None else _f32(g_vshaped)
        with _lib.on_device(dev):
            if ctx.prepared is not None and gvs is None and d_shape is None and d_so is None:
                _chk(lib.gab_flame_backward_prepared(C.byref(rig), _p(ctx.prepared), *[_p(t) for t in ins[2:6]], _p(v_shaped), _p(ws), _p(gv),
                                                     _p(d_expr), _p(d_rot), _p(d_neck), _p(d_jaw), _p(d_eyes), _p(d_trans), _p(scratch),
                                                     0, None, None, _stream(dev)), "gab_flame_backward_prepared")
            else:
                _chk(lib.gab_flame_backward(C.byref(rig), *[_p(t) for t in ins], _p(so), _p(v_shaped), _p(ws), _p(gv), _p(gvs),
                                            _p(d_shape), _p(d_expr), _p(d_rot), _p(d_neck), _p(d_jaw), _p(d_eyes), _p(d_trans), _p(d_so),
                                            _p(scratch), 0, None, None, _stream(dev)), "gab_flame_backward")
        sh = ctx.shapes
        outs = [None, None if d_shape is None else d_shape.view(sh[0]), d_expr.view(sh[1]), d_rot.view(sh[2]), d_neck.view(sh[3]),
                d_jaw.view(sh[4]), d_eyes.view(sh[5]), d_trans.view(sh[6]), None if d_so is None else d_so.view(ctx.so_shape)]
        return tuple(outs)


def flame_forward(head, shape, expr, rotation, neck, jaw, eyes, translation, static_offset=None):
    """-> (verts (1,V,3), v_shaped (1,V,3)); FlameHead.forward with return_verts_cano=True."""
    return _FlameForward.apply(head, shape, expr, rotation, neck, jaw, eyes, translation, static_offset)


class _FlameForwardTimestep(torch.autograd.Function):
    """FLAME forward on row `t` of the per-timestep parameter tables (what select_mesh_by_timestep does,
    scene/flame_gaussian_model.py:117-135) without the seven fancy-index gathers and their index_put
    backward: the kernels read row t in place, the backward writes row t of ONE zero-filled buffer whose
    slices are returned as the gradients of the full (T, k) tables."""

    _ROWS = ("expr", "rotation", "neck_pose", "jaw_pose", "eyes_pose", "translation")

    @staticmethod
    def forward(ctx, head, t, shape, expr, rotation, neck, jaw, eyes, translation, static_offset):
        ctx.set_materialize_grads(False)
        lib = _lib.gab()
        rig = _rig_struct(head)
        dev = head.v_template.device
        V = rig.V
        tabs = [_f32(x) for x in (expr, rotation, neck, jaw, eyes, translation)]
        T = tabs[0].shape[0]
        widths = [int(x.shape[1]) for x in tabs]
        if widths != [rig.n_expr, 3, 3, 3, 6, 3] or any(x.shape[0] != T for x in tabs) or not (0 <= t < T):
            raise RuntimeError("flame_forward_timestep: expected (T,n_expr),(T,3),(T,3),(T,3),(T,6),(T,3) tables and 0 <= t < T")
        sh = _f32(shape).reshape(-1)
        so = None if static_offset is None else _f32(static_offset).reshape(-1)
        rows = [x.data_ptr() + 4 * t * w for x, w in zip(tabs, widths)]
        verts = torch.empty((1, V, 3), dtype=torch.float32, device=dev)
        v_shaped = torch.empty((1, V, 3), dtype=torch.float32, device=dev)
        ws = torch.empty(_lib.GAB_FLAME_WS_FLOATS, dtype=torch.float32, device=dev)
        need = ctx.needs_input_grad  # (head, t, shape, expr, rotation, neck, jaw, eyes, translation, static_offset)
        ctx.prepared = _prepared_rig(head, rig, sh, so) if _use_prepared(head, need[2], so is not None and need[9]) else None
        # gradient-free frames of a sequence (render mode) take their row of the MFMA-made v_shaped table
        seq = _sequence_table(head, rig, ctx.prepared, tabs[0]) if (ctx.prepared is not None and not any(need[2:])) else None
        with _lib.on_device(dev):
            if seq is not None:
                _chk(lib.gab_flame_forward_sequence(C.byref(rig), _p(ctx.prepared), seq.data_ptr() + 4 * t * 3 * V, *rows, _p(verts), _p(v_shaped),
                                                    _p(ws), _stream(dev)), "gab_flame_forward_sequence")
            elif ctx.prepared is not None:
                _chk(lib.gab_flame_forward_prepared(C.byref(rig), _p(ctx.prepared), *rows, _p(verts), _p(v_shaped), _p(ws), _stream(dev)),
                     "gab_flame_forward_prepared")
            else:
                _chk(lib.gab_flame_forward(C.byref(rig), _p(sh), *rows, _p(so), _p(verts), _p(v_shaped), _p(ws), _stream(dev)),
                     "gab_flame_forward")
        ctx.head, ctx.t, ctx.T, ctx.widths = head, t, T, widths
        ctx.has_so = so is not None
        ctx.shape_shape = shape.shape
        ctx.so_shape = None if static_offset is None else static_offset.shape
        ctx.keep = _Keep(sh, *tabs, *([so] if so is not None else []), v_shaped, ws)
        return verts, v_shaped

    @staticmethod
    def backward(ctx, g_verts, g_vshaped):
        lib = _lib.gab()
        rig = _rig_struct(ctx.head)
        saved = ctx.keep.get()
        sh, tabs = saved[0], list(saved[1:7])
        so = saved[7] if ctx.has_so else None
        v_shaped, ws = saved[-2], saved[-1]
        dev = v_shaped.device
        V, T, t, widths = rig.V, ctx.T, ctx.t, ctx.widths
        need = ctx.needs_input_grad  # (head, t, shape, expr, rotation, neck, jaw, eyes, translation, static_offset)
        f32 = dict(dtype=torch.float32, device=dev)
        # full (T,k) gradient tables: separately allocated (autograd can adopt them as .grad without a copy),
        # zero-filled by ONE launch, then row t is written by the backward kernels
        tables = [torch.empty((T, w), **f32) for w in widths]
        ptrs = (C.c_void_p * len(tables))(*[x.data_ptr() for x in tables])
        sizes = (C.c_int32 * len(tables))(*[T * w for w in widths])
        outp = [x.data_ptr() + 4 * t * w for x, w in zip(tables, widths)]
        d_shape = torch.empty(rig.n_shape, **f32) if need[2] else None
        d_so = torch.empty(3 * V, **f32) if (ctx.has_so and need[9]) else None
        scratch = torch.empty(3 * V, **f32)
        gv = torch.zeros((V, 3), **f32) if g_verts is None else _f32(g_verts)
        gvs = None if g_vshaped is None else _f32(g_vshaped)
        rows = [x.data_ptr() + 4 * t * w for x, w in zip(tabs, widths)]
        with _lib.on_device(dev):
            # the (T,k) tables are zero-filled by the backward's first kernel (no launch of their own)
            if ctx.prepared is not None and gvs is None and d_shape is None and d_so is None:
                _chk(lib.gab_flame_backward_prepared(C.byref(rig), _p(ctx.prepared), *rows[1:5], _p(v_shaped), _p(ws), _p(gv), *outp,
                                                     _p(scratch), len(tables), ptrs, sizes, _stream(dev)), "gab_flame_backward_prepared")
            else:
                _chk(lib.gab_flame_backward(C.byref(rig), _p(sh), *rows, _p(so), _p(v_shaped), _p(ws), _p(gv), _p(gvs), _p(d_shape),
                                            *outp, _p(d_so), _p(scratch), len(tables), ptrs, sizes, _stream(dev)), "gab_flame_backward")
        grads = [tb if need[3 + i] else None for i, tb in enumerate(tables)]
        return (None, None, None if d_shape is None else d_shape.view(ctx.shape_shape), *grads,
                None if d_so is None else d_so.view(ctx.so_shape))


def flame_forward_timestep(head, flame_param: dict, t: int):
    """select_mesh_by_timestep's FLAME call on the flame_param dict (npz schema) -> (verts, v_shaped)."""
    fp = flame_param
    return _FlameForwardTimestep.apply(head, int(t), fp["shape"], fp["expr"], fp["rotation"], fp["neck_pose"], fp["jaw_pose"],
                                       fp["eyes_pose"], fp["translation"], fp.get("static_offset"))


# -------------------------------------------------------------------------------------------------
# per-face frames
# -------------------------------------------------------------------------------------------------
class _FaceFrames(torch.autograd.Function):
    @staticmethod
    def forward(ctx, verts, faces):
        ctx.set_materialize_grads(False)
        lib = _lib.gab()
        _need_cuda(verts, "verts")
        dev = verts.device
        v = _f32(verts)
        fi, is64 = _idx(faces)
        V, F = v.shape[0], fi.shape[0]
        f32 = dict(dtype=torch.float32, device=dev)
        center, R = torch.empty((F, 3), **f32), torch.empty((F, 3, 3), **f32)
        scale, quat = torch.empty((F, 1), **f32), torch.empty((F, 4), **f32)
        # the backward scatters into a zeroed (V,3) buffer: let the forward kernel zero it on the side (no memset launch later)
        ctx.d_verts = torch.empty((V, 3), **f32) if ctx.needs_input_grad[0] else None
        with _lib.on_device(dev):
            _chk(lib.gab_face_frames_forward(V, F, _p(v), _p(fi), is64, _p(center), _p(R), _p(scale), _p(quat), _p(ctx.d_verts),
                                             _stream(dev)), "gab_face_frames_forward")
        ctx.keep = _Keep(v, fi)
        ctx.is64 = is64
        return center, R, scale, quat

    @staticmethod
    def backward(ctx, g_center, g_R, g_scale, g_quat):
        lib = _lib.gab()
        v, fi = ctx.keep.get()
        dev = v.device
        V, F = v.shape[0], fi.shape[0]
        d_verts, ctx.d_verts = ctx.d_verts, None    # the pre-zeroed buffer serves one backward; a repeat allocates + memsets
        prepared = d_verts is not None
        if not prepared:
            d_verts = torch.empty((V, 3), dtype=torch.float32, device=dev)
        gs = [None if g is None else _f32(g) for g in (g_center, g_R, g_scale, g_quat)]
        with _lib.on_device(dev):
            _chk(lib.gab_face_frames_backward(V, F, _p(v), _p(fi), ctx.is64, _p(gs[0]), _p(gs[1]), _p(gs[2]), _p(gs[3]), _p(d_verts),
                                              1 if prepared else 0, _stream(dev)), "gab_face_frames_backward")
        return d_verts, None


def face_frames(verts, faces):
    """verts (V,3), faces (F,3) -> face_center (F,3), face_orien_mat (F,3,3), face_scaling (F,1), face_orien_quat (F,4) WXYZ."""
    return _FaceFrames.apply(verts, faces)


# -------------------------------------------------------------------------------------------------
# select_mesh_by_timestep as ONE autograd node (FLAME forward + face frames)
# -------------------------------------------------------------------------------------------------
class _SubCtx:
    """Stand-in for an autograd ctx so that one node can run two of the Functions above back to back (the frame loop is
    ~0.6 ms of Python per step: every autograd node less is ~40 us of host time forward + backward).  The two sub-functions
    keep their tensors as detached aliases (_Keep), so nothing here refers back to the node's outputs."""

    def __init__(self, needs_input_grad):
        self.needs_input_grad = needs_input_grad

    def set_materialize_grads(self, flag):
        pass


def vertex_corner_csr(faces: torch.Tensor, num_verts: int):
    """(vf_begin int32 (V+1,), vf_list int32 (3F, 4)) for gab_mesh_backward_prepared: one row (4 f + c, i0, i1, i2) per corner c of face
    f = (i0, i1, i2), listed vertex by vertex (stable, so a vertex's corners come in face order).  Static per topology: kept ON the faces
    tensor (it lives exactly as long as the topology it describes -- a recorded step has the table's addresses baked in, so a shared cache
    that evicts would pull memory from under a hipGraph: eight recorded lanes did exactly that to a four-entry cache) and rebuilt when
    the tensor is modified in place."""
    hit = getattr(faces, "_gaa_vertex_corners", None)
    if hit is not None and hit[0] == faces._version and hit[1] == num_verts:
        return hit[2]
    f = faces.detach().long()
    flat = f.reshape(-1)
    order = torch.sort(flat, stable=True).indices          # flat corner index 3 f + c, by vertex
    fo, co = order // 3, order % 3
    vf_list = torch.cat([(4 * fo + co)[:, None], f[fo]], 1).to(torch.int32).contiguous()
    vf_begin = torch.zeros(num_verts + 1, dtype=torch.int32, device=flat.device)
    vf_begin[1:] = torch.cumsum(torch.bincount(flat, minlength=num_verts), 0).to(torch.int32)
    faces._gaa_vertex_corners = (faces._version, num_verts, (vf_begin, vf_list))
    return vf_begin, vf_list


def _mesh_backward_mode() -> str:
    """"merged" (default): the face-frame backward and the skinning backward are one launch (gab_mesh_backward_prepared: corner gather,
    no d_verts buffer); "split": the two launches it replaced -- kept for A/B tests and for what the merged entry does not cover."""
    return os.environ.get("GAA_MESH_BWD", "merged")


class _MeshFramesTimestep(torch.autograd.Function):
    @staticmethod
    def forward(ctx, head, t, faces, shape, expr, rotation, neck, jaw, eyes, translation, static_offset):
        ctx.set_materialize_grads(False)
        need = ctx.needs_input_grad
        c1 = _SubCtx((False, False) + tuple(need[3:]))
        verts, v_shaped = _FlameForwardTimestep.forward(c1, head, t, shape, expr, rotation, neck, jaw, eyes, translation, static_offset)
        ctx.merged = c1.prepared is not None and _mesh_backward_mode() == "merged"
        if ctx.merged and faces.device != verts.device:
            raise RuntimeError(f"select_mesh_by_timestep: faces live on {faces.device}, the vertices on {verts.device}")
        if ctx.merged and any(need):
            # the static vertex -> corner table of the merged backward is built HERE (a dozen torch ops on first use, then cached on the
            # faces tensor): inside the first backward it would be recorded into a stream capture of the step
            ctx.corners = vertex_corner_csr(faces, int(verts.shape[1]))
        c2 = _SubCtx((any(need) and not ctx.merged, False))   # merged: no scatter target to pre-zero
        center, R, scale, quat = _FaceFrames.forward(c2, verts[0], faces)
        ctx.c1, ctx.c2 = c1, c2
        ctx.faces = faces
        return verts, v_shaped, center, R, scale, quat

    @staticmethod
    def _backward_merged(ctx, g_verts, g_center, g_R, g_scale, g_quat):
        c1, c2 = ctx.c1, ctx.c2
        lib = _lib.gab()
        rig = _rig_struct(c1.head)
        saved = c1.keep.get()
        tabs = list(saved[1:7])
        v_shaped, ws = saved[-2], saved[-1]
        v, _ = c2.keep.get()
        dev = v.device
        V, T, t, widths = rig.V, c1.T, c1.t, c1.widths
        need = c1.needs_input_grad
        f32 = dict(dtype=torch.float32, device=dev)
        tables = [torch.empty((T, w), **f32) for w in widths]
        ptrs = (C.c_void_p * len(tables))(*[x.data_ptr() for x in tables])
        sizes = (C.c_int32 * len(tables))(*[T * w for w in widths])
        outp = [x.data_ptr() + 4 * t * w for x, w in zip(tables, widths)]
        rows = [x.data_ptr() + 4 * t * w for x, w in zip(tabs, widths)]
        vf_begin, vf_list = ctx.corners
        scratch = torch.empty(3 * V, **f32)
        gs = [None if g is None else _f32(g) for g in (g_center, g_R, g_scale, g_quat)]
        gv = None if g_verts is None else _f32(g_verts)
        with _lib.on_device(dev):
            _chk(lib.gab_mesh_backward_prepared(C.byref(rig), _p(c1.prepared), *rows[1:5], _p(v_shaped), _p(ws), _p(v),
                                                _p(vf_begin), _p(vf_list), _p(gs[0]), _p(gs[1]), _p(gs[2]), _p(gs[3]), _p(gv), *outp,
                                                _p(scratch), len(tables), ptrs, sizes, _stream(dev)), "gab_mesh_backward_prepared")
        grads = [tb if need[3 + i] else None for i, tb in enumerate(tables)]
        return (None, None, None, None, *grads, None)

    @staticmethod
    def backward(ctx, g_verts, g_vshaped, g_center, g_R, g_scale, g_quat):
        if ctx.merged and g_vshaped is None:
            return _MeshFramesTimestep._backward_merged(ctx, g_verts, g_center, g_R, g_scale, g_quat)
        d_verts, _ = _FaceFrames.backward(ctx.c2, g_center, g_R, g_scale, g_quat)
        if g_verts is not None:
            d_verts = d_verts + g_verts.reshape(d_verts.shape)
        grads = _FlameForwardTimestep.backward(ctx.c1, d_verts.view(1, -1, 3), g_vshaped)
        return (None, None, None) + tuple(grads[2:])


_RIG_BUFFERS = ("v_template", "shapedirs", "posedirs", "J_regressor", "lbs_weights")


def _plain(t) -> bool:
    return t.is_cuda and t.dtype is torch.float32 and t.is_contiguous()


def _mesh_plan(H, head, fp: dict, faces):
    """The frame-independent half of select_mesh_by_timestep for the compiled host (csrc/gaa_host.cpp: MeshPlan): rig struct, prepared rig, topology
    tables -- made once, kept on the head, and re-made when anything it was made from is replaced or modified in place (object identity + version
    counters of shape / static_offset / faces, identity of the five rig buffers, the two mode switches).  None: this call is outside what the native
    node takes (a gradient for shape or static_offset, the classic three-kernel FLAME, the split backward, the sequence table, other dtypes)."""
    shape, so = fp["shape"], fp.get("static_offset")
    bufs = head.__dict__.get("_buffers")                      # an nn.Module's registered buffers (no __getattr__ walk), else plain attributes
    if bufs is None or any(n not in bufs for n in _RIG_BUFFERS):
        bufs = {n: getattr(head, n) for n in _RIG_BUFFERS}
    env = (os.environ.get("GAA_MESH_BWD"), os.environ.get("GAA_MESH_SEQUENCE"))
    st = head.__dict__.get("_gaa_mesh_plan")
    if st is not None:
        (plan, k_shape, v_shape, k_so, v_so, k_faces, v_faces, k_bufs, k_env, k_flags) = st
        if (k_shape is shape and v_shape == shape._version and k_so is so and (so is None or v_so == so._version) and k_faces is faces and v_faces == faces._version
                and k_env == env and k_flags == (shape.requires_grad, so is not None and so.requires_grad, getattr(head, "flame_impl", "prepared"))
                and all(bufs[n] is b for n, b in zip(_RIG_BUFFERS, k_bufs))):
            return plan
    plan = None
    flags = (shape.requires_grad, so is not None and so.requires_grad, getattr(head, "flame_impl", "prepared"))
    ok = (not flags[0] and not flags[1] and flags[2] != "classic" and _mesh_backward_mode() == "merged" and _sequence_mode() == "off"
          and all(_plain(bufs[n]) for n in _RIG_BUFFERS) and faces.is_cuda and faces.is_contiguous() and faces.dtype in (torch.int32, torch.int64)
          and faces.device == bufs["v_template"].device)
    if ok:
        rig = _rig_struct(head)
        sh = _f32(shape).reshape(-1)
        sof = None if so is None else _f32(so).reshape(-1)
        if sh.numel() == rig.n_shape and (sof is None or sof.numel() == 3 * rig.V):
            prepared = _prepared_rig(head, rig, sh, sof)
            vf_begin, vf_list = vertex_corner_csr(faces, int(rig.V))
            plan = H.make_mesh_plan(*[bufs[n] for n in _RIG_BUFFERS], [int(x) for x in head.parents.tolist()], int(head.n_shape_params), prepared, faces, vf_begin, vf_list)
    head.__dict__["_gaa_mesh_plan"] = (plan, shape, shape._version, so, None if so is None else so._version, faces, faces._version, tuple(bufs[n] for n in _RIG_BUFFERS), env, flags)
    return plan


def mesh_frames_timestep(head, flame_param: dict, t: int, faces):
    """select_mesh_by_timestep + update_mesh_properties in one autograd node:
    -> (verts (1,V,3), verts_cano (1,V,3), face_center, face_orien_mat, face_scaling, face_orien_quat)."""
    fp = flame_param
    H = _host.get()
    if H is not None:
        expr, rot, neck, jaw, eyes, trans = fp["expr"], fp["rotation"], fp["neck_pose"], fp["jaw_pose"], fp["eyes_pose"], fp["translation"]
        if _plain(expr) and _plain(rot) and _plain(neck) and _plain(jaw) and _plain(eyes) and _plain(trans):
            plan = _mesh_plan(H, head, fp, faces)
            if plan is not None:
                return H.mesh_frames(plan, expr, rot, neck, jaw, eyes, trans, int(t))
    return _MeshFramesTimestep.apply(head, int(t), faces, fp["shape"], fp["expr"], fp["rotation"], fp["neck_pose"], fp["jaw_pose"],
                                     fp["eyes_pose"], fp["translation"], fp.get("static_offset"))


# -------------------------------------------------------------------------------------------------
# per-splat local -> world
# -------------------------------------------------------------------------------------------------
def binding_csr(binding: torch.Tensor, num_faces: int):
    """(order int32 (N,), face_begin int32 (F+1,), splat_face int32 (N,), slot int32 (N,)) for gab_bind_backward_csr;
    depends on `binding` only.  slot is the inverse permutation of order (the splat's position in the CSR)."""
    b = binding.detach().long()
    order64 = torch.sort(b, stable=True).indices
    order = order64.to(torch.int32).contiguous()
    counts = torch.bincount(b, minlength=num_faces)
    face_begin = torch.zeros(num_faces + 1, dtype=torch.int32, device=b.device)
    face_begin[1:] = torch.cumsum(counts, 0).to(torch.int32)
    slot = torch.empty_like(order)
    slot[order64] = torch.arange(order.numel(), dtype=torch.int32, device=b.device)
    return order, face_begin, b.to(torch.int32).contiguous(), slot


class _BindSplats(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz, log_scaling, rotation, binding, face_R, face_scale, face_center, face_quat, csr=None, opacity_logit=None):
        lib = _lib.gab()
        _need_cuda(xyz, "_xyz")
        dev = xyz.device
        x, ls, q = _f32(xyz), _f32(log_scaling), _f32(rotation)
        b, is64 = _idx(binding)
        fR, fs, fc, fq = _f32(face_R), _f32(face_scale), _f32(face_center), _f32(face_quat)
        N, F = x.shape[0], fc.shape[0]
        f32 = dict(dtype=torch.float32, device=dev)
        ox, osc, oq = torch.empty((N, 3), **f32), torch.empty((N, 3), **f32), torch.empty((N, 4), **f32)
        ol = None if opacity_logit is None else _f32(opacity_logit)
        oo = None if ol is None else torch.empty_like(ol)
        with _lib.on_device(dev):
            _chk(lib.gab_bind_forward(N, F, _p(x), _p(ls), _p(q), _p(b), is64, _p(fc), _p(fR), _p(fs), _p(fq), _p(ox), _p(osc), _p(oq),
                                      _p(ol), _p(oo), _stream(dev)), "gab_bind_forward")
        ctx.keep = _Keep(x, ls, q, b, fR, fs, fc, fq, oo)
        ctx.is64 = is64
        if csr is not None and (csr[0].numel() != N or csr[1].numel() != F + 1 or any(t.numel() != N for t in csr[2:4])):
            raise RuntimeError(f"binding CSR does not match this call: {csr[0].numel()} ordered splats / {csr[1].numel() - 1} faces, "
                               f"expected {N} / {F} (rebuild it with binding_csr after the binding changed)")
        ctx.csr = csr
        if oo is None:
            return ox, osc, oq
        return ox, osc, oq, oo

    @staticmethod
    def backward(ctx, g_xyz, g_scaling, g_rot, g_opacity=None):
        lib = _lib.gab()
        x, ls, q, b, fR, fs, fc, fq, oo = ctx.keep.get()
        dev = x.device
        N, F = x.shape[0], fc.shape[0]
        f32 = dict(dtype=torch.float32, device=dev)
        d_x, d_ls, d_q = torch.empty((N, 3), **f32), torch.empty((N, 3), **f32), torch.empty((N, 4), **f32)
        d_face = torch.empty(17 * F, **f32)   # four contiguous blocks: center | orien_mat | scaling | orien_quat
        gs = [None if g is None else _f32(g) for g in (g_xyz, g_scaling, g_rot)]
        go = None if (oo is None or g_opacity is None) else _f32(g_opacity)
        d_ol = None if oo is None else torch.empty_like(oo)
        with _lib.on_device(dev):
            if ctx.csr is not None:
                order, face_begin = ctx.csr[0], ctx.csr[1]
                splat_face, slot = (ctx.csr[2], ctx.csr[3]) if len(ctx.csr) >= 4 else (None, None)
                rows = torch.empty(_lib.GAB_BIND_ROW_FLOATS * N, **f32) if slot is not None else None   # two-pass scratch
                _chk(lib.gab_bind_backward_csr(N, F, _p(x), _p(ls), _p(q), _p(fR), _p(fs), _p(fq), _p(gs[0]), _p(gs[1]), _p(gs[2]),
                                               _p(order), _p(face_begin), _p(d_x), _p(d_ls), _p(d_q), _p(d_face), _p(oo), _p(go), _p(d_ol),
                                               _p(splat_face), _p(slot), _p(rows), _stream(dev)), "gab_bind_backward_csr")
            else:
                _chk(lib.gab_bind_backward(N, F, _p(x), _p(ls), _p(q), _p(b), ctx.is64, _p(fc), _p(fR), _p(fs), _p(fq), _p(gs[0]),
                                           _p(gs[1]), _p(gs[2]), _p(d_x), _p(d_ls), _p(d_q), _p(d_face), _p(oo), _p(go), _p(d_ol),
                                           _stream(dev)), "gab_bind_backward")
        return (d_x, d_ls, d_q, None, d_face[3 * F: 12 * F].view(F, 3, 3), d_face[12 * F: 13 * F].view(F, 1), d_face[: 3 * F].view(F, 3),
                d_face[13 * F:].view(F, 4), None, d_ol)


def bind_splats(xyz, log_scaling, rotation, binding, face_R, face_scale, face_center, face_quat, csr=None, opacity_logit=None):
    """-> (get_xyz, get_scaling, get_rotation) of a mesh-bound GaussianModel, in one kernel.
    csr = binding_csr(binding, F) selects the atomic-free deterministic backward.
    opacity_logit = the model's `_opacity`: a fourth output, get_opacity = sigmoid(_opacity), from the same launches."""
    return _BindSplats.apply(xyz, log_scaling, rotation, binding, face_R, face_scale, face_center, face_quat, csr, opacity_logit)
