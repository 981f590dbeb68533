"""Zero-edit model-side boundary: rebinds the per-frame methods of the reference's OWN classes to
the fused MI355X path, so train.py / render.py / fps_benchmark_*.py run unchanged.

    import gaussianavatars_amd.patch as P;  P.patch_reference()          # from inside a reference checkout
    python -m gaussianavatars_amd.run fps_benchmark_demo.py --point_path ...   # the same, as a launcher

What is rebound (attribute names, lazy initialisation and autograd connectivity preserved):

    FlameGaussianModel.select_mesh_by_timestep   scene/flame_gaussian_model.py:117-135  -> gab_flame_* + gab_face_frames_* (one node)
    FlameGaussianModel.update_mesh_properties    scene/flame_gaussian_model.py:137-154  -> gab_face_frames_*
    FlameHead.forward (batch 1, no landmarks)    flame_model/flame.py:485-558           -> gab_flame_*  (update_mesh_by_param_dict, :91-115)
    GaussianModel.get_xyz / get_scaling / get_rotation / get_opacity   scene/gaussian_model.py:113-160 -> gab_bind_* (one launch for all four)
    GaussianModel.get_features_split (new)       the two SH leaf tensors, read in place by gsr_forward_ex (no per-frame cat, :152-156)
    gaussian_renderer.render (optional)          gaussian_renderer/__init__.py:19-101   -> the mirror with the split-SH / leaf fast paths

`patch_classes` works on any class pair with the reference's attribute names; the repository's own mirror classes
(gaussianavatars_amd/gaussian_model.py) are written in composed torch like the reference and go through the very same
patch, which is how the GPU tests exercise it (the reference checkout does not travel to the GPU box).

There is no CPU fallback: a patched method called on host tensors raises the binding library's "no CPU implementation"
error.  `obj.binding_impl = "unfused"` on an instance is an explicit opt-out that runs the original composed-torch methods
(the A/B leg of bench.py and of the parity tests).
"""
from __future__ import annotations

import os
import sys

import torch

from . import binding as fused

_ORIG: dict = {}   # (class, attribute) -> the original attribute, for unpatch() and the "unfused" opt-out


def _default_impl() -> str:
    """GAA_BINDING_IMPL=unfused: objects that do not say otherwise keep the reference's composed-torch methods (CPU tensors, A/B runs)."""
    return os.environ.get("GAA_BINDING_IMPL", "fused")


def _fused(self) -> bool:
    return getattr(self, "binding_impl", _default_impl()) != "unfused"


# ------------------------------------------------------------------------------------------------
# bound accessors: one evaluation per (mesh update, leaf versions), shared by the four properties
# ------------------------------------------------------------------------------------------------
class _BoundCache:
    __slots__ = ("leaves", "versions", "faces", "binding", "binding_version", "grad", "out")


def _same(cache, leaves, faces, binding, grad) -> bool:
    # identity + version of objects the cache itself keeps alive: no id() reuse, no stale hit after densification
    return (cache is not None and cache.grad == grad and cache.binding is binding and cache.binding_version == binding._version
            and all(a is b for a, b in zip(cache.leaves, leaves)) and all(a is b for a, b in zip(cache.faces, faces))
            and cache.versions == tuple(t._version for t in leaves + faces))


def binding_csr_cached(self, num_faces: int):
    """The per-face CSR of `self.binding` for the atomic-free bind backward; rebuilt whenever the binding tensor is
    replaced (densification: cat / prune), modified in place, or the face count changes."""
    b = self.binding
    hit = getattr(self, "_gaa_csr", None)
    if hit is None or hit[0] is not b or hit[1] != b._version or hit[2] != num_faces or hit[3][0].numel() != b.shape[0]:
        hit = (b, b._version, num_faces, fused.binding_csr(b, num_faces))
        self._gaa_csr = hit
    return hit[3]


def bound(self):
    """(get_xyz, get_scaling, get_rotation, get_opacity) of a mesh-bound model from ONE launch.  The reference recomputes
    each accessor on every call (get_xyz twice per render(), gaussian_renderer/__init__.py:27,54); the values only change
    when the mesh, the binding or a leaf does, so they are cached on exactly that.  The shared autograd node keeps its
    inputs for its whole life (binding._Keep), so it can be walked by several backward passes like the reference's
    per-call graphs can."""
    leaves = (self._xyz, self._scaling, self._rotation, self._opacity)
    faces = (self.face_orien_mat, self.face_scaling, self.face_center, self.face_orien_quat)
    grad = torch.is_grad_enabled()
    cache = getattr(self, "_gaa_bound", None)
    if not _same(cache, leaves, faces, self.binding, grad):
        cache = _BoundCache()
        cache.leaves, cache.faces, cache.binding, cache.grad = leaves, faces, self.binding, grad
        cache.binding_version = self.binding._version
        cache.versions = tuple(t._version for t in leaves + faces)
        csr = binding_csr_cached(self, self.face_center.shape[0])
        cache.out = fused.bind_splats(self._xyz, self._scaling, self._rotation, self.binding, self.face_orien_mat, self.face_scaling,
                                      self.face_center, self.face_orien_quat, csr=csr, opacity_logit=self._opacity)
        self._gaa_bound = cache
    return cache.out


def _leaves_complete(self) -> bool:
    """The four leaves the one-launch bind reads all hold N rows.  Not so in the middle of GaussianModel.create_from_pcd
    (scene/gaussian_model.py:185-206 reads `self.get_xyz.shape[0]` after setting `_xyz` and before `_scaling` / `_rotation` / `_opacity` exist):
    such a call takes the reference's own per-accessor code, once, at initialisation."""
    n = self._xyz.shape[0]
    return n > 0 and self._scaling.shape[0] == n and self._rotation.shape[0] == n and self._opacity.shape[0] == n


def _make_accessor(cls, name: str, index: int, mesh_attr: str):
    orig = _ORIG[(cls, name)]

    def getter(self):
        if self.binding is None or not _fused(self) or not _leaves_complete(self):
            return orig.fget(self)
        if getattr(self, mesh_attr) is None:          # same lazy initialisation as the reference (:119-120,131-132,146-147)
            self.select_mesh_by_timestep(0)
        return bound(self)[index]

    getter.__name__ = name
    getter.__doc__ = f"fused {name} (gaussianavatars_amd.patch); original: {cls.__module__}.{cls.__name__}.{name}"
    return property(getter)


def _make_opacity(cls):
    orig = _ORIG[(cls, "get_opacity")]

    def get_opacity(self):
        if self.binding is None or not _fused(self) or self.face_center is None or not _leaves_complete(self):
            return orig.fget(self)
        return bound(self)[3]   # sigmoid(_opacity) rides in the bind kernel (same values, one launch fewer)

    return property(get_opacity)


def _get_features_split(self):
    """(dc (N,1,3), rest (N,K,3)): the two SH leaf tensors for the rasterizer's split-SH entry (None without a rest block)."""
    return (self._features_dc, self._features_rest) if self._features_rest.shape[1] > 0 else None


# ------------------------------------------------------------------------------------------------
# mesh update
# ------------------------------------------------------------------------------------------------
def _select_mesh_by_timestep(self, timestep, original=False):
    if not _fused(self):
        return _ORIG[(type(self)._gaa_patched_base, "select_mesh_by_timestep")](self, timestep, original)
    self.timestep = timestep
    fp = self.flame_param_orig if original and self.flame_param_orig is not None else self.flame_param
    faces = self.flame_model.faces
    verts, verts_cano, c, R, s, q = fused.mesh_frames_timestep(_prepared(self.flame_model), fp, timestep, faces)
    self.face_center, self.face_orien_mat, self.face_scaling, self.face_orien_quat = c, R, s, q
    self.verts, self.faces, self.verts_cano = verts, faces, verts_cano


def _update_mesh_properties(self, verts, verts_cano):
    if not _fused(self):
        return _ORIG[(type(self)._gaa_patched_base, "update_mesh_properties")](self, verts, verts_cano)
    faces = self.flame_model.faces
    c, R, s, q = fused.face_frames(verts.squeeze(0), faces)
    self.face_center, self.face_orien_mat, self.face_scaling, self.face_orien_quat = c, R, s, q
    self.verts, self.faces, self.verts_cano = verts, faces, verts_cano


def _prepared(head):
    """The kernels read the rig buffers in place: make sure they are contiguous fp32 (they are for the reference's FlameHead;
    a re-registered copy otherwise) and that the head carries the `n_shape_params` the rig struct is built from."""
    if getattr(head, "_gaa_prepared", False):
        return head
    for name in ("v_template", "shapedirs", "posedirs", "J_regressor", "lbs_weights"):
        b = getattr(head, name)
        if b.dtype != torch.float32 or not b.is_contiguous():
            head.register_buffer(name, b.float().contiguous(), persistent=name in head.state_dict())
    head._gaa_prepared = True
    return head


def _make_flame_forward(cls):
    orig = _ORIG[(cls, "forward")]

    def forward(self, shape, expr, rotation, neck, jaw, eyes, translation, zero_centered_at_root_node=False,
                return_landmarks=True, return_verts_cano=False, static_offset=None, dynamic_offset=None):
        fusable = (getattr(self, "impl", _default_impl()) != "unfused" and not zero_centered_at_root_node and not return_landmarks
                   and shape.shape[0] == 1)
        if not fusable:
            kw = dict(zero_centered_at_root_node=zero_centered_at_root_node, return_landmarks=return_landmarks,
                      return_verts_cano=return_verts_cano, static_offset=static_offset, dynamic_offset=dynamic_offset)
            return orig(self, shape, expr, rotation, neck, jaw, eyes, translation, **kw)
        verts, v_shaped = fused.flame_forward(_prepared(self), shape, expr, rotation, neck, jaw, eyes, translation, static_offset)
        return [verts, v_shaped] if return_verts_cano else verts

    return forward


# ------------------------------------------------------------------------------------------------
def _hook_spatial_order(G) -> None:
    """Keeps the splats of a model of the REFERENCE's class in Morton order of their positions (gaussian_model.spatial_order_default:
    GAA_SPATIAL_SORT=0 opts out): after `load_ply` (scene/gaussian_model.py:282-332; the reference's FlameGaussianModel.load_ply calls it
    through super()) and after every `densify_and_prune` (:501-515, called from train.py:197-206), which appends its new splats at the end.
    The order is a layout choice -- images and gradients do not depend on it -- that lets a workgroup of the binning pass meet a compact set
    of tiles.  gaussian_model.spatial_resort moves the six leaves, their Adam moments (through the model's own
    _prune_optimizer), the densification statistics and the binding together.  Classes whose load_ply takes `spatial_sort` itself (this
    package's mirror) are left alone."""
    import inspect

    load = G.__dict__.get("load_ply")
    if load is not None and "spatial_sort" not in inspect.signature(load).parameters:
        _ORIG[(G, "load_ply")] = load

        def load_ply(self, *a, **k):
            out = load(self, *a, **k)
            from .gaussian_model import spatial_order_default, spatial_resort

            if getattr(self, "_xyz", None) is not None:
                self._gaa_order = torch.arange(self._xyz.shape[0], device=self._xyz.device)   # row i of the model = row i of the file, so far
                self._gaa_order_lost = False
            if spatial_order_default() and getattr(self, "_xyz", None) is not None and self._xyz.shape[0] > 1:
                spatial_resort(self)
            return out

        load_ply.__doc__ = load.__doc__
        G.load_ply = load_ply
    # `_gaa_order` (row i of the model = row _gaa_order[i] of the loaded file, -1 for splats born since) follows the rows through the reference's
    # own pruning and appending (scene/gaussian_model.py:371-399, :426-444); a row count that changed any other way drops it (None) instead of
    # leaving a permutation that points at the wrong rows
    prune = G.__dict__.get("prune_points")
    if prune is not None:
        _ORIG[(G, "prune_points")] = prune

        def prune_points(self, mask):
            order, n = getattr(self, "_gaa_order", None), self._xyz.shape[0]
            out = prune(self, mask)        # (narrows `mask` in place to the splats it really removes: every face keeps one)
            tracked = isinstance(order, torch.Tensor) and order.shape[0] == n and mask.shape[0] == n
            self._gaa_order = order.to(mask.device)[~mask] if tracked else None
            if self._gaa_order is None or self._gaa_order.shape[0] != self._xyz.shape[0]:
                self._gaa_order, self._gaa_order_lost = None, True
            return out

        prune_points.__doc__ = prune.__doc__
        G.prune_points = prune_points
    postfix = G.__dict__.get("densification_postfix")
    if postfix is not None:
        _ORIG[(G, "densification_postfix")] = postfix

        def densification_postfix(self, *a, **k):
            order, n = getattr(self, "_gaa_order", None), self._xyz.shape[0]
            out = postfix(self, *a, **k)   # (appends the new splats behind the old ones)
            grown = self._xyz.shape[0] - n
            if isinstance(order, torch.Tensor) and order.shape[0] == n and grown >= 0:
                self._gaa_order = torch.cat([order, torch.full((grown,), -1, dtype=order.dtype, device=order.device)])
            else:
                self._gaa_order, self._gaa_order_lost = None, True
            return out

        densification_postfix.__doc__ = postfix.__doc__
        G.densification_postfix = densification_postfix
    dens = G.__dict__.get("densify_and_prune")
    if dens is not None:
        _ORIG[(G, "densify_and_prune")] = dens

        def densify_and_prune(self, *a, **k):
            out = dens(self, *a, **k)
            from .gaussian_model import spatial_order_default, spatial_resort

            if spatial_order_default() and self._xyz.shape[0] > 1:
                spatial_resort(self)
            return out

        densify_and_prune.__doc__ = dens.__doc__
        G.densify_and_prune = densify_and_prune


def patch_classes(gaussian_model_cls, flame_gaussian_model_cls=None, flame_head_cls=None) -> None:
    """Rebinds the per-frame methods of classes shaped like the reference's (idempotent)."""
    G = gaussian_model_cls
    if not getattr(G, "_gaa_patched", False):
        for name in ("get_xyz", "get_scaling", "get_rotation", "get_opacity"):
            _ORIG[(G, name)] = G.__dict__[name]
        G.get_xyz = _make_accessor(G, "get_xyz", 0, "face_center")
        G.get_scaling = _make_accessor(G, "get_scaling", 1, "face_scaling")
        G.get_rotation = _make_accessor(G, "get_rotation", 2, "face_orien_quat")
        G.get_opacity = _make_opacity(G)
        if "get_features_split" not in G.__dict__:
            G.get_features_split = property(_get_features_split)
        _hook_spatial_order(G)
        G._gaa_patched = True
    F = flame_gaussian_model_cls
    if F is not None and not F.__dict__.get("_gaa_patched_flame", False):
        for name in ("select_mesh_by_timestep", "update_mesh_properties"):
            _ORIG[(F, name)] = F.__dict__[name]
        F._gaa_patched_base = F
        F.select_mesh_by_timestep = _select_mesh_by_timestep
        F.update_mesh_properties = _update_mesh_properties
        F._gaa_patched_flame = True
    H = flame_head_cls
    if H is not None and not H.__dict__.get("_gaa_patched_head", False):
        _ORIG[(H, "forward")] = H.__dict__["forward"]
        H.forward = _make_flame_forward(H)
        H._gaa_patched_head = True


def unpatch_classes(*classes) -> None:
    for (cls, name), orig in list(_ORIG.items()):
        if cls in classes:
            setattr(cls, name, orig)
            del _ORIG[(cls, name)]
    for cls in classes:
        for flag in ("_gaa_patched", "_gaa_patched_flame", "_gaa_patched_head", "_gaa_patched_base", "_gaa_patched_stats"):
            if flag in cls.__dict__:
                delattr(cls, flag)
        if isinstance(cls.__dict__.get("get_features_split"), property) and cls.__dict__["get_features_split"].fget is _get_features_split:
            delattr(cls, "get_features_split")


def patch_reference(reference_root: str | None = None, fast_render: bool = True, stub_torchvision: bool = True, pin: bool = True) -> dict:
    """Makes a checkout of the reference run on MI355X without editing it.

    1. registers the import shims the reference's unconditional imports need (gaussianavatars_amd.shims) and puts this
       repository's root on sys.path so that `diff_gaussian_rasterization` resolves to the HIP rasterizer;
    2. imports the reference's `scene.gaussian_model`, `scene.flame_gaussian_model`, `flame_model.flame` and rebinds their
       per-frame methods (patch_classes);
    3. with fast_render, replaces `gaussian_renderer.render` by the mirror with the same signature and return dict
       (split-SH read in place, screen-space leaf without the zeros_like + 0 + retain_grad round trip).
    4. with pin (default; GAA_PIN=0 opts out), moves the process next to its GPU (frame_parallel.pin_host_process: the host conditions
       bench.py measures under -- the frame loop is host-paced; forked DataLoader workers keep the original CPU mask).
    5. (GAA_LOSS_SEED=0 opts out) installs loss.install_backward_seed: train.py's `loss.backward()` seeds the backward pass with a cached
       device 1 instead of a one-element fill launch.
    6. (GAA_FUSED_LOSS=0 opts out) patch_loss_and_stats: `utils.loss_utils.l1_loss` / `ssim` and `GaussianModel.add_densification_stats`
       -> the fused kernels of include/gls.h (train.py:131-132,198 unchanged).
    Returns {'shims': [...], 'classes': [...], 'render': bool, 'pinned_cpus': [...] | None, 'backward_seed': bool, 'loss': [...]}.  Call it
    before the entry script imports `render`."""
    from . import shims

    repo_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if repo_root not in sys.path:
        sys.path.insert(0, repo_root)
    if reference_root is not None and reference_root not in sys.path:
        sys.path.insert(0, reference_root)
    served = shims.install(stub_torchvision=stub_torchvision)
    import importlib

    gm = importlib.import_module("scene.gaussian_model")
    fgm = importlib.import_module("scene.flame_gaussian_model")
    flame = importlib.import_module("flame_model.flame")
    patch_classes(gm.GaussianModel, fgm.FlameGaussianModel, flame.FlameHead)
    did_render = False
    if fast_render:
        gr = importlib.import_module("gaussian_renderer")
        from .gaussian_renderer import render as fast

        if gr.render is not fast:
            _ORIG[(gr, "render")] = gr.render
            gr.render = fast
        did_render = True
    pinned = None
    if pin:
        from .frame_parallel import pin_host_process

        pinned = pin_host_process()
    seeded = False
    if os.environ.get("GAA_LOSS_SEED", "1") != "0":
        from .loss import install_backward_seed

        install_backward_seed()
        seeded = True
    fused_loss = patch_loss_and_stats(gm.GaussianModel) if os.environ.get("GAA_FUSED_LOSS", "1") != "0" else []
    return dict(shims=served, classes=[gm.GaussianModel, fgm.FlameGaussianModel, flame.FlameHead], render=did_render, pinned_cpus=pinned,
                backward_seed=seeded, loss=fused_loss)


def patch_loss_and_stats(gaussian_model_cls=None, loss_utils=None) -> list:
    """SURVEY.md 8(f) N3 behind the zero-edit boundary: the training-step neighbours of the render path, rebound BEFORE the entry script
    imports them (train.py:22 `from utils.loss_utils import l1_loss, ssim`):

        utils.loss_utils.l1_loss / ssim              utils/loss_utils.py:17-18,36-63 -> loss.l1_loss_paired / loss.ssim_paired: train.py:131-132's two
                                                     calls on the same (image, gt) become ONE fused pass (gls_l1_ssim_*) from the second iteration on
        GaussianModel.add_densification_stats        scene/gaussian_model.py:517-519 (train.py:198) -> gls_add_densification_stats, one launch

    Host tensors / other dtypes / other window sizes keep the reference's own functions (kept on the module as `_gaa_orig_*`).  GAA_FUSED_LOSS=0
    opts out.  Returns the names rebound (idempotent)."""
    import importlib

    from . import loss as L

    done = []
    lu = loss_utils if loss_utils is not None else importlib.import_module("utils.loss_utils")
    if not getattr(lu, "_gaa_patched", False):
        orig_l1, orig_ssim = lu.l1_loss, lu.ssim
        _ORIG[(lu, "l1_loss")], _ORIG[(lu, "ssim")] = orig_l1, orig_ssim

        def l1_loss(network_output, gt):
            if not (isinstance(network_output, torch.Tensor) and network_output.is_cuda and network_output.dtype is torch.float32
                    and isinstance(gt, torch.Tensor) and gt.is_cuda and gt.dtype is torch.float32):
                return orig_l1(network_output, gt)
            return L.l1_loss_paired(network_output, gt)

        def ssim(img1, img2, window_size=11, size_average=True):
            if not (window_size == 11 and isinstance(img1, torch.Tensor) and img1.is_cuda and img1.dtype is torch.float32 and img1.dim() in (3, 4)
                    and isinstance(img2, torch.Tensor) and img2.is_cuda and img2.dtype is torch.float32 and img1.shape == img2.shape):
                return orig_ssim(img1, img2, window_size, size_average)
            return L.ssim_paired(img1, img2, window_size, size_average)

        l1_loss.__doc__, ssim.__doc__ = L.l1_loss_paired.__doc__, L.ssim_paired.__doc__
        lu._gaa_orig_l1_loss, lu._gaa_orig_ssim = orig_l1, orig_ssim
        lu.l1_loss, lu.ssim = l1_loss, ssim
        lu._gaa_patched = True
        done += ["utils.loss_utils.l1_loss", "utils.loss_utils.ssim"]
    G = gaussian_model_cls
    if G is not None and "add_densification_stats" in G.__dict__ and not G.__dict__.get("_gaa_patched_stats", False):
        _ORIG[(G, "add_densification_stats")] = G.__dict__["add_densification_stats"]
        G.add_densification_stats = L.add_densification_stats
        G._gaa_patched_stats = True
        done.append(f"{G.__name__}.add_densification_stats")
    return done
