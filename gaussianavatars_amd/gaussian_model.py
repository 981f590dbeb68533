"""Host-side mirror of the model half of the hot path: the accessors of
scene/gaussian_model.py:113-163 and the mesh update of scene/flame_gaussian_model.py:91-154, with
the same names, attributes and lazy-init behaviour, so a caller written against the reference
(`render()`, train.py's inner loop, fps_benchmark_demo.py) reads the same.

The classes below state the per-frame methods in COMPOSED TORCH, the way the reference's own classes do, and then go
through `gaussianavatars_amd.patch.patch_classes` at the bottom of this file -- the very call that rebinds the
reference's classes in a checkout (`patch_reference()`).  `binding_impl="unfused"` on an instance opts out of the
fused kernels (the A/B leg of bench.py and the parity tests).  /root/reference does not travel to the GPU box; these
stand-ins are what the GPU tests and bench.py drive, through the same patched methods.

Only the per-frame path is mirrored.  Optimiser surgery, densification and dataset loading are out of scope
 and stay with the reference's own files.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch
from torch import nn

from . import unfused


class GaussianModel:
    """Accessor surface of the reference GaussianModel (scene/gaussian_model.py:52-163)."""

    def __init__(self, sh_degree: int, binding_impl: str = "fused"):
        self.active_sh_degree = 0
        self.max_sh_degree = sh_degree
        self._xyz = torch.empty(0)
        self._features_dc = torch.empty(0)
        self._features_rest = torch.empty(0)
        self._scaling = torch.empty(0)
        self._rotation = torch.empty(0)
        self._opacity = torch.empty(0)
        self.max_radii2D = torch.empty(0)
        # mesh binding (reference :64-72)
        self.face_center = None
        self.face_scaling = None
        self.face_orien_mat = None
        self.face_orien_quat = None
        self.binding = None
        self.binding_counter = None
        self.timestep = None
        self.num_timesteps = 1
        if binding_impl not in ("fused", "unfused"):
            raise ValueError("binding_impl must be 'fused' or 'unfused'")
        self.binding_impl = binding_impl

    # ---- parameter loading (synthetic stand-in for load_ply / create_from_pcd) ----------------
    def load_arrays(self, arrs: Dict[str, np.ndarray], device="cuda", requires_grad: bool = True):
        """arrs uses the reference's leaf names (_xyz, _features_dc, _features_rest, _scaling, _rotation,
        _opacity, optional binding).  Like load_ply (:323) this activates the full SH degree."""
        for k in ("_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity"):
            src = arrs[k]
            if isinstance(src, np.ndarray) and not src.flags.writeable:   # (io.load_ply maps the file read-only: a host tensor must not alias it)
                src = np.array(src)
            t = torch.as_tensor(src, dtype=torch.float32, device=device).contiguous()
            setattr(self, k, nn.Parameter(t.requires_grad_(requires_grad)))
        if arrs.get("binding") is not None:
            self.binding = torch.as_tensor(arrs["binding"], device=device)
            nf = int(self.binding.max().item()) + 1
            self.binding_counter = torch.bincount(self.binding.long(), minlength=nf).int()
        self.active_sh_degree = self.max_sh_degree
        self.max_radii2D = torch.zeros((self._xyz.shape[0]), device=device)
        # densification statistics, as training_setup allocates them (scene/gaussian_model.py:210-211)
        self.xyz_gradient_accum = torch.zeros((self._xyz.shape[0], 1), device=device)
        self.denom = torch.zeros((self._xyz.shape[0], 1), device=device)

    def load_ply(self, path, device="cuda", spatial_sort: Optional[bool] = None, face_centers=None, **kwargs):
        """scene/gaussian_model.py:282-332: leaf tensors from the reference's PLY (binding -> int32).  `spatial_sort` (round 4: ON by
        default, `GAA_SPATIAL_SORT=0` or `spatial_sort=False` keep the file's order): the splats in Morton order of their positions
        (io.spatial_sort; a bound model passes the template's face centres) -- same images, every per-splat tensor permuted the same way
        (row i is then NOT the file's row i), the binning pass of large models ~1.6 x faster because a workgroup's consecutive splats share
        tiles.  save_ply writes whatever order the model is in."""
        from . import io as gio

        arrs = gio.load_ply(str(path), self.max_sh_degree)
        if spatial_sort is None:
            spatial_sort = spatial_order_default()
        self._gaa_order = None
        if spatial_sort:
            arrs, order = gio.spatial_sort(arrs, face_centers, return_order=True)
            self._gaa_order = torch.as_tensor(np.asarray(order), dtype=torch.long)   # row i of the model = row _gaa_order[i] of the file (the permutation is kept)
        self.load_arrays(arrs, device=device)

    def save_ply(self, path):
        """scene/gaussian_model.py:253-275"""
        from . import io as gio

        arrs = {k: getattr(self, k).detach().cpu().numpy() for k in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation")}
        if self.binding is not None:
            arrs["binding"] = self.binding.detach().cpu().numpy()
        gio.save_ply(str(path), arrs)

    # ---- accessors (scene/gaussian_model.py:113-160, composed torch; rebound to the fused kernels below) ----------
    @property
    def get_scaling(self):
        if self.binding is None:
            return torch.exp(self._scaling)
        if self.face_scaling is None:
            self.select_mesh_by_timestep(0)
        return unfused.bind_scaling(self._scaling, self.binding, self.face_scaling)

    @property
    def get_rotation(self):
        if self.binding is None:
            return torch.nn.functional.normalize(self._rotation)
        if self.face_orien_quat is None:
            self.select_mesh_by_timestep(0)
        return unfused.bind_rotation(self._rotation, self.binding, self.face_orien_quat)

    @property
    def get_xyz(self):
        if self.binding is None:
            return self._xyz
        if self.face_center is None:
            self.select_mesh_by_timestep(0)
        return unfused.bind_xyz(self._xyz, self.binding, self.face_orien_mat, self.face_scaling, self.face_center)

    @property
    def get_features(self):
        return torch.cat((self._features_dc, self._features_rest), dim=1)

    @property
    def get_opacity(self):
        return torch.sigmoid(self._opacity)

    def get_covariance(self, scaling_modifier=1):
        # python cov3D path (pipe.compute_cov3D_python, default off): like the reference (:162-163) it
        # uses the LOCAL rotation, which is only right for un-bound splats.
        s = scaling_modifier * self.get_scaling
        q = self._rotation / self._rotation.norm(dim=1, keepdim=True)
        r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
        R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                         2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                         2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1).view(-1, 3, 3)
        L = R * s[:, None, :]
        S = L @ L.transpose(1, 2)
        return torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], 1)

    def select_mesh_by_timestep(self, timestep):
        raise NotImplementedError

    def update_densification_stats(self, viewspace_point_tensor, radii):
        """The two statistics lines of a training iteration, train.py:197-198, as one launch (include/gls.h):
            self.max_radii2D[vis] = max(self.max_radii2D[vis], radii[vis])          # train.py:197
            self.add_densification_stats(viewspace_point_tensor, vis)              # gaussian_model.py:517-519
        with vis = radii > 0 (render()'s visibility_filter)."""
        from .loss import densification_stats

        if viewspace_point_tensor.grad is None:
            raise RuntimeError("viewspace_point_tensor has no .grad: call backward() first")
        densification_stats(radii, viewspace_point_tensor.grad, self.max_radii2D, self.xyz_gradient_accum, self.denom)

    def oneupSHdegree(self):
        if self.active_sh_degree < self.max_sh_degree:
            self.active_sh_degree += 1


class FlameHead(nn.Module):
    """Buffers + forward of the reference FlameHead (flame_model/flame.py:83-184, 485-558) on a rig given
    as arrays (the licensed flame2023.pkl is not available; gaussianavatars_amd.synthetic.flame_rig
    produces the same schema)."""

    def __init__(self, rig: Dict[str, np.ndarray], impl: str = "fused"):
        super().__init__()
        for k in ("v_template", "shapedirs", "posedirs", "J_regressor", "lbs_weights"):
            self.register_buffer(k, torch.as_tensor(rig[k], dtype=torch.float32).contiguous())
        self.register_buffer("parents", torch.as_tensor(rig["parents"], dtype=torch.long))
        self.register_buffer("faces", torch.as_tensor(rig["faces"], dtype=torch.long), persistent=False)
        self.n_shape_params = 300
        self.n_expr_params = self.shapedirs.shape[2] - 300
        self.impl = impl

    def forward(self, shape, expr, rotation, neck, jaw, eyes, translation, zero_centered_at_root_node=False,
                return_landmarks=True, return_verts_cano=False, static_offset=None, dynamic_offset=None):
        if zero_centered_at_root_node or return_landmarks:
            raise NotImplementedError("only the per-frame path of GaussianAvatars is mirrored "
                                      "(zero_centered_at_root_node=False, return_landmarks=False)")
        rig = dict(v_template=self.v_template, shapedirs=self.shapedirs, posedirs=self.posedirs,
                   J_regressor=self.J_regressor, lbs_weights=self.lbs_weights, parents=self.parents)
        verts, v_shaped = unfused.flame_forward(rig, shape, expr, rotation, neck, jaw, eyes, translation, static_offset)
        return [verts, v_shaped] if return_verts_cano else verts


class FlameGaussianModel(GaussianModel):
    """Mesh-update surface of scene/flame_gaussian_model.py:22-41,91-154."""

    def __init__(self, sh_degree: int, rig: Dict[str, np.ndarray], disable_flame_static_offset=False,
                 not_finetune_flame_params=False, n_shape=300, n_expr=100, binding_impl: str = "fused", device="cuda"):
        super().__init__(sh_degree, binding_impl=binding_impl)
        self.disable_flame_static_offset = disable_flame_static_offset
        self.not_finetune_flame_params = not_finetune_flame_params
        self.n_shape, self.n_expr = n_shape, n_expr
        self.flame_model = FlameHead(rig, impl=binding_impl).to(device)
        self.flame_param = None
        self.flame_param_orig = None
        if self.binding is None:
            nf = len(self.flame_model.faces)
            self.binding = torch.arange(nf, device=device)
            self.binding_counter = torch.ones(nf, dtype=torch.int32, device=device)

    def load_flame_param(self, arrs: Dict[str, np.ndarray], device="cuda", requires_grad: bool = False):
        """flame_param.npz schema (scene/flame_gaussian_model.py:61-71,229-237)."""
        fp = {k: torch.as_tensor(v, dtype=torch.float32, device=device) for k, v in arrs.items()}
        if requires_grad:  # the rows train.py puts in Adam groups (:186-207)
            for k in ("rotation", "neck_pose", "jaw_pose", "eyes_pose", "translation", "expr"):
                fp[k].requires_grad_(True)
        self.flame_param = fp
        self.num_timesteps = fp["expr"].shape[0]

    def load_ply(self, path, device="cuda", spatial_sort: Optional[bool] = None, **kwargs):
        """scene/flame_gaussian_model.py:229-237: the PLY plus the flame_param.npz stored next to it.  `spatial_sort`: as GaussianModel.load_ply
        (default on), by the template centre of each splat's face."""
        import os

        from . import io as gio

        if spatial_sort is None:
            spatial_sort = spatial_order_default()
        super().load_ply(path, device=device, spatial_sort=spatial_sort, face_centers=template_face_centers(self) if spatial_sort else None)
        if not kwargs.get("has_target", False):
            self.load_flame_param(gio.load_flame_param(os.path.join(os.path.dirname(str(path)), "flame_param.npz")), device=device)

    def save_ply(self, path):
        """scene/flame_gaussian_model.py:219-224"""
        import os

        from . import io as gio

        super().save_ply(path)
        gio.save_flame_param(os.path.join(os.path.dirname(str(path)), "flame_param.npz"),
                             {k: v.detach().cpu().numpy() for k, v in self.flame_param.items()})

    def update_mesh_by_param_dict(self, flame_param):
        shape = flame_param["shape"] if "shape" in flame_param else self.flame_param["shape"]
        static_offset = flame_param["static_offset"] if "static_offset" in flame_param else self.flame_param["static_offset"]
        dev = self.flame_model.v_template.device
        verts, verts_cano = self.flame_model(
            shape[None, ...], flame_param["expr"].to(dev), flame_param["rotation"].to(dev), flame_param["neck"].to(dev),
            flame_param["jaw"].to(dev), flame_param["eyes"].to(dev), flame_param["translation"].to(dev),
            zero_centered_at_root_node=False, return_landmarks=False, return_verts_cano=True, static_offset=static_offset)
        self.update_mesh_properties(verts, verts_cano)

    def select_mesh_by_timestep(self, timestep, original=False):
        self.timestep = timestep
        fp = self.flame_param_orig if original and self.flame_param_orig is not None else self.flame_param
        verts, verts_cano = self.flame_model(
            fp["shape"][None, ...], fp["expr"][[timestep]], fp["rotation"][[timestep]], fp["neck_pose"][[timestep]],
            fp["jaw_pose"][[timestep]], fp["eyes_pose"][[timestep]], fp["translation"][[timestep]],
            zero_centered_at_root_node=False, return_landmarks=False, return_verts_cano=True,
            static_offset=fp["static_offset"], dynamic_offset=fp["dynamic_offset"][[timestep]])
        self.update_mesh_properties(verts, verts_cano)

    def update_mesh_properties(self, verts, verts_cano):
        faces = self.flame_model.faces
        c, R, s, q = unfused.face_frames(verts.squeeze(0), faces)
        self.face_center, self.face_orien_mat, self.face_scaling, self.face_orien_quat = c, R, s, q
        self.verts = verts
        self.faces = faces
        self.verts_cano = verts_cano


# The same rebinding `patch_reference()` applies to the reference's classes (gaussianavatars_amd/patch.py).
from .patch import patch_classes as _patch_classes  # noqa: E402

_patch_classes(GaussianModel, FlameGaussianModel, FlameHead)


def spatial_order_default() -> bool:
    """Whether loaders (and the densification hook of patch.py) keep the splats in Morton order of their positions: yes unless
    GAA_SPATIAL_SORT=0.  A layout choice -- the rasterizer's outputs do not depend on the order of the splats -- that decides how far apart in
    time the binning pass writes the entries of one tile (492 -> 141 MB written per frame at 2 M splats)."""
    import os

    return os.environ.get("GAA_SPATIAL_SORT", "1") != "0"


def template_face_centers(model):
    """(F,3) numpy: centres of the faces of the model's template mesh (flame_model.v_template / .faces), None for a model without a mesh."""
    fm = getattr(model, "flame_model", None)
    if fm is None or not hasattr(fm, "v_template") or not hasattr(fm, "faces"):
        return None
    v = fm.v_template.detach().cpu().numpy().reshape(-1, 3)
    f = fm.faces.detach().cpu().numpy().astype(np.int64).reshape(-1, 3)
    return v[f].mean(1)


def spatial_resort(model) -> torch.Tensor:
    """Puts the splats of a LIVE model -- this package's classes or the reference's own -- into Morton order of their positions
    (io.spatial_sort) and returns the permutation.  Densification appends its new splats (scene/gaussian_model.py:426-515), so a model that
    was loaded in order drifts out of it; this is the re-sort.  Everything that is indexed by splat moves together: the six leaf
    parameters, with an optimiser attached their Adam moments too -- through the reference's own `_prune_optimizer` (:349-371), which indexes
    parameters and moments with whatever it is given: a permutation instead of a keep-mask --, the densification statistics
    (xyz_gradient_accum, denom, max_radii2D) and the binding.  Recorded steps (graphs.py) must be captured again afterwards."""
    from . import io as gio

    with torch.no_grad():
        binding = getattr(model, "binding", None)
        arrs = {"_xyz": model._xyz.detach().cpu().numpy(), "binding": None if binding is None else binding.detach().cpu().numpy()}
        centers = template_face_centers(model) if binding is not None else None
        pos = arrs["_xyz"] if centers is None else centers[arrs["binding"].astype(np.int64)] + 1e-3 * arrs["_xyz"]
        perm = torch.as_tensor(gio.morton_order(pos), dtype=torch.long, device=model._xyz.device)
        names = {"xyz": "_xyz", "f_dc": "_features_dc", "f_rest": "_features_rest", "opacity": "_opacity", "scaling": "_scaling", "rotation": "_rotation"}
        if getattr(model, "optimizer", None) is not None and hasattr(model, "_prune_optimizer"):
            moved = model._prune_optimizer(perm)
            for group, attr in names.items():
                setattr(model, attr, moved[group])
        else:
            for attr in names.values():
                p = getattr(model, attr)
                setattr(model, attr, nn.Parameter(p.detach()[perm].contiguous().requires_grad_(p.requires_grad)))
        n = perm.shape[0]
        for aux in ("xyz_gradient_accum", "denom", "max_radii2D"):
            t = getattr(model, aux, None)
            if isinstance(t, torch.Tensor) and t.dim() >= 1 and t.shape[0] == n:
                setattr(model, aux, t[perm])
        if binding is not None:
            model.binding = binding[perm]
        # the order is recorded: row i of the model is row `_gaa_order[i]` of what was loaded (composed over every re-sort since; densification's new
        # splats get -1), so that tools matching splats by index across PLYs / checkpoints can undo it
        # (`inv = torch.empty_like(o); inv[o] = arange`: row j of the file is row inv[j] of the model)
        prev = getattr(model, "_gaa_order", None)
        if isinstance(prev, torch.Tensor) and prev.shape[0] == n:
            model._gaa_order = prev.to(perm.device)[perm]
        elif prev is None and not getattr(model, "_gaa_order_lost", False):
            model._gaa_order = perm.clone()          # first sort of a freshly loaded model: the file's order is the order it had until now
        else:
            # the row count changed without the rows being followed (patch._hook_spatial_order follows the reference's prune_points /
            # densification_postfix; anything else does not): no permutation is better than one that names the wrong rows
            model._gaa_order = None
            model._gaa_order_lost = True
    return perm
