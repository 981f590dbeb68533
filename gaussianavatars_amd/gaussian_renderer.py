"""Mirror of the reference's render boundary, gaussian_renderer/__init__.py:19-101: same signature,
same settings construction, same return dict.  (With the repository root on PYTHONPATH the
reference's own file runs unchanged against the top-level `diff_gaussian_rasterization` package;
this mirror exists because /root/reference does not travel to the GPU box.)"""
from __future__ import annotations

import math

import torch

from . import patch as _patch   # (module level: a function-level `from . import x` costs ~1.5 us per call on a step whose host side is counted in microseconds)
from .loss import l1_loss as _l1_loss
from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer, rasterize_bound, rasterize_leaves

C0 = 0.28209479177387814


_C1 = 0.4886025119029199
_C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)
_C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
       1.445305721320277, -0.5900435899266435)


def eval_sh(deg: int, sh: torch.Tensor, dirs: torch.Tensor) -> torch.Tensor:
    """Real SH basis up to degree 3 (the polynomial of utils/sh_utils.py:57-112) for coefficients laid
    out (P, M, 3) and unit directions (P, 3) -> (P, 3)."""
    x, y, z = dirs[:, 0:1], dirs[:, 1:2], dirs[:, 2:3]
    res = C0 * sh[:, 0]
    if deg > 0:
        res = res - _C1 * y * sh[:, 1] + _C1 * z * sh[:, 2] - _C1 * x * sh[:, 3]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        res = (res + _C2[0] * xy * sh[:, 4] + _C2[1] * yz * sh[:, 5] + _C2[2] * (2.0 * zz - xx - yy) * sh[:, 6]
               + _C2[3] * xz * sh[:, 7] + _C2[4] * (xx - yy) * sh[:, 8])
    if deg > 2:
        res = (res + _C3[0] * y * (3 * xx - yy) * sh[:, 9] + _C3[1] * xy * z * sh[:, 10]
               + _C3[2] * y * (4 * zz - xx - yy) * sh[:, 11] + _C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[:, 12]
               + _C3[4] * x * (4 * zz - xx - yy) * sh[:, 13] + _C3[5] * z * (xx - yy) * sh[:, 14]
               + _C3[6] * x * (xx - 3 * yy) * sh[:, 15])
    return res


_ZERO_POINTS: dict = {}


def _screenspace_leaf(xyz: torch.Tensor) -> torch.Tensor:
    """The dummy (P,3) tensor whose .grad receives dL/d(screen-space mean) (reference :27-31).  The reference builds
    `zeros_like(xyz) + 0` and calls retain_grad() every frame (a fill, an add and a gradient copy); the rasterizer never
    reads the values, so every frame gets a fresh LEAF over one cached block of zeros: autograd adopts the rasterizer's
    gradient buffer as its .grad without a copy."""
    key = (xyz.device, tuple(xyz.shape), xyz.dtype)
    z = _ZERO_POINTS.get(key)
    if z is None:
        if len(_ZERO_POINTS) > 8:
            _ZERO_POINTS.clear()
        z = _ZERO_POINTS[key] = torch.zeros(xyz.shape, dtype=xyz.dtype, device=xyz.device)
    return z.detach().requires_grad_(True)


def _dev(t, device):
    """`t` on `device`.  Host tensors (the DataLoader cameras of train.py: transposed VIEWS, scene/cameras.py:44-46) are made contiguous BEFORE the upload, so that
    the rasterizer's entry takes the uploaded tensor as it is -- a strided device tensor would be copied once more there and kept in a cache it can never hit."""
    if isinstance(t, torch.Tensor):
        return t if t.device == device else t.contiguous().to(device)
    return torch.as_tensor(t, device=device)


def _bound_fast_path(pc, pipe, override_color) -> bool:
    """A patched mesh-bound model rendered the default way: the rasterizer's bound entry takes the model's leaves and per-face frames
    and evaluates get_xyz / get_scaling / get_rotation / get_opacity inside its first kernel -- one autograd
    node instead of two, no world-space tensors.  Anything else (python SH / covariance paths, colour overrides, unbound or
    "unfused" models, `pc.bound_render = False`) takes the reference-shaped path below."""
    if override_color is not None or pipe.compute_cov3D_python or pipe.convert_SHs_python:
        return False
    if getattr(pc, "binding_impl", _patch._default_impl()) == "unfused" or not getattr(pc, "bound_render", True):
        return False
    if not getattr(type(pc), "_gaa_patched", False) or getattr(pc, "get_features_split", None) is None:
        return False
    return pc._xyz.is_cuda   # bound: the bound entry; unbound: the same entry without faces (the three activations in-kernel)


def _render_bound(viewpoint_camera, pc, pipe, bg_color, scaling_modifier):
    unbound = getattr(pc, "binding", None) is None
    if not unbound and pc.face_center is None:          # same lazy initialisation as the reference's accessors (scene/gaussian_model.py:119-120)
        pc.select_mesh_by_timestep(0)
    device = pc._xyz.device
    screenspace_points = _screenspace_leaf(pc._xyz)
    raster_settings = GaussianRasterizationSettings(
        image_height=int(viewpoint_camera.image_height), image_width=int(viewpoint_camera.image_width),
        tanfovx=math.tan(viewpoint_camera.FoVx * 0.5), tanfovy=math.tan(viewpoint_camera.FoVy * 0.5), bg=bg_color,
        scale_modifier=scaling_modifier, viewmatrix=_dev(viewpoint_camera.world_view_transform, device),
        projmatrix=_dev(viewpoint_camera.full_proj_transform, device), sh_degree=pc.active_sh_degree,
        campos=_dev(viewpoint_camera.camera_center, device), prefiltered=False, debug=pipe.debug)
    dc, rest = pc.get_features_split
    if unbound:
        image, radii, visible = rasterize_leaves(pc._xyz, screenspace_points, dc, rest, pc._opacity, pc._scaling, pc._rotation, raster_settings)
        return {"render": image, "viewspace_points": screenspace_points, "visibility_filter": visible, "radii": radii}
    csr = _patch.binding_csr_cached(pc, pc.face_center.shape[0])
    image, radii, visible = rasterize_bound(pc._xyz, screenspace_points, dc, rest, pc._opacity, pc._scaling, pc._rotation, pc.face_orien_mat,
                                            pc.face_scaling, pc.face_center, pc.face_orien_quat, pc.binding, csr, raster_settings)
    return {"render": image, "viewspace_points": screenspace_points, "visibility_filter": visible, "radii": radii}


def render(viewpoint_camera, pc, pipe, bg_color: torch.Tensor, scaling_modifier=1.0, override_color=None):
    if _bound_fast_path(pc, pipe, override_color):
        return _render_bound(viewpoint_camera, pc, pipe, bg_color, scaling_modifier)
    xyz = pc.get_xyz
    device = xyz.device
    screenspace_points = _screenspace_leaf(xyz)
    tanfovx = math.tan(viewpoint_camera.FoVx * 0.5)
    tanfovy = math.tan(viewpoint_camera.FoVy * 0.5)
    raster_settings = GaussianRasterizationSettings(
        image_height=int(viewpoint_camera.image_height),
        image_width=int(viewpoint_camera.image_width),
        tanfovx=tanfovx,
        tanfovy=tanfovy,
        bg=bg_color,
        scale_modifier=scaling_modifier,
        viewmatrix=_dev(viewpoint_camera.world_view_transform, device),
        projmatrix=_dev(viewpoint_camera.full_proj_transform, device),
        sh_degree=pc.active_sh_degree,
        campos=_dev(viewpoint_camera.camera_center, device),
        prefiltered=False,
        debug=pipe.debug,
    )
    rasterizer = GaussianRasterizer(raster_settings=raster_settings)
    means3D = xyz
    means2D = screenspace_points
    opacity = pc.get_opacity
    scales = rotations = cov3D_precomp = None
    if pipe.compute_cov3D_python:
        cov3D_precomp = pc.get_covariance(scaling_modifier)
    else:
        scales = pc.get_scaling
        rotations = pc.get_rotation
    shs = shs_rest = colors_precomp = None
    if override_color is None:
        if pipe.convert_SHs_python:
            # python SH path of the reference (:74-79): colours from composed torch ops
            feats = pc.get_features  # (P, M, 3)
            d = xyz - raster_settings.campos[None]
            d = d / d.norm(dim=1, keepdim=True)
            colors_precomp = torch.clamp_min(eval_sh(pc.active_sh_degree, feats, d) + 0.5, 0.0)
        else:
            # the reference passes pc.get_features (a per-frame cat of the two leaf tensors); when the model offers
            # them separately the rasterizer reads both in place
            split = getattr(pc, "get_features_split", None)
            if split is not None:
                shs, shs_rest = split
            else:
                shs = pc.get_features
    else:
        colors_precomp = override_color
    rendered_image, radii = rasterizer(means3D=means3D, means2D=means2D, shs=shs, colors_precomp=colors_precomp,
                                       opacities=opacity, scales=scales, rotations=rotations, cov3D_precomp=cov3D_precomp,
                                       **({"shs_rest": shs_rest} if shs_rest is not None else {}))
    visible = getattr(rasterizer, "visibility_filter", None)   # our rasterizer hands out radii > 0 from its forward kernel
    if visible is None:
        visible = radii > 0
    return {"render": rendered_image, "viewspace_points": screenspace_points, "visibility_filter": visible, "radii": radii}


def l1_loss(network_output, gt):
    """utils/loss_utils.py:17-18, on the fused kernel of include/gls.h (see loss.py)."""
    return _l1_loss(network_output, gt)
