"""The binding half written as composed torch ops -- i.e. what the reference executes as ~200 small
ATen launches per frame -- restated from its semantics:

  FLAME forward ........ flame_model/flame.py:485-558, flame_model/lbs.py:25-304
  face frames .......... utils/graphics_utils.py:90-135, scene/flame_gaussian_model.py:137-154
  roma stand-ins ....... rotmat_to_unitquat / quat_product (roma is not installed)
  splat local->world ... scene/gaussian_model.py:113-150

It is the plain-PyTorch fp32 reference the fused HIP binding kernels are tested against on the GPU
(same device, same dtype), and the `--binding unfused` leg of bench.py.  The default product path
is gaussianavatars_amd/binding.py (fused kernels); nothing here is a fallback for it.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

# ---- quaternion helpers (XYZW like roma unless stated) ---------------------------------------


def quat_xyzw_to_wxyz(q):
    return torch.roll(q, 1, dims=-1)


def quat_wxyz_to_xyzw(q):
    return torch.roll(q, -1, dims=-1)


def quat_product(p, q):
    """Hamilton product, XYZW."""
    pv, pw = p[..., :3], p[..., 3:]
    qv, qw = q[..., :3], q[..., 3:]
    v = pw * qv + qw * pv + torch.cross(pv, qv, dim=-1)
    w = pw * qw - (pv * qv).sum(-1, keepdim=True)
    return torch.cat([v, w], -1)


def rotmat_to_unitquat(R):
    """(...,3,3) -> (...,4) XYZW.  Branch on argmax(R00, R11, R22, trace) (SciPy's from_matrix
    algorithm, which roma follows), then normalise.  Differentiable w.r.t. R."""
    shp = R.shape[:-2]
    R = R.reshape(-1, 3, 3)
    dec = torch.stack([R[:, 0, 0], R[:, 1, 1], R[:, 2, 2], R[:, 0, 0] + R[:, 1, 1] + R[:, 2, 2]], 1)
    choice = dec.argmax(1)
    cands = []
    for i in range(3):
        j, k = (i + 1) % 3, (i + 2) % 3
        q = [None] * 4
        q[i] = 1 - dec[:, 3] + 2 * R[:, i, i]
        q[j] = R[:, j, i] + R[:, i, j]
        q[k] = R[:, k, i] + R[:, i, k]
        q[3] = R[:, k, j] - R[:, j, k]
        cands.append(torch.stack(q, 1))
    cands.append(torch.stack([R[:, 2, 1] - R[:, 1, 2], R[:, 0, 2] - R[:, 2, 0], R[:, 1, 0] - R[:, 0, 1], 1 + dec[:, 3]], 1))
    allq = torch.stack(cands, 1)  # (N,4 choices,4)
    q = allq[torch.arange(R.shape[0], device=R.device), choice]
    q = q / q.norm(dim=1, keepdim=True)
    return q.reshape(*shp, 4)


# ---- FLAME / LBS -----------------------------------------------------------------------------


def rodrigues(rot_vecs):
    """axis-angle (N,3) -> (N,3,3); the epsilon is added to the vector before the norm."""
    angle = torch.norm(rot_vecs + 1e-8, dim=1, keepdim=True)
    d = rot_vecs / angle
    c, s = torch.cos(angle)[:, None], torch.sin(angle)[:, None]
    z = torch.zeros_like(d[:, 0])
    K = torch.stack([z, -d[:, 2], d[:, 1], d[:, 2], z, -d[:, 0], -d[:, 1], d[:, 0], z], 1).view(-1, 3, 3)
    eye = torch.eye(3, dtype=rot_vecs.dtype, device=rot_vecs.device)[None]
    return eye + s * K + (1 - c) * torch.bmm(K, K)


def flame_forward(rig, shape, expr, rotation, neck, jaw, eyes, translation, static_offset=None):
    """rig: dict of buffers (v_template (V,3), shapedirs (V,3,400), posedirs (36,3V), J_regressor (5,V),
    lbs_weights (V,5), parents).  Batch-1 inputs shaped like the reference's.  -> (verts, v_shaped)."""
    B = shape.shape[0]
    betas = torch.cat([shape, expr], 1)
    pose = torch.cat([rotation, neck, jaw, eyes], 1)
    v_shaped = rig["v_template"][None] + torch.einsum("bl,mkl->bmk", betas, rig["shapedirs"])
    if static_offset is not None:
        v_shaped = v_shaped + static_offset
    J = torch.einsum("bik,ji->bjk", v_shaped, rig["J_regressor"])
    nj = J.shape[1]
    R = rodrigues(pose.reshape(-1, 3)).view(B, nj, 3, 3)
    eye = torch.eye(3, dtype=R.dtype, device=R.device)
    pose_feature = (R[:, 1:] - eye).reshape(B, -1)
    v_posed = v_shaped + (pose_feature @ rig["posedirs"]).view(B, -1, 3)
    parents = rig["parents"]
    rel = J.clone()
    rel[:, 1:] = J[:, 1:] - J[:, parents[1:]]
    Tm = torch.cat([F.pad(R, [0, 0, 0, 1]), F.pad(rel[..., None], [0, 0, 0, 1], value=1.0)], -1)  # (B,nj,4,4)
    chain = [Tm[:, 0]]
    for i in range(1, nj):
        chain.append(chain[int(parents[i])] @ Tm[:, i])
    G = torch.stack(chain, 1)
    Jh = F.pad(J[..., None], [0, 0, 0, 1])
    A = G - F.pad(G @ Jh, [3, 0])
    T = (rig["lbs_weights"][None] @ A.view(B, nj, 16)).view(B, -1, 4, 4)
    vh = F.pad(v_posed, [0, 1], value=1.0)
    verts = (T @ vh[..., None])[:, :, :3, 0]
    return verts + translation[:, None, :], v_shaped


# ---- per-face frames -------------------------------------------------------------------------


def _safe_len(x, eps=1e-20):
    return torch.sqrt(torch.clamp((x * x).sum(-1, keepdim=True), min=eps))


def face_frames(verts, faces):
    """verts (V,3), faces (F,3) -> centre (F,3), R (F,3,3) with columns a0 a1 a2, scale (F,1), quat WXYZ (F,4)."""
    v0, v1, v2 = verts[faces[:, 0]], verts[faces[:, 1]], verts[faces[:, 2]]
    center = torch.stack([v0, v1, v2], 1).mean(1)
    e1, e2 = v1 - v0, v2 - v0
    a0 = e1 / _safe_len(e1)
    n = torch.cross(a0, e2, dim=-1)
    a1 = n / _safe_len(n)
    m = torch.cross(a1, a0, dim=-1)
    a2 = -(m / _safe_len(m))
    R = torch.stack([a0, a1, a2], -1)
    scale = (_safe_len(e1) + (a2 * e2).sum(-1, keepdim=True).abs()) / 2
    quat = quat_xyzw_to_wxyz(rotmat_to_unitquat(R))
    return center, R, scale, quat


# ---- per-splat local -> world ----------------------------------------------------------------


def bind_xyz(xyz, binding, face_R, face_scale, face_center):
    b = binding.long()
    return torch.bmm(face_R[b], xyz[..., None]).squeeze(-1) * face_scale[b] + face_center[b]


def bind_scaling(log_scaling, binding, face_scale):
    return torch.exp(log_scaling) * face_scale[binding.long()]


def bind_rotation(rotation, binding, face_quat):
    rot = F.normalize(rotation)
    fq = F.normalize(face_quat[binding.long()])
    return quat_xyzw_to_wxyz(quat_product(quat_wxyz_to_xyzw(fq), quat_wxyz_to_xyzw(rot)))
