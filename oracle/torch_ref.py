"""Independent fp64, autograd-differentiable restatement of SURVEY.md Appendix A.1 + A.3.

TEST INFRASTRUCTURE ONLY.  Purpose: pin the *analytic* backward of oracle/gsr_oracle.c (A.4) and
of the HIP kernels against torch autograd + finite differences, without sharing a line of
derivative code with them.  The combinatorial structure (sorted per-tile lists, tile ranges and
the per-pixel early-termination index n_contrib) is taken as given -- no gradient flows through
it in the reference either (Appendix A.4, last bullet).
"""
from __future__ import annotations

import torch

C0 = 0.28209479177387814
C1 = 0.4886025119029199
C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
      1.445305721320277, -0.5900435899266435]


def _sh_rgb(deg, sh, dirs):
    """sh: (P,M,3), dirs: (P,3) unit.  Same polynomial as utils/sh_utils.py:57-112 but written
    for the (P,M,3) layout the rasterizer receives."""
    x, y, z = dirs[:, 0:1], dirs[:, 1:2], dirs[:, 2:3]
    res = C0 * sh[:, 0]
    if deg > 0:
        res = res - C1 * y * sh[:, 1] + C1 * z * sh[:, 2] - C1 * x * sh[:, 3]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        res = (res + C2[0] * xy * sh[:, 4] + C2[1] * yz * sh[:, 5] + C2[2] * (2 * zz - xx - yy) * sh[:, 6]
               + C2[3] * xz * sh[:, 7] + C2[4] * (xx - yy) * sh[:, 8])
    if deg > 2:
        res = (res + C3[0] * y * (3 * xx - yy) * sh[:, 9] + C3[1] * xy * z * sh[:, 10]
               + C3[2] * y * (4 * zz - xx - yy) * sh[:, 11] + C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[:, 12]
               + C3[4] * x * (4 * zz - xx - yy) * sh[:, 13] + C3[5] * z * (xx - yy) * sh[:, 14]
               + C3[6] * x * (xx - 3 * yy) * sh[:, 15])
    return res


def render(H, W, tanfovx, tanfovy, bg, scale_modifier, viewmatrix, projmatrix, sh_degree, campos,
           means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
           radii, ranges, point_list, n_contrib):
    """All float tensors fp64.  viewmatrix/projmatrix are the transposed-storage tensors the
    rasterizer receives.  radii/ranges/point_list/n_contrib: the oracle's integer outputs."""
    dt = torch.float64
    Vm = viewmatrix.to(dt).T      # W2C
    PV = projmatrix.to(dt).T      # P @ W2C
    P = means3D.shape[0]
    ones = torch.ones(P, 1, dtype=dt)
    ph = torch.cat([means3D, ones], 1)
    p_view = ph @ Vm.T
    p_hom = ph @ PV.T
    p_w = 1.0 / (p_hom[:, 3:4] + 1e-7)
    p_proj = p_hom[:, :3] * p_w
    ndc = p_proj[:, :2] + means2D[:, :2]
    px = ((ndc[:, 0] + 1.0) * W - 1.0) * 0.5
    py = ((ndc[:, 1] + 1.0) * H - 1.0) * 0.5

    if cov3D_precomp is None:
        q = rotations
        r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
        R = torch.stack([
            1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
            2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
            2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1).reshape(P, 3, 3)
        S = torch.diag_embed(scale_modifier * scales)
        L = R @ S
        Sigma = L @ L.transpose(1, 2)
    else:
        c = cov3D_precomp
        Sigma = torch.stack([c[:, 0], c[:, 1], c[:, 2], c[:, 1], c[:, 3], c[:, 4], c[:, 2], c[:, 4], c[:, 5]], 1).reshape(P, 3, 3)

    fx = W / (2.0 * tanfovx)
    fy = H / (2.0 * tanfovy)
    tz = p_view[:, 2]
    limx, limy = 1.3 * tanfovx, 1.3 * tanfovy
    # Outside the 1.3x frustum the reference treats the clamped t.x / t.y as constants in the
    # backward (x_grad_mul / y_grad_mul, A.4 K8), so they are detached here.
    rx, ry = p_view[:, 0] / tz, p_view[:, 1] / tz
    tx = torch.where((rx < -limx) | (rx > limx), (torch.clamp(rx, -limx, limx) * tz).detach(), p_view[:, 0])
    ty = torch.where((ry < -limy) | (ry > limy), (torch.clamp(ry, -limy, limy) * tz).detach(), p_view[:, 1])
    zeros = torch.zeros_like(tz)
    J = torch.stack([fx / tz, zeros, -fx * tx / (tz * tz), zeros, fy / tz, -fy * ty / (tz * tz)], 1).reshape(P, 2, 3)
    A = J @ Vm[:3, :3]
    cov2 = A @ Sigma @ A.transpose(1, 2)
    a = cov2[:, 0, 0] + 0.3
    b = cov2[:, 0, 1]
    c_ = cov2[:, 1, 1] + 0.3
    det = a * c_ - b * b
    det = torch.where(det == 0, torch.ones_like(det), det)
    conA, conB, conC = c_ / det, -b / det, a / det

    if colors_precomp is None:
        d = means3D - campos.to(dt)[None]
        d = d / d.norm(dim=1, keepdim=True)
        rgb = torch.clamp_min(_sh_rgb(sh_degree, shs, d) + 0.5, 0.0)
    else:
        rgb = colors_precomp

    op = opacities.reshape(-1)
    gx = (W + 15) // 16
    out = torch.zeros(3, H, W, dtype=dt)
    bgt = torch.as_tensor(bg, dtype=dt)
    for t in range(ranges.shape[0]):
        r0, r1 = int(ranges[t, 0]), int(ranges[t, 1])
        ty0, tx0 = (t // gx) * 16, (t % gx) * 16
        ys = torch.arange(ty0, min(ty0 + 16, H))
        xs = torch.arange(tx0, min(tx0 + 16, W))
        if len(ys) == 0 or len(xs) == 0:
            continue
        yy, xx = torch.meshgrid(ys, xs, indexing="ij")
        pixx, pixy = xx.reshape(-1).to(dt), yy.reshape(-1).to(dt)
        npx = pixx.shape[0]
        if r1 <= r0:
            out[:, yy.reshape(-1), xx.reshape(-1)] = bgt[:, None].expand(3, npx)
            continue
        ids = torch.as_tensor(point_list[r0:r1].astype("int64"))
        dx = px[ids][None, :] - pixx[:, None]
        dy = py[ids][None, :] - pixy[:, None]
        power = -0.5 * (conA[ids][None] * dx * dx + conC[ids][None] * dy * dy) - conB[ids][None] * dx * dy
        G = torch.exp(power)
        alpha_raw = op[ids][None] * G
        alpha = torch.clamp_max(alpha_raw, 0.99)
        # the 0.99 clamp is ignored in the reference gradient (A.4): straight-through
        alpha = alpha_raw + (alpha - alpha_raw).detach()
        nc = torch.as_tensor(n_contrib[yy.reshape(-1).numpy(), xx.reshape(-1).numpy()].astype("int64"))
        k = torch.arange(r1 - r0)[None, :]
        live = (power <= 0) & (alpha.detach() >= 1.0 / 255.0) & (k < nc[:, None])
        a_eff = torch.where(live, alpha, torch.zeros_like(alpha))
        Tcum = torch.cumprod(1 - a_eff, 1)
        Tprev = torch.cat([torch.ones(npx, 1, dtype=dt), Tcum[:, :-1]], 1)
        w = a_eff * Tprev
        col = w @ rgb[ids]
        Tfin = Tcum[:, -1]
        res = col + Tfin[:, None] * bgt[None]
        out[:, yy.reshape(-1), xx.reshape(-1)] = res.T
    return out
