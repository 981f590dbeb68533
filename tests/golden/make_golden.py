#!/usr/bin/env python3
"""Generates tests/golden/*.npz by IMPORTING THE REFERENCE (read-only, /root/reference) in this
container -- it cannot travel to the GPU box, so the vectors it produces are committed.

    python tests/golden/make_golden.py

Pins produced (the reference has no tests or golden vectors of its own):
  raster_pins.npz   utils/sh_utils.eval_sh                         -> SH colour of the oracle's K1
                    utils/general_utils.build_scaling_rotation/..  -> cov3D packing + quaternion->R
                    utils/graphics_utils.getWorld2View2/getProjectionMatrix + scene/cameras.py:44-47
                                                                   -> transposed-matrix convention, pixel coords
  binding_pins.npz  flame_model/lbs.py (lbs, blend_shapes)         -> FLAME forward on a small synthetic rig
                    utils/graphics_utils.compute_face_orientation  -> per-face frames
                    scipy Rotation.from_matrix (roma's algorithm)  -> rotmat -> quaternion, up to sign
  loss_pins.npz     utils/loss_utils.l1_loss / ssim (values + autograd gradients, fp32 torch-CPU)
                    scene/gaussian_model.GaussianModel.add_densification_stats + train.py:197
                                                                   -> densification statistics update
  model_pins.npz    the reference's OWN classes at full rig size (V = 5143, F = 10144, 400 betas, 20 000 bound splats):
                    scene/flame_gaussian_model.FlameGaussianModel.select_mesh_by_timestep / update_mesh_by_param_dict /
                    update_mesh_properties, flame_model/flame.FlameHead.forward, flame_model/lbs.lbs,
                    scene/gaussian_model.GaussianModel.get_xyz / get_scaling / get_rotation / get_opacity / get_covariance,
                    and torch autograd through all of them down to the flame_param rows
                    (inputs are regenerated from gaussianavatars_amd.synthetic seeds; only outputs are stored, the large
                    ones at a fixed stride).  roma is absent: its four functions come from gaussianavatars_amd/shims/roma.py,
                    which tests/test_binding_cpu.py pins to SciPy.
"""
import math
import os
import sys
from unittest import mock

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

from utils import sh_utils, graphics_utils, general_utils  # noqa: E402  (reference modules)
from flame_model import lbs as ref_lbs  # noqa: E402

from gaussianavatars_amd import synthetic as S  # noqa: E402


def cpu_zeros():
    orig = torch.zeros

    def z(*a, **k):
        k.pop("device", None)
        return orig(*a, **k)

    return mock.patch("torch.zeros", z)


def raster_pins():
    g = np.random.default_rng(101)
    N = 400
    # -- SH
    means = g.normal(0, 0.2, (N, 3)).astype(np.float32)
    campos = np.array([0.1, -0.2, 1.0], np.float32)
    shs = g.normal(0, 0.4, (N, 16, 3)).astype(np.float32)
    d = torch.tensor(means - campos)
    d = d / d.norm(dim=1, keepdim=True)
    rgb = {}
    for deg in range(4):
        sh_view = torch.tensor(shs).transpose(1, 2)  # (N,3,16) as gaussian_renderer/__init__.py:75
        out = sh_utils.eval_sh(deg, sh_view, d)
        rgb[deg] = torch.clamp_min(out + 0.5, 0.0).numpy()
    # -- cov3D (the reference normalises q; the rasterizer uses it raw -> feed unit quaternions)
    scales = np.exp(g.normal(-4, 0.5, (N, 3))).astype(np.float32)
    q = g.normal(0, 1, (N, 4))
    q = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32)
    mod = 1.3
    with cpu_zeros():
        L = general_utils.build_scaling_rotation(mod * torch.tensor(scales), torch.tensor(q))
        cov = general_utils.strip_symmetric(L @ L.transpose(1, 2)).numpy()
    # -- camera conventions (dataset-style camera, scene/cameras.py:44-47)
    th = 0.3
    Rm = np.array([[math.cos(th), 0, math.sin(th)], [0, 1, 0], [-math.sin(th), 0, math.cos(th)]])
    T = np.array([0.05, -0.02, 1.2])
    fovx, fovy = 0.5, 0.7
    W2Ct = torch.tensor(graphics_utils.getWorld2View2(Rm, T)).transpose(0, 1)
    Pt = graphics_utils.getProjectionMatrix(znear=0.01, zfar=100.0, fovX=fovx, fovY=fovy).transpose(0, 1)
    full = (W2Ct.unsqueeze(0).bmm(Pt.unsqueeze(0))).squeeze(0)
    center = W2Ct.inverse()[3, :3]
    pts = torch.tensor(g.normal(0, 0.15, (N, 3)).astype(np.float32))
    ph = torch.cat([pts, torch.ones(N, 1)], 1) @ full   # row-vector convention of the reference
    pv = torch.cat([pts, torch.ones(N, 1)], 1) @ W2Ct
    ndc = ph[:, :2] / (ph[:, 3:4] + 1e-7)
    Wd, Hd = 320, 240
    pix = torch.stack([((ndc[:, 0] + 1) * Wd - 1) * 0.5, ((ndc[:, 1] + 1) * Hd - 1) * 0.5], 1)
    np.savez_compressed(
        os.path.join(HERE, "raster_pins.npz"),
        sh_means=means, sh_campos=campos, sh_shs=shs, sh_rgb0=rgb[0], sh_rgb1=rgb[1], sh_rgb2=rgb[2], sh_rgb3=rgb[3],
        cov_scales=scales, cov_quat=q, cov_mod=np.float32(mod), cov_expected=cov,
        cam_viewmatrix=W2Ct.numpy(), cam_projmatrix=full.numpy(), cam_center=center.numpy(), cam_fovx=np.float32(fovx),
        cam_fovy=np.float32(fovy), cam_W=np.int32(Wd), cam_H=np.int32(Hd), cam_points=pts.numpy(), cam_pix=pix.numpy(),
        cam_depth=pv[:, 2].numpy(),
    )


def binding_pins():
    g = np.random.default_rng(202)
    V, Fn, J, NB = 180, 300, 5, 40
    rig = dict(
        v_template=g.normal(0, 0.1, (V, 3)).astype(np.float32),
        shapedirs=g.normal(0, 0.01, (V, 3, NB)).astype(np.float32),
        posedirs=g.normal(0, 0.01, (36, 3 * V)).astype(np.float32),
        J_regressor=np.abs(g.normal(0, 1, (J, V))).astype(np.float32),
        lbs_weights=np.abs(g.normal(0, 1, (V, J))).astype(np.float32),
        parents=np.array([-1, 0, 1, 1, 1], np.int64),
    )
    rig["J_regressor"] /= rig["J_regressor"].sum(1, keepdims=True)
    rig["lbs_weights"] /= rig["lbs_weights"].sum(1, keepdims=True)
    faces = np.stack([g.permutation(V)[:3] for _ in range(Fn)]).astype(np.int64)
    B = 1
    betas = g.normal(0, 1, (B, NB)).astype(np.float32)
    pose = g.normal(0, 0.3, (B, 15)).astype(np.float32)
    pose[0, 3:6] = 0.0   # one exactly-zero joint rotation: the 1e-8 epsilon path of batch_rodrigues
    trans = g.normal(0, 0.05, (B, 3)).astype(np.float32)
    static_offset = g.normal(0, 0.002, (1, V, 3)).astype(np.float32)
    t = torch.tensor
    # the reference's FlameHead.forward body (flame_model/flame.py:511-536), using its own lbs module
    v_shaped = t(rig["v_template"])[None] + ref_lbs.blend_shapes(t(betas), t(rig["shapedirs"]))
    v_shaped = v_shaped + t(static_offset)
    verts, Jt, _ = ref_lbs.lbs(t(pose), v_shaped, t(rig["posedirs"]), t(rig["J_regressor"]), t(rig["parents"]),
                               t(rig["lbs_weights"]), dtype=torch.float32)
    verts = verts + t(trans)[:, None, :]
    R, scale = graphics_utils.compute_face_orientation(verts[0], t(faces), return_scale=True)
    center = verts[:, t(faces)].mean(dim=-2).squeeze(0)
    from scipy.spatial.transform import Rotation

    quat_xyzw = Rotation.from_matrix(R.numpy().astype(np.float64)).as_quat().astype(np.float32)
    np.savez_compressed(
        os.path.join(HERE, "binding_pins.npz"),
        faces=faces, betas=betas, pose=pose, trans=trans, static_offset=static_offset,
        verts=verts.numpy(), v_shaped=v_shaped.numpy(), face_R=R.numpy(), face_scale=scale.numpy(), face_center=center.numpy(),
        face_quat_xyzw_scipy=quat_xyzw, **{"rig_" + k: v for k, v in rig.items()},
    )


def loss_pins():
    from utils import loss_utils  # reference module

    g = np.random.default_rng(303)
    out = {}
    # smooth-ish images in [0,1] (a flat patch, an edge and noise) so every branch of the map matters
    def image(shape):
        img = g.uniform(0, 1, shape).astype(np.float32)
        img[..., : shape[-2] // 3, :] = 0.5 + 0.01 * g.normal(size=img[..., : shape[-2] // 3, :].shape)
        return np.clip(img, 0, 1).astype(np.float32)

    cases = {"chw": (3, 45, 37), "bchw": (2, 3, 20, 33), "tiny": (1, 7, 5)}
    for name, shape in cases.items():
        a, b = image(shape), image(shape)
        b[..., ::4, ::3] = a[..., ::4, ::3]   # exact ties: d|x-y| = 0 there
        ta = torch.tensor(a, requires_grad=True)
        tb = torch.tensor(b)
        l1 = loss_utils.l1_loss(ta, tb)
        (g_l1,) = torch.autograd.grad(l1, ta)
        ss = loss_utils.ssim(ta, tb)
        (g_ss,) = torch.autograd.grad(ss, ta)
        out.update({f"{name}_a": a, f"{name}_b": b, f"{name}_l1": l1.detach().numpy(), f"{name}_ssim": ss.detach().numpy(),
                    f"{name}_g_l1": g_l1.numpy(), f"{name}_g_ssim": g_ss.numpy()})
        if len(shape) == 4:
            out[f"{name}_ssim_per_image"] = loss_utils.ssim(ta, tb, size_average=False).detach().numpy()
    out["window_2d"] = loss_utils.create_window(11, 1)[0, 0].numpy()

    # densification statistics: the reference method itself (its module needs three absent packages only at import)
    absent = ("plyfile", "simple_knn", "simple_knn._C", "roma", "iopath", "iopath.common", "iopath.common.file_io", "cv2",
              "PIL", "PIL.Image", "tyro", "dearpygui", "dearpygui.dearpygui", "lpips", "tensorboard", "matplotlib", "matplotlib.pyplot")
    with mock.patch.dict(sys.modules, {k: mock.MagicMock() for k in absent if k not in sys.modules}):
        from scene.gaussian_model import GaussianModel as RefGaussianModel
    P = 500
    radii = g.integers(-0, 40, P).astype(np.int32)
    radii[g.uniform(size=P) < 0.4] = 0
    vgrad = g.normal(0, 1e-3, (P, 3)).astype(np.float32)
    max_r = g.integers(0, 30, P).astype(np.float32)
    acc = np.abs(g.normal(0, 1e-3, (P, 1))).astype(np.float32)
    den = g.integers(0, 9, (P, 1)).astype(np.float32)
    m = RefGaussianModel.__new__(RefGaussianModel)
    m.xyz_gradient_accum, m.denom, m.max_radii2D = torch.tensor(acc), torch.tensor(den), torch.tensor(max_r)
    vis = torch.tensor(radii) > 0
    tr = torch.tensor(radii)
    m.max_radii2D[vis] = torch.max(m.max_radii2D[vis], tr[vis])  # train.py:197
    vp = torch.zeros(P, 3, requires_grad=True)
    vp.grad = torch.tensor(vgrad)
    m.add_densification_stats(vp, vis)                           # gaussian_model.py:517-519
    out.update(ds_radii=radii, ds_vgrad=vgrad, ds_max_in=max_r, ds_acc_in=acc, ds_den_in=den,
               ds_max_out=m.max_radii2D.numpy(), ds_acc_out=m.xyz_gradient_accum.numpy(), ds_den_out=m.denom.numpy())
    np.savez_compressed(os.path.join(HERE, "loss_pins.npz"), **out)


from tests.model_pin_inputs import MODEL_PINS, model_pin_inputs  # noqa: E402  (shared with the tests that consume the pins)


def model_pins():
    from gaussianavatars_amd import shims

    shims.install(stub_torchvision=True)   # roma / plyfile / simple_knn ...: absent third-party imports of scene/*.py
    from flame_model.flame import FlameHead as RefHead
    from scene.flame_gaussian_model import FlameGaussianModel as RefFGM
    from scene.gaussian_model import GaussianModel as RefGM

    assert not getattr(RefGM, "_gaa_patched", False), "pins must come from the UNPATCHED reference classes"
    c = MODEL_PINS
    rig, seq, sp, w, pd = model_pin_inputs()

    def build(dt):
        """The reference's FlameGaussianModel around the synthetic rig, in dtype dt."""
        t = lambda a: torch.tensor(a, dtype=dt) if np.asarray(a).dtype.kind == "f" else torch.tensor(a)
        # FlameHead without its asset files (flame2023.pkl is licence-gated): the buffers its __init__ would register
        head = RefHead.__new__(RefHead)
        torch.nn.Module.__init__(head)
        head.n_shape_params, head.n_expr_params, head.dtype = S.N_SHAPE, S.N_EXPR, dt
        for k in ("v_template", "shapedirs", "posedirs", "J_regressor", "lbs_weights"):
            head.register_buffer(k, t(rig[k]))
        head.register_buffer("parents", t(rig["parents"]))
        head.register_buffer("faces", t(rig["faces"]), persistent=False)
        m = RefFGM.__new__(RefFGM)
        RefGM.__init__(m, 3)                       # the reference constructor of the base class (activations, empty leaves)
        m.disable_flame_static_offset = m.not_finetune_flame_params = False
        m.n_shape, m.n_expr = S.N_SHAPE, S.N_EXPR
        m.flame_model = head
        m.flame_param = {k: t(v) for k, v in seq.items()}
        for k in ("expr", "rotation", "neck_pose", "jaw_pose", "eyes_pose", "translation"):
            m.flame_param[k].requires_grad_(True)
        m.flame_param_orig = None
        for k in ("_xyz", "_scaling", "_rotation", "_opacity", "_features_dc", "_features_rest"):
            setattr(m, k, torch.nn.Parameter(t(sp[k])))
        m.binding = t(sp["binding"])
        return m, t

    m, t = build(torch.float32)
    m64, t64 = build(torch.float64)   # the same reference code in double: the value the fp32 gradients of the FLAME rows are noisy around
    out = {}
    fs, ss = c["face_stride"], c["splat_stride"]
    # Faces that are slivers at one of the tested timesteps (height or first edge under a tenth of the median height, measured in fp64): the
    # gradient through such a face's frame is amplified by 1 / height (> 1e4 here) and ANY fp32 evaluation of it is a lottery.  The smooth-weight
    # check leaves the splats bound to them out (weight 0); the white-noise check above keeps them.
    ok = np.ones(len(rig["faces"]), bool)
    for ts in c["steps"]:
        m64.select_mesh_by_timestep(ts)
        v = m64.verts.detach().numpy()[0]
        fa = np.asarray(rig["faces"])
        e1, e2 = v[fa[:, 1]] - v[fa[:, 0]], v[fa[:, 2]] - v[fa[:, 0]]
        l1 = np.linalg.norm(e1, axis=1)
        h = np.linalg.norm(np.cross(e1, e2), axis=1) / l1
        ok &= np.minimum(l1, h) > 0.1 * np.median(h)
    out["smooth_face_ok"] = ok
    print(f"model pins: {int((~ok).sum())} sliver faces of {ok.size} carry no smooth weight ({int((~ok)[np.asarray(sp['binding'])].sum())} splats)")
    for ts in c["steps"]:
        for p in (m._xyz, m._scaling, m._rotation, m._opacity, *m.flame_param.values()):
            p.grad = None
        m.select_mesh_by_timestep(ts)          # FlameHead.forward -> lbs -> update_mesh_properties (reference code)
        x, sc, ro, op = m.get_xyz, m.get_scaling, m.get_rotation, m.get_opacity
        loss = (x * t(w["xyz"])).sum() + (sc * t(w["scaling"])).sum() + (ro * t(w["rotation"])).sum() + (op * t(w["opacity"])).sum()
        loss.backward()
        pre = f"t{ts}_"
        out.update({pre + "verts": m.verts.detach().numpy()[0], pre + "verts_cano": m.verts_cano.detach().numpy()[0],
                    pre + "face_center": m.face_center.detach().numpy()[::fs], pre + "face_orien_mat": m.face_orien_mat.detach().numpy()[::fs],
                    pre + "face_scaling": m.face_scaling.detach().numpy()[::fs], pre + "face_orien_quat": m.face_orien_quat.detach().numpy()[::fs],
                    pre + "xyz": x.detach().numpy()[::ss], pre + "scaling": sc.detach().numpy()[::ss],
                    pre + "rotation": ro.detach().numpy()[::ss], pre + "opacity": op.detach().numpy()[::ss],
                    pre + "loss": np.float64(loss.item()),
                    pre + "g_xyz": m._xyz.grad.numpy()[::ss], pre + "g_scaling": m._scaling.grad.numpy()[::ss],
                    pre + "g_rotation": m._rotation.grad.numpy()[::ss], pre + "g_opacity": m._opacity.grad.numpy()[::ss]})
        for k in ("expr", "rotation", "neck_pose", "jaw_pose", "eyes_pose", "translation"):
            gk = m.flame_param[k].grad.numpy()
            assert np.abs(np.delete(gk, ts, axis=0)).max() == 0.0          # only row t receives gradient
            out[pre + "gf_" + k] = gk[ts]
        # The per-splat weights are white noise, so the FLAME-row gradients are sums with heavy cancellation through the
        # ill-conditioned face frames (1/edge-length factors): the reference's own fp32 result sits up to 5e-4 away from the
        # exact value.  Store the fp64 evaluation of the same reference code as the anchor, and the reference's fp32 deviation
        # from it as the yardstick the tests scale their tolerance with.
        for p in m64.flame_param.values():
            p.grad = None
        m64.select_mesh_by_timestep(ts)
        l64 = ((m64.get_xyz * t64(w["xyz"])).sum() + (m64.get_scaling * t64(w["scaling"])).sum() +
               (m64.get_rotation * t64(w["rotation"])).sum() + (m64.get_opacity * t64(w["opacity"])).sum())
        l64.backward()
        for k in ("expr", "rotation", "neck_pose", "jaw_pose", "eyes_pose", "translation"):
            g64 = m64.flame_param[k].grad.numpy()[ts]
            out[pre + "gf64_" + k] = g64
            out[pre + "gf_dev_" + k] = np.float64(np.abs(out[pre + "gf_" + k] - g64).max() / np.abs(g64).max())
        # the same once more with the SMOOTH weights (tests/model_pin_inputs.py): a well-conditioned loss, for which the tests hold the
        # fused kernels to a tight bar
        w2 = {k: v * ok[np.asarray(sp["binding"])][:, None].astype(v.dtype) for k, v in w["smooth"].items()}
        grads = {}
        for mm, tt, tag in ((m, t, "gfs_"), (m64, t64, "gfs64_")):
            for p in (mm._xyz, mm._scaling, mm._rotation, mm._opacity, *mm.flame_param.values()):   # (the leaf pins above are VIEWS of the old .grad)
                p.grad = None
            mm.select_mesh_by_timestep(ts)
            mm.verts.retain_grad()
            ls = ((mm.get_xyz * tt(w2["xyz"])).sum() + (mm.get_scaling * tt(w2["scaling"])).sum() + (mm.get_rotation * tt(w2["rotation"])).sum() +
                  (mm.get_opacity * tt(w2["opacity"])).sum())
            ls.backward()
            for k in ("expr", "rotation", "neck_pose", "jaw_pose", "eyes_pose", "translation"):
                grads[tag + k] = mm.flame_param[k].grad.numpy()[ts].copy()
            grads[tag + "verts"] = mm.verts.grad.numpy()[0].copy()          # dL/d(posed vertices): where the two halves of the backward meet
            if tag == "gfs64_":
                out[pre + "verts64"] = mm.verts.detach().numpy()[0].copy()
        out[pre + "gfs64_verts"] = grads["gfs64_verts"]
        out[pre + "gfs_dev_verts"] = np.float64(np.abs(grads["gfs_verts"] - grads["gfs64_verts"]).max() / np.abs(grads["gfs64_verts"]).max())
        for k in ("expr", "rotation", "neck_pose", "jaw_pose", "eyes_pose", "translation"):
            out[pre + "gfs64_" + k] = grads["gfs64_" + k]
            out[pre + "gfs_dev_" + k] = np.float64(np.abs(grads["gfs_" + k] - grads["gfs64_" + k]).max() / np.abs(grads["gfs64_" + k]).max())
    # python cov3D path (pipe.compute_cov3D_python): get_covariance uses the LOCAL rotation (scene/gaussian_model.py:162-163)
    with cpu_zeros():
        out["cov3D_mod"] = m.get_covariance(c["cov_mod"]).detach().numpy()[::ss]
    out["cov3D_timestep"] = np.int32(c["steps"][-1])
    # viewer path: update_mesh_by_param_dict (the reference calls .cuda() on the dict entries: identity on this host)
    with mock.patch.object(torch.Tensor, "cuda", lambda self, *a, **k: self):
        with torch.no_grad():
            m.update_mesh_by_param_dict({k: t(v) for k, v in pd.items()})
    out.update(pd_verts=m.verts.numpy()[0], pd_face_center=m.face_center.numpy()[::fs], pd_face_orien_quat=m.face_orien_quat.numpy()[::fs],
               pd_face_scaling=m.face_scaling.numpy()[::fs])
    np.savez_compressed(os.path.join(HERE, "model_pins.npz"), **out)


if __name__ == "__main__":
    raster_pins()
    binding_pins()
    loss_pins()
    model_pins()
    for f in ("raster_pins.npz", "binding_pins.npz", "loss_pins.npz", "model_pins.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")
