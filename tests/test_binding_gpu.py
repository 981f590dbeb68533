"""GPU parity of the fused binding kernels (include/gab.h) against (a) golden vectors generated from
the reference's own lbs / compute_face_orientation code and (b) the composed-torch fp32 restatement
on the same device, forward and backward.  Tolerances (fp32, different summation order):
forward rel 2e-5 of the tensor's max magnitude, gradients rel 2e-4."""
import os

import numpy as np
import pytest
import torch

from gaussianavatars_amd import synthetic as S
from gaussianavatars_amd import unfused as U

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _close(a, b, rtol, what):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    scale = float(b.abs().max()) + 1e-30
    err = float((a - b).abs().max()) / scale
    assert err < rtol, f"{what}: rel err {err:.3e} (max |ref| {scale:.3e})"


class _Head:
    def __init__(self, rig, dev, n_shape):
        for k in ("v_template", "shapedirs", "posedirs", "J_regressor", "lbs_weights"):
            setattr(self, k, torch.as_tensor(rig[k], dtype=torch.float32, device=dev).contiguous())
        self.parents = torch.as_tensor(rig["parents"], device=dev)
        self.n_shape_params = n_shape

    def rigdict(self):
        return dict(v_template=self.v_template, shapedirs=self.shapedirs, posedirs=self.posedirs, J_regressor=self.J_regressor,
                    lbs_weights=self.lbs_weights, parents=self.parents)


def test_flame_forward_matches_reference_golden():
    from gaussianavatars_amd import binding as B

    dev = _dev()
    pins = np.load(os.path.join(G, "binding_pins.npz"))
    rig = {k[4:]: pins[k] for k in pins.files if k.startswith("rig_")}
    head = _Head(rig, dev, 30)
    t = lambda a: torch.as_tensor(a, device=dev)
    betas, pose = t(pins["betas"]), t(pins["pose"])
    verts, v_shaped = B.flame_forward(head, betas[:, :30], betas[:, 30:], pose[:, 0:3], pose[:, 3:6], pose[:, 6:9], pose[:, 9:15],
                                      t(pins["trans"]), t(pins["static_offset"]))
    _close(v_shaped, torch.as_tensor(pins["v_shaped"]), 2e-5, "v_shaped vs reference")
    _close(verts, torch.as_tensor(pins["verts"]), 2e-5, "verts vs reference lbs")
    c, R, s, q = B.face_frames(verts[0], t(pins["faces"]))
    _close(c, torch.as_tensor(pins["face_center"]), 2e-5, "face_center")
    _close(R, torch.as_tensor(pins["face_R"]), 5e-5, "face_orien_mat vs compute_face_orientation")
    _close(s, torch.as_tensor(pins["face_scale"]), 2e-5, "face_scaling")
    qs = torch.as_tensor(pins["face_quat_xyzw_scipy"])
    qx = torch.roll(q.cpu(), -1, dims=-1)
    sign = torch.sign((qx * qs).sum(1, keepdim=True))
    _close(qx * sign, qs, 1e-4, "face_orien_quat vs SciPy")


@pytest.mark.parametrize("with_shape_grad", [False, True])
def test_flame_full_size_forward_backward_vs_torch(with_shape_grad):
    from gaussianavatars_amd import binding as B

    dev = _dev()
    rig = S.flame_rig(4)
    seq = S.flame_sequence(8, 4)
    head = _Head(rig, dev, 300)
    t = lambda a, g=True: torch.as_tensor(a, device=dev).clone().requires_grad_(g)
    mk = lambda: dict(shape=t(seq["shape"][None], with_shape_grad), expr=t(seq["expr"][[3]]), rot=t(seq["rotation"][[3]]),
                      neck=t(seq["neck_pose"][[3]]), jaw=t(seq["jaw_pose"][[3]]), eyes=t(seq["eyes_pose"][[3]]),
                      trans=t(seq["translation"][[3]]), so=t(seq["static_offset"], with_shape_grad))
    a, b = mk(), mk()
    v1, vs1 = B.flame_forward(head, a["shape"], a["expr"], a["rot"], a["neck"], a["jaw"], a["eyes"], a["trans"], a["so"])
    v2, vs2 = U.flame_forward(head.rigdict(), b["shape"], b["expr"], b["rot"], b["neck"], b["jaw"], b["eyes"], b["trans"], b["so"])
    _close(v1, v2, 2e-5, "verts")
    _close(vs1, vs2, 2e-5, "v_shaped")
    g = torch.Generator(device="cpu").manual_seed(0)
    w1 = torch.randn(v1.shape, generator=g).to(dev)
    w2 = torch.randn(v1.shape, generator=g).to(dev) * 0.1
    ((v1 * w1).sum() + (vs1 * w2).sum()).backward()
    ((v2 * w1).sum() + (vs2 * w2).sum()).backward()
    for k in a:
        if a[k].requires_grad:
            _close(a[k].grad, b[k].grad, 3e-4, f"d{k}")
        else:
            assert a[k].grad is None


def test_flame_prepared_rig_path_vs_torch_and_classic():
    """The training configuration (shape and static_offset not optimised, no gradient into v_shaped): one forward launch and two
    backward launches on the prepared rig (include/gab.h: gab_flame_prepare).  Same values and gradients as composed torch and as the
    classic three-kernel path; the prepared buffer is re-made when shape changes in place."""
    from gaussianavatars_amd import binding as B

    dev = _dev()
    rig = S.flame_rig(4)
    seq = S.flame_sequence(8, 4)
    t = lambda a, g=True: torch.as_tensor(a, device=dev).clone().requires_grad_(g)
    mk = lambda: dict(shape=t(seq["shape"][None], False), expr=t(seq["expr"][[5]]), rot=t(seq["rotation"][[5]]),
                      neck=t(seq["neck_pose"][[5]]), jaw=t(seq["jaw_pose"][[5]]), eyes=t(seq["eyes_pose"][[5]]),
                      trans=t(seq["translation"][[5]]), so=t(seq["static_offset"], False))
    heads = [_Head(rig, dev, 300) for _ in range(2)]
    heads[1].flame_impl = "classic"
    a, b, c = mk(), mk(), mk()
    v1, vs1 = B.flame_forward(heads[0], a["shape"], a["expr"], a["rot"], a["neck"], a["jaw"], a["eyes"], a["trans"], a["so"])
    v3, vs3 = B.flame_forward(heads[1], c["shape"], c["expr"], c["rot"], c["neck"], c["jaw"], c["eyes"], c["trans"], c["so"])
    v2, vs2 = U.flame_forward(heads[0].rigdict(), b["shape"], b["expr"], b["rot"], b["neck"], b["jaw"], b["eyes"], b["trans"], b["so"])
    assert getattr(heads[0], "_gab_prepared", None) is not None and getattr(heads[1], "_gab_prepared", None) is None
    _close(v1, v2, 2e-5, "verts (prepared) vs torch")
    _close(vs1, vs2, 2e-5, "v_shaped (prepared) vs torch")
    _close(v1, v3, 1e-5, "verts prepared vs classic")
    w1 = torch.randn(v1.shape, generator=torch.Generator(device="cpu").manual_seed(1)).to(dev)
    for v in (v1, v2, v3):
        (v * w1).sum().backward()
    for k in a:
        if a[k].requires_grad:
            _close(a[k].grad, b[k].grad, 3e-4, f"d{k} (prepared) vs torch")
            _close(a[k].grad, c[k].grad, 1e-4, f"d{k} prepared vs classic")
    # an in-place change of shape invalidates the prepared buffer
    before = heads[0]._gab_prepared[1]
    with torch.no_grad():
        a["shape"].mul_(0.5)
        b["shape"].mul_(0.5)
    v1b, _ = B.flame_forward(heads[0], a["shape"], a["expr"], a["rot"], a["neck"], a["jaw"], a["eyes"], a["trans"], a["so"])
    v2b, _ = U.flame_forward(heads[0].rigdict(), b["shape"], b["expr"], b["rot"], b["neck"], b["jaw"], b["eyes"], b["trans"], b["so"])
    assert heads[0]._gab_prepared[1] is not before
    _close(v1b, v2b, 2e-5, "verts after an in-place shape update")


def test_sequence_table_on_the_matrix_cores_equals_the_per_frame_kernel(monkeypatch):
    """include/gab.h: gab_blend_sequence -- v_shaped of every frame of an expression sequence as ONE (T x 100) . (100 x 3V) fp32 product on
    the matrix cores (v_mfma_f32_32x32x2_f32) -- and gab_flame_forward_sequence, the per-frame forward that takes its row of that table.
    Stated tolerance: the product's terms are the per-frame kernel's, summed in another order: |difference| <= 1e-6 of the vertices' range
    (measured ~1e-7); posed vertices likewise.  The table is built on the SECOND gradient-free frame of a table (binding._sequence_table),
    used from then on, bypassed by frames that need gradients, and re-made when `expr` changes in place."""
    from gaussianavatars_amd import binding as B

    dev = _dev()
    rig = S.flame_rig(4)
    T = 45                                  # not a multiple of the 32-frame tiles
    seq = S.flame_sequence(T, 4)
    head = _Head(rig, dev, 300)
    fp = {k: torch.as_tensor(v, device=dev).clone() for k, v in seq.items()}
    monkeypatch.setenv("GAA_MESH_SEQUENCE", "off")
    with torch.no_grad():
        want = [B.flame_forward_timestep(head, fp, t) for t in range(T)]
    monkeypatch.setenv("GAA_MESH_SEQUENCE", "auto")
    with torch.no_grad():
        B.flame_forward_timestep(head, fp, 0)
        assert head._gab_sequence[1] is None                      # seen once: nothing built yet
        got1 = B.flame_forward_timestep(head, fp, 1)
        table = head._gab_sequence[1]
        assert table is not None and tuple(table.shape) == (T, 3 * 5143)
        got = [B.flame_forward_timestep(head, fp, t) for t in range(T)]
        assert head._gab_sequence[1] is table                     # one table for the whole sequence
    rng = float(torch.stack([w[1] for w in want]).abs().max())
    for t in range(T):
        (v, vs), (wv, wvs) = got[t], want[t]
        assert float((table[t].view(-1) - wvs.view(-1)).abs().max()) <= 1e-6 * rng, t
        assert torch.equal(vs.view(-1), table[t].view(-1))       # the frame's v_shaped IS its row
        assert float((v - wv).abs().max()) <= 2e-6 * rng, t
    assert float((got1[0] - want[1][0]).abs().max()) <= 2e-6 * rng
    # a frame that needs gradients does not touch the table; its values are the per-frame kernel's
    fpg = dict(fp, expr=fp["expr"].clone().requires_grad_(True))
    v, _ = B.flame_forward_timestep(head, fpg, 7)
    v.sum().backward()
    assert fpg["expr"].grad is not None and head._gab_sequence[1] is table
    # an in-place change of the expression table: the old table is not used again
    with torch.no_grad():
        fp["expr"].mul_(0.5)
        B.flame_forward_timestep(head, fp, 3)                     # new key, seen once
        v2, vs2 = B.flame_forward_timestep(head, fp, 3)           # builds
        assert head._gab_sequence[1] is not table
        monkeypatch.setenv("GAA_MESH_SEQUENCE", "off")
        w2, ws2 = B.flame_forward_timestep(head, fp, 3)
    assert float((vs2 - ws2).abs().max()) <= 1e-6 * rng and float((v2 - w2).abs().max()) <= 2e-6 * rng


@pytest.mark.parametrize("with_verts_grad", [False, True])
def test_mesh_backward_gather_equals_the_scatter_form(monkeypatch, with_verts_grad):
    """select_mesh_by_timestep + update_mesh_properties as one autograd node (binding.mesh_frames_timestep): its backward with the face-frame
    and skinning backward as ONE gathering launch (gab_mesh_backward_prepared: no d_verts buffer, no global atomics into the vertices)
    against the scatter + skinning launches it replaces, full-size rig, every per-face output weighted, with and without a gradient
    arriving at the posed vertices themselves; twice over the same forward.  Both are fp32 sums in different orders (float atomics land in arrival
    order, so even the same kernel differs run to run): 1e-4 of each row's max -- except d_translation, 1e-3: it is the plain sum of the 5143
    per-vertex gradients, whose face-frame parts (orientation, scale: translation invariant) cancel exactly in mathematics and are hundreds of
    times larger than what is left, so the sum of 161 workgroup partials carries their rounding (tools/mesh_bwd_repeat.py, 40 repetitions of this
    test's body on an MI355X: worst run-to-run difference 1.6e-4 of the row's max for translation, 2.6e-5 for the rotation, <= 1.2e-5 elsewhere;
    the composed-torch formulation sums the same terms in fp32 and is no better conditioned).  The gathering form sums d_translation from the face
    centres' gradients since (5e-7): it is held to 1e-4 on every row, the 1e-3 is the scatter form's and the comparison's."""
    from gaussianavatars_amd import binding as B

    dev = _dev()
    rig = S.flame_rig(4)
    seq = S.flame_sequence(8, 4)
    head = _Head(rig, dev, 300)
    faces = torch.as_tensor(rig["faces"], device=dev)
    F = faces.shape[0]
    gen = torch.Generator(device="cpu").manual_seed(7)
    wts = [torch.randn(s, generator=gen).to(dev) for s in ((F, 3), (F, 3, 3), (F, 1), (F, 4), (1, rig["v_template"].shape[0], 3))]
    keys = ("expr", "rotation", "neck_pose", "jaw_pose", "eyes_pose", "translation")
    tol = {k: (1e-3 if k == "translation" else 1e-4) for k in keys}
    res = {}
    for mode in ("merged", "split"):
        monkeypatch.setenv("GAA_MESH_BWD", mode)
        fp = {k: torch.as_tensor(v, device=dev).clone().requires_grad_(k in keys) for k, v in seq.items()}
        verts, cano, center, R, scale, quat = B.mesh_frames_timestep(head, fp, 5, faces)
        loss = (center * wts[0]).sum() + (R * wts[1]).sum() + (scale * wts[2]).sum() + (quat * wts[3]).sum()
        if with_verts_grad:
            loss = loss + (verts * wts[4]).sum()
        loss.backward(retain_graph=True)
        first = {k: fp[k].grad.clone() for k in keys}
        for k in keys:
            fp[k].grad = None
        loss.backward()
        for k in keys:   # (the gathering form sums d_translation from the face centres' gradients: no cancelling sum, 1e-4 like the rest)
            bar = 1e-4 if mode == "merged" else tol[k]
            assert float((fp[k].grad - first[k]).abs().max()) <= bar * float(first[k].abs().max()), f"{mode} {k}: a second backward over the same forward differs"
            assert float(fp[k].grad[:5].abs().max()) == 0.0 and float(fp[k].grad[6:].abs().max()) == 0.0, f"{k}: gradient outside row 5"
        res[mode] = first
    for k in keys:
        _close(res["merged"][k][5], res["split"][k][5], tol[k], f"d_{k}: gather vs scatter")


def test_face_frames_and_bind_forward_backward_vs_torch():
    from gaussianavatars_amd import binding as B

    dev = _dev()
    verts0, faces = S.head_mesh()
    g = np.random.default_rng(3)
    verts0 = verts0 + g.normal(0, 5e-4, verts0.shape).astype(np.float32)
    sp = S.bound_splats(30000, S.FLAME_F, 3, 2)
    # three backward variants: atomics (no CSR), the two-pass CSR form (the model's), the one-pass CSR kernel
    for idx_dtype, csr_form in ((torch.int64, None), (torch.int32, "two-pass"), (torch.int32, "one-pass")):
        t = lambda a: torch.as_tensor(a, device=dev).clone().requires_grad_(True)
        va, vb = t(verts0), t(verts0)
        fa = torch.as_tensor(faces, device=dev).to(idx_dtype)
        binding = torch.as_tensor(sp["binding"], device=dev).to(idx_dtype)
        xa, sa, ra = t(sp["_xyz"]), t(sp["_scaling"]), t(sp["_rotation"])
        xb, sb, rb = t(sp["_xyz"]), t(sp["_scaling"]), t(sp["_rotation"])
        c1, R1, s1, q1 = B.face_frames(va, fa)
        c2, R2, s2, q2 = U.face_frames(vb, fa.long())
        for n, x, y in (("center", c1, c2), ("R", R1, R2), ("scale", s1, s2), ("quat", q1, q2)):
            _close(x, y, 5e-5, f"face {n}")
        csr = B.binding_csr(binding, fa.shape[0]) if csr_form else None
        if csr_form == "two-pass":
            order, face_begin, splat_face, slot = csr
            assert torch.equal(slot[order.long()].long(), torch.arange(order.numel(), device=dev)) and torch.equal(splat_face.long(), binding.long())
        elif csr_form == "one-pass":
            csr = csr[:2]
        o1 = B.bind_splats(xa, sa, ra, binding, R1, s1, c1, q1, csr=csr)
        o2 = (U.bind_xyz(xb, binding, R2, s2, c2), U.bind_scaling(sb, binding, s2), U.bind_rotation(rb, binding, q2))
        gen = torch.Generator(device="cpu").manual_seed(1)
        loss1 = loss2 = 0.0
        for n, x, y in zip(("xyz", "scaling", "rotation"), o1, o2):
            _close(x, y, 2e-5, f"bound {n}")
            w = torch.randn(x.shape, generator=gen).to(dev)
            loss1 = loss1 + (x * w).sum()
            loss2 = loss2 + (y * w).sum()
        loss1.backward()
        loss2.backward()
        for n, x, y in (("_xyz", xa, xb), ("_scaling", sa, sb), ("_rotation", ra, rb), ("verts", va, vb)):
            _close(x.grad, y.grad, 3e-4, f"d{n} ({idx_dtype}, {csr_form})")


def test_model_fused_equals_unfused_end_to_end():
    """select_mesh_by_timestep -> render -> L1 -> backward through the mirrored model, both binding modes."""
    import bench

    dev = _dev()
    outs = {}
    for mode in ("fused", "unfused"):
        g, cam = bench.build_scene(dev, 20000, 3, 200, 288, 6, mode, True)
        bg = torch.ones(3, device=dev)
        target = torch.ones((3, 288, 200), device=dev)
        loss = bench.one_step(g, cam, bg, target, 2, True)
        outs[mode] = dict(loss=loss, xyz=g._xyz.grad, rot=g._rotation.grad, sc=g._scaling.grad, expr=g.flame_param["expr"].grad,
                          jaw=g.flame_param["jaw_pose"].grad, trans=g.flame_param["translation"].grad, dc=g._features_dc.grad)
    for k in outs["fused"]:
        _close(outs["fused"][k], outs["unfused"][k], 2e-3, f"end-to-end {k}")
