"""CPU tests of the binding half: the composed-torch restatement (gaussianavatars_amd/unfused.py, the
fp32 reference the fused kernels are checked against on the GPU) against golden vectors generated
from the REFERENCE's own flame_model/lbs.py and utils/graphics_utils.py (tests/golden/make_golden.py),
and the roma stand-ins against SciPy."""
import os

import numpy as np
import pytest
import torch

from gaussianavatars_amd import unfused as U

G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def pins():
    return np.load(os.path.join(G, "binding_pins.npz"))


def _rig(pins, device="cpu"):
    return {k[4:]: torch.as_tensor(pins[k], device=device) for k in pins.files if k.startswith("rig_")}


def test_flame_forward_matches_reference_lbs(pins):
    rig = _rig(pins)
    t = torch.as_tensor
    betas, pose = t(pins["betas"]), t(pins["pose"])
    rig2 = dict(rig)
    verts, v_shaped = U.flame_forward(rig2, betas[:, :30], betas[:, 30:], pose[:, 0:3], pose[:, 3:6], pose[:, 6:9], pose[:, 9:15],
                                      t(pins["trans"]), t(pins["static_offset"]))
    np.testing.assert_allclose(v_shaped.numpy(), pins["v_shaped"], atol=1e-6)
    np.testing.assert_allclose(verts.numpy(), pins["verts"], atol=2e-6)


def test_face_frames_match_reference_compute_face_orientation(pins):
    verts = torch.as_tensor(pins["verts"])[0]
    c, R, s, q = U.face_frames(verts, torch.as_tensor(pins["faces"]))
    np.testing.assert_allclose(c.numpy(), pins["face_center"], atol=1e-6)
    np.testing.assert_allclose(R.numpy(), pins["face_R"], atol=1e-5)
    np.testing.assert_allclose(s.numpy(), pins["face_scale"], atol=1e-6)
    # quaternion: SciPy's from_matrix (the algorithm roma implements), equal up to sign
    qs = pins["face_quat_xyzw_scipy"]
    qx = U.quat_wxyz_to_xyzw(q).numpy()
    sign = np.sign((qx * qs).sum(1, keepdims=True))
    np.testing.assert_allclose(qx * sign, qs, atol=2e-5)


def test_quat_product_matches_scipy():
    from scipy.spatial.transform import Rotation

    g = np.random.default_rng(0)
    p, q = g.normal(size=(50, 4)), g.normal(size=(50, 4))
    p /= np.linalg.norm(p, axis=1, keepdims=True)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    ours = U.quat_product(torch.as_tensor(p), torch.as_tensor(q)).numpy()
    ref = (Rotation.from_quat(p) * Rotation.from_quat(q)).as_quat()
    sign = np.sign((ours * ref).sum(1, keepdims=True))
    np.testing.assert_allclose(ours * sign, ref, atol=1e-12)


def test_bind_rotation_equals_rotation_composition():
    """get_rotation = face rotation o local rotation (scene/gaussian_model.py:125-138)."""
    from scipy.spatial.transform import Rotation

    g = np.random.default_rng(1)
    fq = g.normal(size=(7, 4))
    rot = g.normal(size=(40, 4))
    b = g.integers(0, 7, 40)
    out = U.bind_rotation(torch.as_tensor(rot), torch.as_tensor(b), torch.as_tensor(fq)).numpy()  # WXYZ
    to_xyzw = lambda a: np.roll(a, -1, axis=1)
    Rf = Rotation.from_quat(to_xyzw(fq / np.linalg.norm(fq, axis=1, keepdims=True)))[b]
    Rl = Rotation.from_quat(to_xyzw(rot / np.linalg.norm(rot, axis=1, keepdims=True)))
    np.testing.assert_allclose(Rotation.from_quat(to_xyzw(out)).as_matrix(), (Rf * Rl).as_matrix(), atol=1e-10)


def test_gab_library_exports_every_declared_symbol():
    import re

    from gaussianavatars_amd import _lib

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    txt = re.sub(r"/\*.*?\*/", "", open(os.path.join(root, "include", "gab.h")).read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(gab_[a-z0-9_]+)\s*\(", txt)))
    lib = _lib.gab()
    assert len(names) == 23
    for n in names:
        assert hasattr(lib, n) and n in _lib.GAB_SYMBOLS, n


def test_fused_path_refuses_cpu_tensors():
    from gaussianavatars_amd import binding as B

    with pytest.raises(RuntimeError, match="no CPU implementation"):
        B.face_frames(torch.zeros(4, 3), torch.zeros(2, 3, dtype=torch.long))


def test_binding_csr_arrays_are_consistent():
    """The four arrays gab_bind_backward_csr's two-pass form takes (include/gab.h) depend on `binding` only and are pure torch:
    order groups the splats by face (stable), face_begin are its CSR offsets, slot inverts order, splat_face is the binding."""
    import torch

    from gaussianavatars_amd.binding import binding_csr

    g = torch.Generator().manual_seed(7)
    F, N = 37, 500
    binding = torch.randint(0, F, (N,), generator=g)
    binding[binding == 5] = 6                          # an empty face
    order, face_begin, splat_face, slot = binding_csr(binding, F)
    assert order.dtype == face_begin.dtype == splat_face.dtype == slot.dtype == torch.int32
    assert face_begin.shape == (F + 1,) and int(face_begin[0]) == 0 and int(face_begin[-1]) == N
    assert torch.equal(torch.sort(order.long()).values, torch.arange(N))           # a permutation
    assert torch.equal(slot[order.long()].long(), torch.arange(N))                # slot is its inverse
    assert torch.equal(splat_face.long(), binding)
    for f in range(F):
        seg = order[int(face_begin[f]): int(face_begin[f + 1])].long()
        assert (binding[seg] == f).all() and (seg[1:] > seg[:-1]).all()            # grouped by face, stable inside a face
    assert int(face_begin[6]) == int(face_begin[5])


def test_vertex_corner_table_lists_every_corner_once_by_vertex():
    """The static table gab_mesh_backward_prepared gathers through (include/gab.h): one row (4 f + c, i0, i1, i2) per corner c of face
    f = (i0, i1, i2), grouped by the corner's vertex (stable: a vertex's corners in face order), vf_begin its CSR offsets; pure torch,
    kept on the faces tensor itself (no shared cache that could evict a table a recorded step still reads)."""
    import torch

    from gaussianavatars_amd import synthetic as S
    from gaussianavatars_amd.binding import vertex_corner_csr

    rig = S.flame_rig(4)
    faces = torch.as_tensor(rig["faces"]).long()
    V, F = rig["v_template"].shape[0], faces.shape[0]
    vf_begin, vf_list = vertex_corner_csr(faces, V)
    assert vf_begin.dtype == vf_list.dtype == torch.int32 and vf_begin.shape == (V + 1,) and vf_list.shape == (3 * F, 4)
    assert int(vf_begin[0]) == 0 and int(vf_begin[-1]) == 3 * F
    code = vf_list[:, 0].long()
    f, c = code >> 2, code & 3
    assert int(c.max()) <= 2
    assert torch.equal(torch.sort(3 * f + c).values, torch.arange(3 * F))            # every corner exactly once
    assert torch.equal(vf_list[:, 1:].long(), faces[f])                              # the row carries its face's three vertices
    owner = faces[f, c]                                                              # the vertex the corner belongs to
    assert (owner[1:] >= owner[:-1]).all()                                           # grouped by vertex ...
    same = owner[1:] == owner[:-1]
    assert ((3 * f + c)[1:][same] > (3 * f + c)[:-1][same]).all()                    # ... stable inside a vertex
    counts = torch.bincount(faces.reshape(-1), minlength=V)
    assert torch.equal((vf_begin[1:] - vf_begin[:-1]).long(), counts)
    assert vertex_corner_csr(faces, V)[1] is vf_list                                 # cached
    others = [faces.clone() for _ in range(9)]                                       # nine more topologies do not evict the first one's table
    for o in others:
        assert vertex_corner_csr(o, V)[1] is not vf_list
    assert vertex_corner_csr(faces, V)[1] is vf_list
    faces[0] = faces[0].flip(0)                                                      # modified in place: rebuilt
    assert vertex_corner_csr(faces, V)[1] is not vf_list
