"""The zero-edit model-side boundary without a GPU: import shims, `patch_reference()` on the
reference's own classes (when /root/reference is present -- it is not on the GPU box, those tests skip there), and
`patch_classes` on the repository's reference-shaped mirror classes."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
needs_ref = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "scene")), reason="reference checkout not present on this box")


def _run(code: str) -> str:
    """Runs `code` in a fresh interpreter (the reference's top-level packages `scene`, `utils`, ... stay out of this process)."""
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-c", textwrap.dedent(code)], cwd=REF, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    return r.stdout


@needs_ref
def test_patch_reference_rebinds_the_reference_classes_and_dispatches_to_the_fused_path():
    out = _run("""
        import torch
        from gaussianavatars_amd import patch
        info = patch.patch_reference()
        import scene.gaussian_model as gm, scene.flame_gaussian_model as fgm, flame_model.flame as fl, gaussian_renderer as gr
        import gaussianavatars_amd.gaussian_renderer as mirror
        assert fgm.FlameGaussianModel.select_mesh_by_timestep is patch._select_mesh_by_timestep
        assert fgm.FlameGaussianModel.update_mesh_properties is patch._update_mesh_properties
        for name in ("get_xyz", "get_scaling", "get_rotation", "get_opacity"):
            assert "gaussianavatars_amd" in gm.GaussianModel.__dict__[name].fget.__module__, name
        assert isinstance(gm.GaussianModel.get_features_split, property)
        assert gr.render is mirror.render
        assert {"roma", "plyfile", "simple_knn._C", "nvdiffrast.torch"} <= set(info["shims"])
        import diff_gaussian_rasterization as dgr, gaussianavatars_amd.rasterizer as R
        assert dgr.GaussianRasterizer is R.GaussianRasterizer            # the HIP rasterizer serves the reference's import
        # a reference-class instance on host tensors: the patched methods run and refuse (no CPU implementation)
        import numpy as np
        from gaussianavatars_amd import synthetic as S
        rig, seq = S.flame_rig(4), S.flame_sequence(2, 4)
        head = fl.FlameHead.__new__(fl.FlameHead); torch.nn.Module.__init__(head)
        head.n_shape_params, head.n_expr_params, head.dtype = 300, 100, torch.float32
        for k in ("v_template", "shapedirs", "posedirs", "J_regressor", "lbs_weights", "parents", "faces"):
            head.register_buffer(k, torch.tensor(rig[k]))
        m = fgm.FlameGaussianModel.__new__(fgm.FlameGaussianModel); gm.GaussianModel.__init__(m, 3)
        m.flame_model, m.flame_param, m.flame_param_orig = head, {k: torch.tensor(v) for k, v in seq.items()}, None
        sp = S.bound_splats(S.FLAME_F, S.FLAME_F, 3, 2)
        for k in ("_xyz", "_scaling", "_rotation", "_opacity"):
            setattr(m, k, torch.tensor(sp[k]))
        m.binding = torch.tensor(sp["binding"])
        for call in (lambda: m.select_mesh_by_timestep(0), lambda: m.get_xyz,
                     lambda: head(torch.zeros(1, 300), torch.zeros(1, 100), *[torch.zeros(1, 3)] * 3, torch.zeros(1, 6), torch.zeros(1, 3),
                                  return_landmarks=False)):
            try:
                call()
            except RuntimeError as e:
                assert "no CPU implementation" in str(e), e
            else:
                raise AssertionError("a patched method ran on host tensors")
        # the opt-out runs the reference's own composed-torch code (and un-bound models are untouched)
        m.binding_impl = head.impl = "unfused"
        m.select_mesh_by_timestep(1)
        assert m.get_xyz.shape == (S.FLAME_F, 3) and m.face_orien_quat.shape == (S.FLAME_F, 4)
        u = gm.GaussianModel(3); u._xyz = torch.ones(4, 3)
        assert u.get_xyz is u._xyz
        print("OK")
    """)
    assert out.strip().endswith("OK")


@needs_ref
def test_the_references_own_render_drives_the_drop_in_rasterizer_package():
    """/root/reference/gaussian_renderer/__init__.py:19-101, unmodified, against the top-level
    `diff_gaussian_rasterization` package of this repository.  No GPU here, so `.cuda()` / device="cuda" are neutralised and the
    native call is replaced by a recorder at the autograd-Function boundary: the test checks everything on the way there -- the
    12-field settings tuple built by keyword, the keyword call of GaussianRasterizer.forward, both Exceptions of the argument
    contract, the (image, radii) pair coming back and `radii > 0` -- and that without the recorder the same call ends in the
    rasterizer's "no CPU path" error (i.e. it really is the HIP entry that is reached)."""
    out = _run("""
        import math, types, torch
        from unittest import mock
        from gaussianavatars_amd import shims
        shims.install(stub_torchvision=True)
        import gaussian_renderer as gr                      # the reference's module, importing OUR diff_gaussian_rasterization
        import diff_gaussian_rasterization as dgr, gaussianavatars_amd.rasterizer as R
        assert gr.GaussianRasterizer is R.GaussianRasterizer and gr.render.__module__ == "gaussian_renderer"
        from gaussianavatars_amd import synthetic as S
        from scene.gaussian_model import GaussianModel
        sp = S.random_splats(500, 3, 3)
        pc = GaussianModel(3)
        pc._xyz, pc._scaling, pc._rotation = torch.tensor(sp["means3D"]), torch.tensor(sp["scales"]).log(), torch.tensor(sp["rotations"])
        pc._opacity = torch.logit(torch.tensor(sp["opacities"]).clamp(1e-4, 1 - 1e-4))
        pc._features_dc, pc._features_rest = torch.tensor(sp["shs"][:, :1]), torch.tensor(sp["shs"][:, 1:])
        pc.active_sh_degree = 3
        cam = S.orbit_camera(64, 48)
        for k in ("world_view_transform", "full_proj_transform", "camera_center"):
            setattr(cam, k, torch.as_tensor(getattr(cam, k)))
        pipe = types.SimpleNamespace(debug=False, compute_cov3D_python=False, convert_SHs_python=False)
        seen = {}
        def recorder(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, rs, sh_rest=None):
            seen.update(means3D=means3D, means2D=means2D, sh=sh, colors=colors_precomp, opacities=opacities, scales=scales, rotations=rotations,
                        cov=cov3Ds_precomp, rs=rs)
            P = means3D.shape[0]
            radii = torch.arange(P, dtype=torch.int32) % 3
            return torch.zeros(3, rs.image_height, rs.image_width), radii, radii > 0
        zl = torch.zeros_like
        with mock.patch.object(torch.Tensor, "cuda", lambda self, *a, **k: self), \
             mock.patch("torch.zeros_like", lambda t, **k: zl(t, **{kk: v for kk, v in k.items() if kk != "device"})):
            with mock.patch.object(R._RasterizeGaussians, "apply", staticmethod(recorder)):
                pkg = gr.render(cam, pc, pipe, torch.ones(3))
            rs = seen["rs"]
            assert rs._fields == ("image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix", "projmatrix",
                                  "sh_degree", "campos", "prefiltered", "debug")
            assert (rs.image_height, rs.image_width, rs.sh_degree, rs.scale_modifier) == (cam.image_height, cam.image_width, 3, 1.0)
            assert abs(rs.tanfovx - math.tan(cam.FoVx * 0.5)) < 1e-12
            assert seen["means3D"] is pc._xyz and tuple(seen["sh"].shape) == (500, 16, 3) and seen["colors"].numel() == 0 and seen["cov"].numel() == 0
            assert tuple(seen["scales"].shape) == (500, 3) and tuple(seen["rotations"].shape) == (500, 4) and tuple(seen["opacities"].shape) == (500, 1)
            assert seen["means2D"].requires_grad and tuple(seen["means2D"].shape) == (500, 3)
            assert set(pkg) == {"render", "viewspace_points", "visibility_filter", "radii"} and pkg["visibility_filter"].dtype == torch.bool
            assert tuple(pkg["render"].shape) == (3, cam.image_height, cam.image_width)
            # without the recorder: the native entry is reached and refuses host tensors
            try:
                gr.render(cam, pc, pipe, torch.ones(3))
            except RuntimeError as e:
                assert "no CPU path" in str(e), e
            else:
                raise AssertionError("the rasterizer accepted host tensors")
            # the argument contract's two Exceptions (rasterizer __init__.py of the upstream package), through the reference's call
            rast = dgr.GaussianRasterizer(raster_settings=rs)
            for kw, msg in ((dict(shs=None, colors_precomp=None, scales=seen["scales"], rotations=seen["rotations"]), "SHs or precomputed colors"),
                            (dict(shs=seen["sh"], scales=None, rotations=None, cov3D_precomp=None), "scale/rotation pair or precomputed 3D covariance")):
                try:
                    rast(means3D=pc._xyz, means2D=seen["means2D"], opacities=seen["opacities"], **kw)
                except Exception as e:
                    assert msg in str(e), e
                else:
                    raise AssertionError("no exception")
        print("OK")
    """)
    assert out.strip().endswith("OK")


@needs_ref
def test_entry_scripts_import_closure_is_complete():
    """Every module-level import of the reference's entry scripts resolves once the shims are installed."""
    out = _run("""
        import ast, sys
        from gaussianavatars_amd import patch
        patch.patch_reference()
        for script in ("train.py", "render.py", "fps_benchmark_demo.py", "fps_benchmark_dataset.py", "metrics.py"):
            for node in ast.parse(open(script).read()).body:
                if isinstance(node, (ast.Import, ast.ImportFrom)):
                    exec(compile(ast.Module([node], []), script, "exec"), {})
        print("OK")
    """)
    assert out.strip().endswith("OK")


@needs_ref
def test_reference_ply_writer_and_reader_interoperate_with_io():
    """scene/gaussian_model.py:253-275 (through the plyfile stand-in) writes what gaussianavatars_amd.io reads, and the
    reference's load_ply (:282-332) reads what io writes: property order, channel-major f_rest, binding_0 as float -> int32."""
    out = _run("""
        import os, tempfile, numpy as np, torch
        from unittest import mock
        from gaussianavatars_amd import shims, io as gio, synthetic as S
        shims.install(stub_torchvision=True)
        from scene.gaussian_model import GaussianModel
        sp = S.bound_splats(300, 120, 3, 9)
        m = GaussianModel(3)
        for k in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation"):
            setattr(m, k, torch.tensor(sp[k]))
        m.binding = torch.tensor(sp["binding"])
        d = tempfile.mkdtemp()
        m.save_ply(os.path.join(d, "ref.ply"))                         # the reference's writer
        back = gio.load_ply(os.path.join(d, "ref.ply"), 3)
        for k in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation"):
            assert np.array_equal(back[k], sp[k]), k
        assert back["binding"].dtype == np.int32 and np.array_equal(back["binding"], sp["binding"])
        gio.save_ply(os.path.join(d, "ours.ply"), sp)
        assert open(os.path.join(d, "ours.ply"), "rb").read() == open(os.path.join(d, "ref.ply"), "rb").read()   # byte-identical files
        orig = torch.tensor
        with mock.patch("torch.tensor", lambda *a, **k: orig(*a, **{kk: v for kk, v in k.items() if kk != "device"})):
            m2 = GaussianModel(3); m2.load_ply(os.path.join(d, "ours.ply"))   # the reference's reader
        for k in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation"):
            assert np.array_equal(getattr(m2, k).detach().numpy(), sp[k]), k
        assert m2.binding.dtype == torch.int32 and np.array_equal(m2.binding.numpy(), sp["binding"])
        print("OK")
    """)
    assert out.strip().endswith("OK")


@needs_ref
def test_patched_reference_model_keeps_its_splats_in_morton_order():
    """patch._hook_spatial_order: the reference's own GaussianModel.load_ply (scene/gaussian_model.py:282-332), rebound by
    patch_reference(), leaves the splats in Morton order of their positions (the default; every per-splat tensor permuted alike), and in the
    file's order with GAA_SPATIAL_SORT=0."""
    out = _run("""
        import os, tempfile, numpy as np, torch
        from unittest import mock
        from gaussianavatars_amd import patch, io as gio, synthetic as S
        patch.patch_reference(pin=False)
        from scene.gaussian_model import GaussianModel
        sp = S.bound_splats(400, 120, 3, 9)
        sp.pop("binding")
        d = tempfile.mkdtemp()
        gio.save_ply(os.path.join(d, "u.ply"), sp)
        orig = torch.tensor
        def load():
            with mock.patch("torch.tensor", lambda *a, **k: orig(*a, **{kk: v for kk, v in k.items() if kk != "device"})):
                m = GaussianModel(3); m.load_ply(os.path.join(d, "u.ply"))
            return m
        m = load()
        x = m._xyz.detach().numpy()
        perm = gio.morton_order(sp["_xyz"])
        assert not np.array_equal(x, sp["_xyz"]) and np.array_equal(x, sp["_xyz"][perm])
        for k in ("_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation"):
            assert np.array_equal(getattr(m, k).detach().numpy(), sp[k][perm]), k
        os.environ["GAA_SPATIAL_SORT"] = "0"
        assert np.array_equal(load()._xyz.detach().numpy(), sp["_xyz"])
        print("OK")
    """)
    assert out.strip().endswith("OK")


def test_shims_only_fill_gaps_and_uninstall_cleanly():
    from gaussianavatars_amd import shims

    before = set(sys.modules)
    served = shims.install()
    try:
        import numpy  # a real package is never shadowed

        assert "numpy" not in served and not getattr(numpy, "__gaussianavatars_amd_shim__", False)
        if "roma" in served:
            import roma

            q = torch.nn.functional.normalize(torch.randn(7, 4), dim=1)
            assert torch.allclose(roma.quat_wxyz_to_xyzw(roma.quat_xyzw_to_wxyz(q)), q)
            ident = torch.tensor([[0.0, 0.0, 0.0, 1.0]]).expand(7, 4)
            assert torch.allclose(roma.quat_product(ident, q), q, atol=1e-7)
        if "dearpygui.dearpygui" in served:
            import dearpygui.dearpygui as dpg

            with pytest.raises(RuntimeError, match="import-time stub"):
                dpg.create_context()
    finally:
        shims.uninstall()
    left = [m for m in set(sys.modules) - before if getattr(sys.modules[m], "__gaussianavatars_amd_shim__", False)]
    assert not [m for m in left if not m.startswith("gaussianavatars_amd.shims")], left   # no third-party name stays served


def test_plyfile_stand_in_round_trip_and_header():
    from gaussianavatars_amd.shims import plyfile

    d = np.zeros(5, dtype=[("x", "f4"), ("y", "f4"), ("red", "u1")])
    d["x"], d["red"] = np.arange(5), 7
    import io

    buf = io.BytesIO()
    plyfile.PlyData([plyfile.PlyElement.describe(d, "vertex")]).write(buf)
    raw = buf.getvalue()
    head = b"ply\nformat binary_little_endian 1.0\nelement vertex 5\nproperty float x\nproperty float y\nproperty uchar red\nend_header\n"
    assert raw.startswith(head) and len(raw) == len(head) + 5 * 9
    back = plyfile.PlyData.read(io.BytesIO(raw))
    assert [p.name for p in back.elements[0].properties] == ["x", "y", "red"]
    assert np.array_equal(back["vertex"]["x"], d["x"]) and np.array_equal(back.elements[0]["red"], d["red"])
    with pytest.raises(plyfile.PlyHeaderParseError):
        plyfile.PlyData.read(io.BytesIO(b"ply\nformat binary_little_endian 1.0\nelement face 1\nproperty list uchar int vertex_indices\nend_header\n"))


def test_distCUDA2_stand_in_matches_brute_force():
    from gaussianavatars_amd.shims.simple_knn._C import distCUDA2

    p = torch.randn(257, 3, generator=torch.Generator().manual_seed(3))
    d2 = ((p[:, None] - p[None]) ** 2).sum(-1)
    d2.fill_diagonal_(float("inf"))
    ref = d2.topk(3, dim=1, largest=False).values.mean(1)
    assert torch.allclose(distCUDA2(p), ref, rtol=1e-4, atol=1e-6)


def test_patched_mirror_classes_keep_the_reference_surface():
    """The repository's mirror classes go through the same patch_classes call: accessors are properties, un-bound models take
    the original path, lazy mesh initialisation still happens, and host tensors are refused by the fused path."""
    from gaussianavatars_amd import patch, synthetic as S
    from gaussianavatars_amd.gaussian_model import FlameGaussianModel, GaussianModel

    assert GaussianModel._gaa_patched and FlameGaussianModel._gaa_patched_flame
    u = GaussianModel(3)
    u._xyz, u._scaling, u._rotation, u._opacity = torch.ones(4, 3), torch.zeros(4, 3), torch.ones(4, 4), torch.zeros(4, 1)
    assert u.get_xyz is u._xyz and torch.equal(u.get_scaling, torch.ones(4, 3)) and torch.allclose(u.get_opacity, torch.full((4, 1), 0.5))
    g = FlameGaussianModel(3, S.flame_rig(4), device="cpu")
    g.load_arrays(S.bound_splats(S.FLAME_F, S.FLAME_F, 3, 2), device="cpu", requires_grad=False)
    g.load_flame_param(S.flame_sequence(2, 4), device="cpu")
    with pytest.raises(RuntimeError, match="no CPU implementation"):
        g.get_xyz                                  # lazy select_mesh_by_timestep(0) -> fused -> refused on the host
    g.binding_impl = "unfused"
    g.flame_model.impl = "unfused"
    assert g.get_xyz.shape == (S.FLAME_F, 3) and g.timestep == 0
    assert type(g).select_mesh_by_timestep is patch._select_mesh_by_timestep


def test_loss_and_statistics_rebinding_keeps_host_tensors_on_the_reference_functions():
    """patch_loss_and_stats (SURVEY.md 8(f) N3 behind the zero-edit boundary): utils.loss_utils.l1_loss / ssim and
    GaussianModel.add_densification_stats are rebound; whatever the kernels do not take -- here: host tensors -- reaches the reference's own
    functions, with the reference's results (the GPU side: tests/test_reference_classes_gpu.py, tests/test_loss_gpu.py)."""
    import types

    import torch

    from gaussianavatars_amd import patch as P

    calls = []
    lu = types.ModuleType("fake_loss_utils")
    lu.l1_loss = lambda a, b: (calls.append("l1"), torch.abs(a - b).mean())[1]
    lu.ssim = lambda a, b, window_size=11, size_average=True: (calls.append("ssim"), torch.tensor(0.5))[1]

    class G:
        def add_densification_stats(self, vsp, f):
            raise AssertionError("the original method must have been replaced")

    done = P.patch_loss_and_stats(G, loss_utils=lu)
    assert done == ["utils.loss_utils.l1_loss", "utils.loss_utils.ssim", "G.add_densification_stats"]
    assert P.patch_loss_and_stats(G, loss_utils=lu) == []          # idempotent
    a, b = torch.rand(3, 8, 8), torch.rand(3, 8, 8)
    assert float(lu.l1_loss(a, b)) == float(torch.abs(a - b).mean()) and float(lu.ssim(a, b)) == 0.5
    assert calls == ["l1", "ssim"]
    g = G()
    n = 50
    g.xyz_gradient_accum, g.denom = torch.rand(n, 1), torch.rand(n, 1)
    vsp = torch.zeros(n, 3, requires_grad=True)
    vsp.grad = torch.randn(n, 3)
    f = torch.rand(n) > 0.5
    want_acc, want_den = g.xyz_gradient_accum.clone(), g.denom.clone()
    want_acc[f] += torch.norm(vsp.grad[f, :2], dim=-1, keepdim=True)
    want_den[f] += 1
    g.add_densification_stats(vsp, f)
    assert torch.equal(g.xyz_gradient_accum, want_acc) and torch.equal(g.denom, want_den)
    P.unpatch_classes(G)
    assert "_gaa_patched_stats" not in G.__dict__
