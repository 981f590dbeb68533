"""An UNCHANGED entry script of the reference, run end to end (BASELINE.json north_star: "train.py/render.py/fps_benchmark_*.py run
unchanged"; SURVEY.md 8(f) N2's missing half): /root/reference/fps_benchmark_demo.py:35-89 through `python -m gaussianavatars_amd.run`.

What the script needs and the snapshot lacks is generated in the reference's own formats by
gaussianavatars_amd.synthetic.write_reference_assets: flame2023.pkl + FLAME_masks.pkl (unpickled by the reference's FlameHead,
flame_model/flame.py:83-184, which then builds its masks and adds the teeth), a mesh-bound point_cloud.ply (read by the reference's
GaussianModel.load_ply through the `plyfile` shim) and the flame_param.npz next to it (scene/flame_gaussian_model.py:229-237).
The reference checkout is read-only and its asset paths are relative to the working directory (flame.py:32-38), so the script runs in
a directory of SYMLINKS to the checkout (no file is copied or edited) whose flame_model/assets/flame/ additionally holds the two pickles.

This box has no GPU: inside the subprocess tests/ref_cpu_env.py maps the hard-coded "cuda" to the host and stands the CPU oracle in for
the rasterizer Function (test infrastructure), and GAA_BINDING_IMPL=unfused keeps the model's composed-torch methods; everything else
-- argument parsing, safe_state, FlameGaussianModel(), load_ply, the three timed rounds of select_mesh_by_timestep + render, the FPS
report -- is the reference's code.  On an MI355X the same command without those two stand-ins is the GPU run."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
needs_ref = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "scene")), reason="reference checkout not present on this box")


def symlink_farm(dst):
    """dst/<entry> -> /root/reference/<entry> for everything but flame_model, which becomes a real directory of links whose
    assets/flame/ can take the generated pickles."""
    for e in os.listdir(REF):
        if e != "flame_model":
            os.symlink(os.path.join(REF, e), os.path.join(dst, e))
    fm = os.path.join(dst, "flame_model")
    os.makedirs(os.path.join(fm, "assets", "flame"))
    for e in os.listdir(os.path.join(REF, "flame_model")):
        if e != "assets":
            os.symlink(os.path.join(REF, "flame_model", e), os.path.join(fm, e))
    for e in os.listdir(os.path.join(REF, "flame_model", "assets")):
        if e != "flame":
            os.symlink(os.path.join(REF, "flame_model", "assets", e), os.path.join(fm, "assets", e))
    for e in os.listdir(os.path.join(REF, "flame_model", "assets", "flame")):
        os.symlink(os.path.join(REF, "flame_model", "assets", "flame", e), os.path.join(fm, "assets", "flame", e))
    return os.path.join(fm, "assets", "flame")


@needs_ref
def test_fps_benchmark_demo_runs_unchanged(tmp_path):
    from gaussianavatars_amd import synthetic as S

    farm = str(tmp_path / "checkout")
    os.makedirs(farm)
    asset_dir = symlink_farm(farm)
    out = S.write_reference_assets(asset_dir, str(tmp_path / "avatar"), os.path.join(REF, "flame_model", "assets", "flame", "head_template_mesh.obj"),
                                   n_frames=3)
    assert os.path.islink(os.path.join(farm, "fps_benchmark_demo.py"))   # the script itself: the checkout's file, not a copy
    code = textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {ROOT!r})
        from tests import ref_cpu_env
        stand_in = ref_cpu_env.install()
        from gaussianavatars_amd import run
        run.main(["fps_benchmark_demo.py", "--point_path", {out["point_cloud"]!r}, "--n_iter", "2", "--height", "160", "--width", "112"])
        import numpy as np
        st = stand_in.last
        print("CALLS", stand_in.calls, "VISIBLE", int((st.radii > 0).sum()), "COVERED", float((st.color < 0.999).mean()))
    """)
    env = dict(os.environ, GAA_BINDING_IMPL="unfused", PYTHONPATH=ROOT, MPLBACKEND="Agg")
    r = subprocess.run([sys.executable, "-c", code], cwd=farm, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = r.stdout.splitlines()
    assert [ln.split(" [")[0] for ln in lines if ln.startswith("Round ")] == ["Round 1", "Round 2", "Round 3"]   # (safe_state stamps every line with the time)
    fps = [float(ln.split(":")[1].split(" [")[0]) for ln in lines if ln.startswith("FPS:")]
    assert len(fps) == 3 and all(f > 0 for f in fps)
    tail = [ln for ln in lines if ln.startswith("CALLS")][0].split()
    assert int(tail[1]) == 6                               # three rounds of two frames reached the rasterizer boundary
    assert int(tail[3]) > 5000 and float(tail[5]) > 0.05   # ... with a visible avatar: most of the 10144 bound splats, >5 % of the pixels covered
    assert "fused model methods on GaussianModel, FlameGaussianModel, FlameHead" in r.stderr


def _run_entry(farm, body, timeout=900):
    code = textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {ROOT!r})
        from tests import ref_cpu_env
        stand_in = ref_cpu_env.install()
        import torch
        losses = []
        _orig_backward = torch.Tensor.backward
        def _observed_backward(self, *a, **k):      # observation only: the scalar every loss.backward() of the script starts from
            if self.dim() == 0:
                losses.append(float(self.detach()))
            return _orig_backward(self, *a, **k)
        torch.Tensor.backward = _observed_backward
        from gaussianavatars_amd import run
    """) + textwrap.dedent(body)
    env = dict(os.environ, GAA_BINDING_IMPL="unfused", PYTHONPATH=ROOT, MPLBACKEND="Agg")
    r = subprocess.run([sys.executable, "-c", code], cwd=farm, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return r


@needs_ref
def test_train_render_and_fps_benchmark_dataset_run_unchanged(tmp_path):
    """/root/reference/train.py:36-214 (Scene -> DataLoader -> select_mesh_by_timestep -> render -> L1 + SSIM + the xyz / scale regularisers
    -> backward -> densification statistics -> densify_and_prune -> optimiser step -> save), then render.py:53-135 and
    fps_benchmark_dataset.py:14-60 on the model it wrote -- all three through `python -m gaussianavatars_amd.run`, the scripts themselves
    symlinks into the read-only checkout.  The dataset is generated in the reference's "DynamicNerf" layout
    (gaussianavatars_amd.synthetic.write_reference_dataset; scene/dataset_readers.py:283-352).  Stand-ins, as for the demo above: "cuda" is
    the host, the CPU oracle (forward AND backward) serves the rasterizer Function, the binding runs composed-torch."""
    from gaussianavatars_amd import io as gio
    from gaussianavatars_amd import synthetic as S

    farm = str(tmp_path / "checkout")
    os.makedirs(farm)
    asset_dir = symlink_farm(farm)
    template = os.path.join(REF, "flame_model", "assets", "flame", "head_template_mesh.obj")
    S.write_reference_assets(asset_dir, str(tmp_path / "avatar"), template, n_frames=2)     # the two FLAME pickles FlameHead() opens
    data, model = str(tmp_path / "data"), str(tmp_path / "model")
    info = S.write_reference_dataset(data, template, n_timesteps=4)
    assert (info["train"], info["val"], info["test"]) == (6, 3, 3)
    for script in ("train.py", "render.py", "fps_benchmark_dataset.py"):
        assert os.path.islink(os.path.join(farm, script))

    # ---- train.py: 30 iterations, densify_and_prune at 15 / 20 / 25, the model saved at 30 (no evaluation pass: LPIPS needs downloaded weights)
    r = _run_entry(farm, f"""
        run.main(["train.py", "-s", {data!r}, "-m", {model!r}, "--bind_to_mesh", "--white_background", "--eval", "--iterations", "30",
                  "--densify_from_iter", "10", "--densification_interval", "5", "--densify_until_iter", "26",
                  "--test_iterations", "100000", "--save_iterations", "30", "--checkpoint_iterations", "100000", "--port", "60123"])
        print("LOSSES", " ".join(f"{{x:.6f}}" for x in losses))
        print("CALLS", stand_in.calls, "BACKWARDS", stand_in.backwards)
    """)
    out = r.stdout.splitlines()
    assert any(ln.startswith("Training complete.") for ln in out)
    losses = [float(x) for x in [ln for ln in out if ln.startswith("LOSSES")][0].split()[1:31]]
    assert len(losses) == 30 and np.isfinite(losses).all()
    assert np.mean(losses[-5:]) < 0.7 * np.mean(losses[:5]), f"the loss does not go down: {losses}"
    tail = [ln for ln in out if ln.startswith("CALLS")][0].split()
    assert (int(tail[1]), int(tail[3])) == (30, 30)          # every iteration went forward and backward through the rasterizer boundary
    ply = os.path.join(model, "point_cloud", "iteration_30", "point_cloud.ply")
    trained = gio.load_ply(ply)
    F = 10144
    assert trained["_xyz"].shape[0] > F, "densify_and_prune added no splat"      # the reference's own densification ran on our gradients / statistics
    assert trained["binding"].min() >= 0 and trained["binding"].max() < F
    fp = np.load(os.path.join(model, "point_cloud", "iteration_30", "flame_param.npz"))
    assert fp["expr"].shape == (4, 100) and fp["static_offset"].shape == (1, 5143, 3) and fp["dynamic_offset"].shape == (4, 5143, 3)
    seq = info["flame_sequence"]
    assert not np.array_equal(fp["expr"], seq["expr"]) and np.abs(fp["expr"] - seq["expr"]).max() < 0.1   # fine-tuned by the optimiser, a little
    assert "host CPUs: unchanged" in r.stderr                # (no GPU here: the default pinning is a no-op)

    # ---- render.py on what train.py wrote (val + test splits: 3 + 3 frames, PNGs of renders and targets)
    r = _run_entry(farm, f"""
        run.main(["render.py", "-m", {model!r}, "--skip_train"])
        print("CALLS", stand_in.calls)
    """)
    assert int([ln for ln in r.stdout.splitlines() if ln.startswith("CALLS")][0].split()[1]) == 6
    from PIL import Image

    for split in ("val", "test"):
        for i in range(3):
            img = np.asarray(Image.open(os.path.join(model, split, "ours_30", "renders", f"{i:05d}.png")))
            gt = np.asarray(Image.open(os.path.join(model, split, "ours_30", "gt", f"{i:05d}.png")))
            assert img.shape == gt.shape == (160, 112, 3)
            assert (img < 250).mean() > 0.05                 # an avatar on the white background, not an empty frame

    # ---- fps_benchmark_dataset.py: three timed rounds on the first test view
    r = _run_entry(farm, f"""
        run.main(["fps_benchmark_dataset.py", "-m", {model!r}, "--skip_train", "--skip_val", "--n_iter", "2"])
        print("CALLS", stand_in.calls)
    """)
    lines = r.stdout.splitlines()
    assert [ln.split(" [")[0] for ln in lines if ln.startswith("Round ")] == ["Round 1", "Round 2", "Round 3"]
    fps = [float(ln.split(":")[1].split(" [")[0]) for ln in lines if ln.startswith("FPS:")]
    assert len(fps) == 3 and all(f > 0 for f in fps)
    assert int([ln for ln in lines if ln.startswith("CALLS")][0].split()[1]) == 6


@needs_ref
def test_ref_on_gpu_stage_puts_an_ignored_scratch_copy_with_generated_assets(tmp_path, monkeypatch):
    """tools/ref_on_gpu.py stage (what precedes the `gpurun` call that runs the reference's scripts on the MI355X): the scratch checkout is a COPY of the
    reference's files (nothing of ours mixed in, nothing edited), the generated assets are in the places the unchanged scripts open, and git ignores all of it."""
    import importlib.util
    import subprocess

    spec = importlib.util.spec_from_file_location("ref_on_gpu", os.path.join(ROOT, "tools", "ref_on_gpu.py"))
    tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tool)
    scratch = str(tmp_path / "_ref_scratch")
    monkeypatch.setattr(tool, "SCRATCH", scratch)
    monkeypatch.setattr(tool, "REF", os.path.join(scratch, "reference"))
    tool.stage(n_splats=12_000, n_timesteps=2, width=112, height=160)
    ref = os.path.join(scratch, "reference")
    for script in ("train.py", "render.py", "fps_benchmark_demo.py", "fps_benchmark_dataset.py", os.path.join("scene", "gaussian_model.py")):
        assert open(os.path.join(ref, script), "rb").read() == open(os.path.join(REF, script), "rb").read(), script      # byte for byte the reference's file
    for asset in ("flame2023.pkl", "FLAME_masks.pkl", "head_template_mesh.obj"):
        assert os.path.exists(os.path.join(ref, "flame_model", "assets", "flame", asset)), asset
    assert os.path.exists(os.path.join(scratch, "avatar", "point_cloud.ply")) and os.path.exists(os.path.join(scratch, "avatar", "flame_param.npz"))
    assert os.path.exists(os.path.join(scratch, "data", "transforms_train.json")) and os.path.exists(os.path.join(scratch, "STAGED.json"))
    # the real location is git-ignored (never committed): `git check-ignore` knows the rule
    r = subprocess.run(["git", "check-ignore", "-q", "_ref_scratch/reference/train.py"], cwd=ROOT)
    assert r.returncode == 0
