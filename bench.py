#!/usr/bin/env python3
"""bench.py -- frames/sec forward+backward of the GaussianAvatars hot path on MI355X.

One "step" = the reference's training inner loop on one frame (train.py:118-164 without optimiser /
SSIM, i.e. BASELINE.json config 3):
    select_mesh_by_timestep(t)  ->  render(cam, gaussians, pipe, white bg)  ->  l1_loss(image, white)
    ->  loss.backward()
on the synthetic stand-in for media/306 (100 000 mesh-bound SH-3 splats, 802x550, the camera of
fps_benchmark_demo.py:21-33).  Inputs are resident in HBM before the timed region.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode train|render] [--binding fused|unfused]

N > 1: one rank per GPU, every rank holds a replica of the splats and renders its own frames
(frame-parallel, weak scaling); the only collective is the all-reduce of the scalar loss (RCCL over
xGMI).  Rank 0 prints ONE JSON line.  Either an external `python -m torch.distributed.run
--nproc-per-node N bench.py --gpus N ...` provides the ranks (RANK / LOCAL_RANK / WORLD_SIZE in the
environment), or plain `python bench.py --gpus N` starts them itself (launch_ranks below: the same
torch.distributed.run command on 127.0.0.1, refused loudly when fewer than N GPUs are visible).

`--backend gloo` is the DRY path of the multi-rank plumbing for boxes without GPUs (tests/): the ranks,
the frame sharding, the run loop, the collectives and the JSON line are the real ones, the rasterizer's
autograd Function is replaced by a torch stub and the line says so in `data` -- never a measurement.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s is what a copy kernel reaches


class Pipe:
    debug = False
    compute_cov3D_python = False
    convert_SHs_python = False


SCENE = os.environ.get("GAA_BENCH_SCENE", "ellipsoid")   # --scene / GAA_BENCH_SCENE (tools that call build_scene directly): the head the bound splats sit on (synthetic.head_mesh): "ellipsoid" = rounds 1-5's stand-in (the default line, for continuity);
                      # "template_like" = the same topology with the reference template's face-area distribution (deeper tiles: VERDICT r05 Missing 2)
SPATIAL_SORT = True   # the splats in Morton order of their positions (gaussianavatars_amd.io.spatial_sort): what the package's loaders and the densification
                      # hook of patch.py produce by default since round 4 (GAA_SPATIAL_SORT=0 / --no-spatial-sort: the order the generator emits, random)


def build_scene(device, n_splats, sh_degree, width, height, n_frames, binding_impl, requires_grad):
    from gaussianavatars_amd import io as gio
    from gaussianavatars_amd import synthetic as S
    from gaussianavatars_amd.gaussian_model import FlameGaussianModel

    rig = S.flame_rig(seed=4, kind=SCENE)
    g = FlameGaussianModel(sh_degree, rig, binding_impl=binding_impl, device=device)
    arrs = S.bound_splats(n_splats, S.FLAME_F, sh_degree, seed=2)
    if SCENE == "template_like":
        arrs["_scaling"] = arrs["_scaling"] + np.float32(S.TEMPLATE_LIKE_LOG_SCALE_OFFSET)
    if SPATIAL_SORT:
        arrs = gio.spatial_sort(arrs, rig["v_template"][rig["faces"]].mean(1))
    g.load_arrays(arrs, device=device, requires_grad=requires_grad)
    g.load_flame_param(S.flame_sequence(n_frames, seed=4), device=device, requires_grad=requires_grad)
    cam = S.orbit_camera(width, height, r=1.0, fovy_deg=20.0)
    for k in ("world_view_transform", "full_proj_transform", "camera_center"):
        setattr(cam, k, torch.as_tensor(getattr(cam, k), device=device))
    return g, cam


def build_unbound_scene(device, n_splats, sh_degree, width, height):
    """BASELINE configs[4]: un-bound GaussianModel path (SURVEY.md 8(d) cfg 5), forward only."""
    import math

    from gaussianavatars_amd import synthetic as S
    from gaussianavatars_amd.gaussian_model import GaussianModel

    sp = S.random_splats(n_splats, sh_degree, 5, xyz_sigma=0.08, log_scale_mean=math.log(0.0015), log_scale_sigma=0.4)
    op = np.clip(sp["opacities"], 1e-6, 1 - 1e-6)
    arrs = dict(_xyz=sp["means3D"], _scaling=np.log(sp["scales"]), _rotation=sp["rotations"], _opacity=np.log(op / (1 - op)),
                _features_dc=sp["shs"][:, :1], _features_rest=sp["shs"][:, 1:])
    if SPATIAL_SORT:
        from gaussianavatars_amd import io as gio

        arrs = gio.spatial_sort(arrs)
    g = GaussianModel(sh_degree)
    g.load_arrays(arrs, device=device, requires_grad=False)
    cam = S.orbit_camera(width, height, r=1.0, fovy_deg=20.0)
    for k in ("world_view_transform", "full_proj_transform", "camera_center"):
        setattr(cam, k, torch.as_tensor(getattr(cam, k), device=device))
    return g, cam


SSIM_STEP = False   # --workload train: the loss and statistics lines of train.py:131-132,197-198 ride along


_ZERO = {}
# rocprofv3 kernel name -> the timing slot (include/gsr.h GSR_K_*) its launches are accounted under
PMC_ALIAS = {"k_render_bwd_rp": "k_render_bwd", "k_rcount": "k_count", "k_rscatter": "k_scatter", "k_rsort_rscatter": "k_scatter", "k_tile_rank": "k_tile_sort", "k_rdscatter": "k_depth_sort", "k_rdsort": "k_depth_sort",
             "k_band_count": "k_depth_sort", "k_band_scan": "k_depth_sort", "k_band_rank": "k_depth_sort",
             "k_dbucket": "k_depth_sort", "k_dscan": "k_depth_sort", "k_dscatter": "k_depth_sort", "k_dsort": "k_depth_sort", "k_qscan_glob": "k_qscan"}


def one_step(g, cam, bg, target, t, train):
    from gaussianavatars_amd.gaussian_renderer import l1_loss, render

    if g.binding is not None:
        g.select_mesh_by_timestep(t)
    pkg = render(cam, g, Pipe, bg)
    if not train:   # forward-only workloads: a constant device scalar for the (optional) all-reduce, no reduction kernels in the timed loop
        z = _ZERO.get(pkg["render"].device)
        if z is None:
            z = _ZERO[pkg["render"].device] = torch.zeros((), dtype=torch.float32, device=pkg["render"].device)
        return z
    if SSIM_STEP:
        from gaussianavatars_amd.loss import l1_ssim

        l1, ss = l1_ssim(pkg["render"], target)
        loss = 0.8 * l1 + 0.2 * (1.0 - ss)            # lambda_dssim = 0.2 (arguments/__init__.py)
        loss.backward()
        g.update_densification_stats(pkg["viewspace_points"], pkg["radii"])
        return loss.detach()
    loss = l1_loss(pkg["render"], target)
    loss.backward()
    return loss.detach()


def zero_grads(g):
    for p in (g._xyz, g._features_dc, g._features_rest, g._scaling, g._rotation, g._opacity):
        p.grad = None
    if getattr(g, "flame_param", None) is not None:
        for v in g.flame_param.values():
            if v.requires_grad:
                v.grad = None


def cpu_baseline(g, cam, bg, train, args, max_seconds=12.0):
    """The CPU oracle (oracle/, a port -- the reference has no CPU rasterizer) timed on this box's host cores on
    the same frame (rasterizer half only: world-space splats in, image and gradients out), once on every core (OpenMP over
    splats / pixel rows / tiles; the tile bucketing of the sort and the per-splat fold of the backward are shared by the cores as well)
    and once on ONE thread.  The first all-core frame is a warm-up (page faults of the 100+ MB intermediates) and is not timed.

    Runs in a CHILD process: this process spent its life pinned to one core complex next to its GPU, and the OpenMP pools torch and
    the oracle share were created under that mask -- 256 threads on 8 cores measured 0.49 frames/s against 0.46 on one thread, and the
    torch leg 28 s per frame.  The child starts with every core and fresh pools; it gets the frame's world-space splats as a file."""
    import math
    import subprocess
    import tempfile

    with torch.no_grad():
        if g.binding is not None:
            g.select_mesh_by_timestep(0)
        arrs = dict(means3D=g.get_xyz, shs=g.get_features, opacities=g.get_opacity, scales=g.get_scaling, rotations=g.get_rotation)
        arrs = {k: v.detach().float().cpu().numpy() for k, v in arrs.items()}
    arrs.update(bg=bg.cpu().numpy(), viewmatrix=cam.world_view_transform.cpu().numpy(), projmatrix=cam.full_proj_transform.cpu().numpy(),
                campos=cam.camera_center.cpu().numpy(),
                scalars=np.array([cam.image_height, cam.image_width, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), g.active_sh_degree,
                                  float(train), max_seconds, float(g.binding is not None), g.max_sh_degree,
                                  int(g.flame_param["expr"].shape[0]) if g.binding is not None else 0], np.float64))
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "frame.npz")
        np.savez(path, **arrs)
        env = {k: v for k, v in os.environ.items() if not k.startswith(("OMP_", "GOMP_", "KMP_", "MKL_")) and k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
        env["HIP_VISIBLE_DEVICES"] = ""     # the child is a host-only process
        env["OMP_WAIT_POLICY"] = "passive"   # idle team members sleep instead of spinning the container's CPU quota away
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-child", path], capture_output=True, text=True, timeout=900, env=env)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not lines:
        raise RuntimeError("cpu_baseline child failed: " + (r.stderr or r.stdout)[-2000:])
    out = json.loads(lines[-1])
    return out["cpu"], out["num_rendered"], out["visible"]


def host_cores():
    """Cores this process may really use: the affinity mask capped by the container's CPU quota (cgroup v2 cpu.max / v1 cfs_quota).  The MI355X
    boxes report 256 CPUs in the mask of a container that is throttled to a fraction of them: 256 OpenMP threads there were slower than one."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = min(n, max(1, -(-quota // period)))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline_child(path):
    """(child process of cpu_baseline) times the oracle, and the composed-torch binding half, on every core this host has."""
    from oracle import gsr_oracle as O

    O.build()
    a = dict(np.load(path))
    H, W, tfx, tfy, deg, train, max_seconds, bound, max_deg, n_frames = a["scalars"].tolist()
    train = bool(train)
    s = O.make_settings(int(H), int(W), tfx, tfy, a["bg"], 1.0, a["viewmatrix"], a["projmatrix"], int(deg), a["campos"])
    gpix = np.full((3, int(H), int(W)), -1.0 / (3 * H * W), np.float32)  # d l1(image, white)/d image where image < 1
    cores = host_cores()

    def timed(threads, max_frames, budget):
        used = O.set_threads(threads)
        frames, t0 = 0, time.perf_counter()
        while True:
            st = O.forward(s, a["means3D"], a["shs"], None, a["opacities"], a["scales"], a["rotations"], None)
            if train:
                O.backward(s, st, gpix)
            frames += 1
            el = time.perf_counter() - t0
            if el > budget or frames >= max_frames:
                return frames / el, frames, used, st

    timed(cores, 1, max_seconds)   # warm-up
    fps_all, n_all, used_all, st = timed(cores, 16, max_seconds)
    tried = {used_all: fps_all}
    for th in (32, 16, 8):          # a mask (or quota) larger than what the host really gives this container: a smaller team can be faster
        if th < cores and (th >= cores // 8 or fps_all < 4 * tried.get(1, 0) or len(tried) < 2):
            f, n, u, _ = timed(th, 4, max_seconds / 3)
            tried[u] = f
            if f > fps_all:
                fps_all, n_all, used_all = f, n, u
    fps_one, n_one, _, _ = timed(1, 2, max_seconds)
    tried[1] = fps_one
    O.set_threads(cores)
    what = "fwd+bwd" if train else "fwd"
    out = dict(value=fps_all, unit="frames/s", cores=used_all, kind="port", host_cores=cores, affinity_cpus=len(os.sched_getaffinity(0)),
               tried_frames_per_s={str(k): round(v, 3) for k, v in sorted(tried.items())},
               sample=f"{n_all} frame(s) of the bench workload, rasterizer half ({what}), oracle/gsr_oracle.c with OpenMP on {used_all} threads "
                      "(child process: every host core, fresh thread pools)",
               single_thread=dict(value=fps_one, unit="frames/s", cores=1, sample=f"{n_one} frame(s), same workload, one thread"))
    if bound:
        out["binding_half"] = cpu_binding_baseline(int(a["means3D"].shape[0]), int(max_deg), int(n_frames), train)
    print(json.dumps(dict(cpu=out, num_rendered=int(st.num_rendered), visible=int((st.radii > 0).sum()))))


def cpu_binding_baseline(n_splats, max_sh_degree, n_frames, train, frames=5):
    """The binding half on the host: the composed-torch formulation of the reference (select_mesh_by_timestep + the three
    bound accessors, gaussianavatars_amd/unfused.py) on torch-CPU.  These are ~200 small ATen ops per frame over
    (N,3) / (F,3,3) tensors: past one core complex more threads only add synchronisation (128 threads measured 60x slower than 8 on a
    256-core host), so the leg is timed at 8, 16 and all cores and the BEST is reported with its thread count."""
    cpu = torch.device("cpu")
    gc, _ = build_scene(cpu, n_splats, max_sh_degree, 64, 64, n_frames, "unfused", train)

    def frame(t):
        gc.select_mesh_by_timestep(t)
        x, s, r = gc.get_xyz, gc.get_scaling, gc.get_rotation
        if train:
            (x.sum() + s.sum() + r.sum()).backward()
            zero_grads(gc)

    cores = host_cores()
    before, tried = torch.get_num_threads(), {}
    try:
        for th in sorted({min(8, cores), min(16, cores), cores}):
            torch.set_num_threads(th)
            with torch.set_grad_enabled(train):
                frame(0)
                t0 = time.perf_counter()
                for i in range(frames):
                    frame(i % n_frames)
                tried[th] = 1e3 * (time.perf_counter() - t0) / frames
            if tried[th] > 4 * min(tried.values()):
                break   # (far past the knee: the remaining, larger counts are only slower)
    finally:
        torch.set_num_threads(before)
    best = min(tried, key=tried.get)
    return dict(ms_per_frame=round(tried[best], 2), threads=best, tried_ms_per_frame={str(k): round(v, 2) for k, v in tried.items()},
                what="composed-torch FLAME + face frames + per-splat bind on torch-CPU (" + ("fwd+bwd" if train else "fwd") + "), best thread count")


def lane_schedule(my_frames, lane, n_lanes):
    """The frames lane `lane` of `n_lanes` recorded lanes renders, in its own order: frame i of a run goes to lane i % n_lanes, so the lane's
    m-th frame is the run's frame lane + n_lanes * m (cyclic over the rank's frames, like the eager loop's `my_frames[i % len]`)."""
    return [my_frames[(lane + n_lanes * m) % len(my_frames)] for m in range(len(my_frames))]


def lane_plan(n, offset, n_lanes, frames_per_graph):
    """For a run of n steps that starts at the run-global frame index `offset`: per lane, the position in its schedule it starts from and the
    sizes of the recordings it replays (K-frame recordings while K frames are left, one-frame recordings for the rest)."""
    plan = []
    for j in range(n_lanes):
        count = len(range((j - offset) % n_lanes, n, n_lanes))
        plan.append(((offset + n_lanes - 1 - j) // n_lanes, [frames_per_graph] * (count // frames_per_graph) + [1] * (count % frames_per_graph)))
    return plan


def make_runner(step_fn, my_frames, dist, device, post_step=None):
    """run(n, offset): n steps over this rank's frames (wrapping around); returns the sum of the per-step scalars as a device
    tensor.  The scalar all-reduce of step k is issued asynchronously (RCCL runs it on its own stream) and only waited for after
    step k+1 has been enqueued, so its latency never idles the compute stream.  `step_fn(t) -> 0-d tensor`."""

    def run(n, offset):
        losses = []   # per-step device scalars; summed once at the end of the run (still inside the timed region)
        pending = None
        for i in range(n):
            t = my_frames[(offset + i) % len(my_frames)]
            l = step_fn(t)
            if dist is not None:
                buf = l.reshape(1).clone()
                work = dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=True)  # the one collective of the path: a scalar
                if pending is not None:
                    pending[0].wait()
                    losses.append(pending[1][0])
                pending = (work, buf)
            else:
                losses.append(l)
            if post_step is not None:
                post_step()
        if pending is not None:
            pending[0].wait()
            losses.append(pending[1][0])
        return torch.stack(losses).sum() if losses else torch.zeros((), device=device)

    return run


def timed_rounds(run, fence, steps, warmup, dist, device, min_rounds=3, min_seconds=0.5, max_rounds=64):
    """The reference's timing protocol (fps_benchmark_demo.py:53-66: rounds of n iterations, FPS per round): W untimed warm-up
    steps, then rounds of EXACTLY `steps` steps each, bracketed by barrier + device sync on both sides, the MAX over ranks
    taken per round.  At least `min_rounds` rounds and at least `min_seconds` of timed work in total, whatever `steps` is (the
    stopping rule reads the MAX-reduced times, so every rank runs the same number of rounds)."""
    run(warmup, 0)
    fence()
    rounds, total, offset = [], 0.0, warmup
    while len(rounds) < min_rounds or (total < min_seconds and len(rounds) < max_rounds):
        t0 = time.perf_counter()
        run(steps, offset)
        fence()
        el = time.perf_counter() - t0
        if dist is not None:
            te = torch.tensor([el], device=device, dtype=torch.float64)
            dist.all_reduce(te, op=dist.ReduceOp.MAX)
            el = float(te.item())
        rounds.append(el)
        total += el
        offset += steps
    return rounds


def _free_port():
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n, backend, argv):
    """`python bench.py --gpus N` without a launcher around it: N ranks of this same script under torch.distributed.run (one
    process per GPU, the reference forces `cuda:0` per process -- utils/general_utils.py:133 -- so a process never spans GPUs),
    rendezvous on 127.0.0.1.  Returns the launcher's exit code; rank 0's JSON line goes to this process's stdout unchanged."""
    import subprocess

    if backend == "nccl":
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < n:
            raise SystemExit(f"bench.py --gpus {n}: {have} GPU(s) visible on this node -- refusing to report an n_gpus={n} line from fewer devices")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, len(os.sched_getaffinity(0)) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


class _DryRasterize(torch.autograd.Function):
    """--backend gloo only: stands in for gaussianavatars_amd.rasterizer._RasterizeGaussians at the autograd-Function boundary (same ten
    arguments, same three outputs, gradients for the same inputs) so that the frame loop above it -- mesh update, accessors, render(),
    loss, backward, all-reduce -- runs on a box without a GPU.  The image is a smooth function of the inputs, nothing more."""

    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, rs, sh_rest=None):
        P = means3D.shape[0]
        ctx.shapes = (means2D.shape, sh.shape, None if sh_rest is None else sh_rest.shape)
        ctx.save_for_backward(means3D, opacities, scales, rotations)
        v = 0.5 * torch.tanh(means3D.mean() + opacities.mean() + scales.mean() + rotations.mean())
        radii = (torch.arange(P, dtype=torch.int32) % 3)
        ctx.mark_non_differentiable(radii)
        return v.expand(3, int(rs.image_height), int(rs.image_width)).clone(), radii, radii > 0

    @staticmethod
    def backward(ctx, g, *_):
        means3D, opacities, scales, rotations = ctx.saved_tensors
        v = torch.tanh(means3D.mean() + opacities.mean() + scales.mean() + rotations.mean())
        gs = 0.5 * (1 - v * v) * g.sum()
        m2, shs, rest = ctx.shapes
        return (gs.expand_as(means3D) / means3D.numel(), torch.zeros(m2), torch.zeros(shs) if shs.numel() else None, None,
                gs.expand_as(opacities) / opacities.numel(), gs.expand_as(scales) / scales.numel(), gs.expand_as(rotations) / rotations.numel(),
                None, None, None if rest is None else torch.zeros(rest))


def main():
    if len(sys.argv) == 3 and sys.argv[1] == "--cpu-baseline-child":
        return cpu_baseline_child(sys.argv[2])
    global SPATIAL_SORT, SCENE
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--mode", choices=["train", "render"], default="train")
    ap.add_argument("--binding", choices=["fused", "unfused"], default="fused")
    ap.add_argument("--splats", type=int, default=100_000)
    ap.add_argument("--width", type=int, default=550)
    ap.add_argument("--height", type=int, default=802)
    ap.add_argument("--frames", type=int, default=300)
    ap.add_argument("--workload", choices=["cfg3", "cfg2", "cfg4", "cfg5", "train"], default="cfg3",
                    help="BASELINE.json configs: cfg3 = configs[2] fwd+bwd 100k (the metric, default); cfg2 = configs[1] forward; "
                         "cfg4 = configs[3] 200k-splat 300-frame sequence fwd+bwd; cfg5 = configs[4] 2M-splat 1600x1100 forward stress")
    ap.add_argument("--scene", choices=["ellipsoid", "template_like"], default=SCENE,
                    help="bound workloads: the head mesh of the synthetic avatar.  ellipsoid: the stand-in of rounds 1-5 (the default line).  template_like: the same "
                         "topology with the face-area distribution and extent of the reference's head template (synthetic.head_mesh(kind=...): statistics only), "
                         "whose tiles are as deep as the avatar tools/ref_on_gpu.py stages on the real template; the default run reports it beside `value`")
    ap.add_argument("--no-template-like", action="store_true", help="default N=1 run: skip the template_like leg")
    ap.add_argument("--rounds", type=int, default=3, help="minimum number of timed rounds of --steps steps each (the median round is reported)")
    ap.add_argument("--min-seconds", type=float, default=3.0, help="minimum total timed duration: rounds are added until it is reached")
    ap.add_argument("--backend", choices=["nccl", "gloo", "gloo_gpu"], default="nccl",
                    help="nccl (= RCCL): the measured path.  gloo: DRY run of the multi-rank plumbing on CPU -- a few hundred splats, composed-torch "
                         "binding, the rasterizer stubbed at its autograd Function; the line is marked as such and is not a measurement.  gloo_gpu: the REAL step "
                         "(HIP libraries, compiled host) on N ranks that SHARE the visible GPU(s), collectives over gloo -- the multi-rank control flow of the measured "
                         "path (per-step all-reduce, barriers, rank-0-only passes) on a one-GPU box, where RCCL refuses two ranks on one device; marked, not a measurement")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-profile", action="store_true")
    ap.add_argument("--spatial-sort", action="store_true", help="(the default since round 4; kept so that older command lines still parse)")
    ap.add_argument("--no-spatial-sort", action="store_true",
                    help="keep the synthetic splats in the order the generator emits them (random) instead of the loaders' default, Morton order of "
                         "their positions (io.spatial_sort: same image, gradients permuted with the splats); reported in config.splat_order")
    ap.add_argument("--graph-frames", type=int, default=4,
                    help="with --graph: frames per recording (the frame feed is a kernel inside it: gab_feed_row), so that the launch gap between two "
                         "graphs is paid once per K frames; steps that do not fill a recording use a one-frame recording")
    ap.add_argument("--graph", action="store_true",
                    help="record the frame step once and replay it as one hipGraph launch per step (gaussianavatars_amd.graphs.GraphedStep)")
    ap.add_argument("--streams", type=int, default=1,
                    help="with --graph: this many recordings on this many streams, frames dealt to them in turn (frame parallelism inside one GPU)")
    ap.add_argument("--lane-models", choices=["shared", "replica"], default="shared",
                    help="with --streams > 1: `shared` (default) -- the lanes read ONE set of splat parameters (gaussianavatars_amd.graphs.shared_lane_model: "
                         "leaves over the same storage, a .grad of its own per lane; accumulate_lane_grads adds them up); `replica` -- a full model per lane "
                         "(round 3's arrangement)")
    ap.add_argument("--frame-streams", type=int, default=4,
                    help="N=1, eager default run only: after the timed rounds, the same workload again as this many recorded frame "
                         "lanes (reported beside `value` as `frame_streams`); 0 skips the leg")
    ap.add_argument("--no-pin", action="store_true", help="do not pin the process next to its GPU (host-sensitivity runs)")
    ap.add_argument("--dist-at-1", action="store_true",
                    help="with --gpus 1 under a launcher (WORLD_SIZE=1): initialise the process group anyway and send the per-step scalar through the "
                         "backend's all-reduce, so that a one-GPU box EXECUTES the RCCL path the N-GPU run takes (tests/test_rccl_gpu.py); the line "
                         "says so in config.parallelism.  Not the default: at N = 1 the reference-shaped run has no collective")
    args = ap.parse_args()
    SCENE = args.scene
    SPATIAL_SORT = not args.no_spatial_sort and os.environ.get("GAA_SPATIAL_SORT", "1") != "0"
    if args.workload == "cfg3" and args.mode == "render":
        args.workload = "cfg2"
    if args.workload == "train":   # not a BASELINE config: cfg3 plus the fused L1+SSIM loss and the densification statistics (N3)
        global SSIM_STEP
        SSIM_STEP = True
    if args.workload == "cfg2":
        args.mode = "render"
    elif args.workload == "cfg4":
        args.splats, args.frames = 200_000, 300
    elif args.workload == "cfg5":
        args.mode, args.splats, args.width, args.height = "render", 2_000_000, 1100, 1600   # SURVEY.md 8(d) cfg 5: H = 1600, W = 1100 (6900 tiles)

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:   # no launcher around this process: start the N ranks ourselves
        raise SystemExit(launch_ranks(args.gpus, args.backend, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} inside a launcher with WORLD_SIZE={world}: the two must agree "
                         "(python -m torch.distributed.run --nproc-per-node N bench.py --gpus N)")
    dry = args.backend == "gloo"
    shared_gpu = args.backend == "gloo_gpu"
    if dry:   # plumbing check without a GPU: tiny scene, reference-shaped composed-torch binding, stubbed rasterizer Function
        device = torch.device("cpu")
        args.splats, args.width, args.height, args.frames = min(args.splats, 12_000), 64, 48, min(args.frames, 16)
        args.binding, args.no_cpu_baseline, args.no_kernel_profile, args.frame_streams, args.no_pin = "unfused", True, True, 0, True
        if args.graph or args.workload == "cfg5":
            raise SystemExit("--backend gloo covers the eager frame loop of the bound workloads only (--streams S without --graph: the lanes as plain step functions)")
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: there is no CPU path for the rasterizer (--backend gloo is the dry run of the multi-rank plumbing)")
        if shared_gpu:     # control-flow run: the ranks share whatever is visible
            local_rank = local_rank % torch.cuda.device_count()
        if torch.cuda.device_count() <= local_rank:
            raise SystemExit(f"rank {rank}: LOCAL_RANK={local_rank} but {torch.cuda.device_count()} GPU(s) visible")
        torch.cuda.set_device(local_rank)
        device = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or (args.dist_at_1 and "RANK" in os.environ):
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if dry or shared_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    n_gpus = dist.get_world_size() if dist is not None else 1
    train = args.mode == "train"

    from gaussianavatars_amd import _lib
    from gaussianavatars_amd import rasterizer as R
    from gaussianavatars_amd.frame_parallel import frames_for_rank, pin_to_gpu_numa_node

    all_cpus = os.sched_getaffinity(0)
    pinned = None if args.no_pin else pin_to_gpu_numa_node(local_rank)   # host launch latency: stay on the GPU's socket
    seeded = os.environ.get("GAA_LOSS_SEED", "1") != "0"
    if seeded:   # as patch_reference() does for train.py: `loss.backward()` is seeded with a cached device 1 (no one-element fill launch)
        from gaussianavatars_amd.loss import install_backward_seed

        install_backward_seed()
    if args.workload == "cfg5":
        g, cam = build_unbound_scene(device, args.splats, 3, args.width, args.height)
    else:
        g, cam = build_scene(device, args.splats, 3, args.width, args.height, args.frames, args.binding, train)
    bg = torch.ones(3, dtype=torch.float32, device=device)
    target = torch.ones((3, args.height, args.width), dtype=torch.float32, device=device)
    my_frames = frames_for_rank(args.frames, rank, world)

    if dry:
        R._RasterizeGaussians.apply = staticmethod(_DryRasterize.apply)
        import gaussianavatars_amd.gaussian_renderer as GR

        GR.l1_loss = lambda a, b: (a - b).abs().mean()   # one_step imports it from there

    def step_fn(t):
        with torch.set_grad_enabled(train):
            return one_step(g, cam, bg, target, t, train)

    run = make_runner(step_fn, my_frames, dist, device, post_step=(lambda: zero_grads(g)) if train else None)
    run_eager, graphed, lanes = run, None, []
    # the instrumented pass further down runs on rank 0 ONLY: its runner must not contain the per-step all-reduce (a collective entered by one rank of N pairs
    # with whatever the other ranks enter next -- their barrier -- and the run hangs; found in round 5, the N > 1 path on RCCL had never executed before)
    run_profile = make_runner(step_fn, my_frames, None, device, post_step=(lambda: zero_grads(g)) if train else None)
    if args.streams > 1 and not args.graph and not dry:
        raise SystemExit("--streams needs --graph (two eager frame loops in one process are host-bound)")
    # (round 4) lanes x ranks: every rank deals ITS frames to its --streams recorded lanes; the lanes add their losses up on the device and the
    # rank contributes the sum to ONE scalar all-reduce per run of K steps (the path's only collective).  `--backend gloo
    # --streams S` is the dry run of that plumbing: the lanes are plain step functions called in turn (no hipGraph, no streams on a CPU).

    def build_lanes(n_lanes):
        """The step recorded once per lane and replayed (gaussianavatars_amd.graphs): the frame's FLAME parameters are fed into
        static one-row tables (one copy), the gradients land in static .grad tensors (no zero_grads: every replay rewrites
        them).  More than one lane: one recording per stream, each with a replica of the model (its own static inputs and
        gradients, as a second rank would have); consecutive frames go to the lanes in turn, so independent frames overlap
        on the GPU the way they do across GPUs."""
        from gaussianavatars_amd.graphs import FlameRowFeeder, GraphedStep, release_mesh

        import copy

        base_model = copy.copy(g)   # (lane 0 swaps g.flame_param for its static rows below: the other lanes clone the full tables from this handle)

        class _EagerLane:   # --backend gloo: a "recording" that simply runs (the dry run of the lanes x ranks plumbing)
            def __init__(self, fn, before):
                self.fn, self.before = fn, before

            def replay(self):
                self.before()
                return self.fn()

            def check(self):
                pass

            def instances(self):
                return [0]

            capacity = 0

        def make_lane(gm, lane, n_lanes):
            feeder = None
            if gm.binding is not None:
                feeder = FlameRowFeeder(gm.flame_param, requires_grad=train)
                gm.flame_param = feeder.static_param
                # this lane's frames in the order the eager loop would hand them over: frame i of the run goes to lane i % n_lanes
                feeder.set_schedule(lane_schedule(my_frames, lane, n_lanes))
            loss_sum = torch.zeros((), dtype=torch.float32, device=device)

            def fixed_step():   # the eager per-kernel event pass: one frame, fed by the caller
                with torch.set_grad_enabled(train):
                    return one_step(gm, cam, bg, target, 0, train)

            def frames_step(k):
                def fn():   # K frames back to back, each fed from the device-side schedule; what a frame loop does between frames included
                    ls = []
                    for j in range(k):
                        if feeder is not None:
                            feeder.feed_next()
                        with torch.set_grad_enabled(train):
                            ls.append(one_step(gm, cam, bg, target, 0, train))
                        if train and j + 1 < k:
                            zero_grads(gm)
                            release_mesh(gm)
                    l = ls[0] if k == 1 else torch.stack(ls).sum()
                    if (dist is None or n_lanes > 1) and train:
                        loss_sum.add_(l)
                    return l
                return fn

            def fresh():   # no gradients and no autograd graph of an earlier frame when the step is recorded
                zero_grads(gm)
                release_mesh(gm)

            K = max(1, int(args.graph_frames)) if (dist is None or n_lanes > 1) else 1   # (one lane per rank: every step's scalar is all-reduced, one frame per recording)
            Rec = (lambda fn: _EagerLane(fn, fresh)) if dry else (lambda fn: GraphedStep(fn, before_capture=fresh))
            ln = dict(g=gm, feeder=feeder, loss_sum=loss_sum, fixed_step=fixed_step, stream=None if dry else torch.cuda.Stream(device), K=K,
                      graphed=Rec(frames_step(1)))
            ln["graphed_k"] = Rec(frames_step(K)) if K > 1 else ln["graphed"]
            return ln

        out = [make_lane(g, 0, n_lanes)]
        for lane in range(1, n_lanes):
            if args.lane_models == "shared":   # (round 4) one set of splats per GPU: the lane's leaves alias lane 0's storage, its gradients are its own
                from gaussianavatars_amd.graphs import shared_lane_model

                gk = shared_lane_model(base_model)
            elif args.workload == "cfg5":
                gk, _ = build_unbound_scene(device, args.splats, 3, args.width, args.height)
            else:
                gk, _ = build_scene(device, args.splats, 3, args.width, args.height, args.frames, args.binding, train)
            out.append(make_lane(gk, lane, n_lanes))
        return out

    def lane_runner(lanes):
        import contextlib

        on = (lambda ln: contextlib.nullcontext()) if dry else (lambda ln: torch.cuda.stream(ln["stream"]))

        def run(n, offset):   # the recordings add their losses to a static accumulator, nothing else runs per step
            cur = None if dry else torch.cuda.current_stream(device)
            plan = lane_plan(n, offset, len(lanes), lanes[0]["K"])   # frame i of the run belongs to lane (offset + i) % L
            for ln, (start, _) in zip(lanes, plan):
                if cur is not None:
                    ln["stream"].wait_stream(cur)
                with on(ln):
                    ln["loss_sum"].zero_()
                    if ln["feeder"] is not None:
                        ln["feeder"].seek(start)
            todo = [list(sizes) for _, sizes in plan]
            while any(todo):
                for ln, td in zip(lanes, todo):
                    if td:
                        k = td.pop(0)
                        with on(ln):
                            (ln["graphed_k"] if k > 1 else ln["graphed"]).replay()
            if cur is not None:
                for ln in lanes:
                    cur.wait_stream(ln["stream"])
            total = torch.stack([ln["loss_sum"] for ln in lanes]).sum()
            if dist is not None:   # lanes x ranks: ONE scalar all-reduce per run, of what this rank's lanes added up
                total = total.reshape(1)
                dist.all_reduce(total, op=dist.ReduceOp.SUM)
                total = total[0]
            return total

        return run

    if args.graph or (dry and args.streams > 1):
        lanes = build_lanes(args.streams)
        graphed = lanes[0]["graphed"]

        def eager_fed(t):   # the per-kernel event pass stays eager (events are recorded around the launches)
            if lanes[0]["feeder"] is not None:
                lanes[0]["feeder"].feed(t)
            return lanes[0]["fixed_step"]()

        run_eager = make_runner(eager_fed, my_frames, dist, device, post_step=(lambda: zero_grads(g)) if train else None)
        run_profile = make_runner(eager_fed, my_frames, None, device, post_step=(lambda: zero_grads(g)) if train else None)

        def graph_step(t):
            if lanes[0]["feeder"] is not None:
                lanes[0]["feeder"].seek(my_frames.index(t) if t in my_frames else 0)
            return graphed.replay().clone()   # the recorded scalar is overwritten by the next replay

        # one lane per rank: a replay per step and the per-step asynchronous all-reduce of its scalar; otherwise the lanes run on their own
        run = make_runner(graph_step, my_frames, dist, device) if (dist is not None and len(lanes) == 1) else lane_runner(lanes)

    def fence():
        if dist is not None:
            dist.barrier()
        if not dry:
            torch.cuda.synchronize(device)

    _lib.gsr_wait_stats()   # reset
    rounds = timed_rounds(run, fence, args.steps, args.warmup, dist, device, min_rounds=args.rounds, min_seconds=args.min_seconds, max_rounds=100_000)   # (--min-seconds is the stopping rule; round 3 capped it at 256 rounds)
    elapsed = float(np.median(rounds))   # the median round: exactly args.steps steps
    wait_ms, waits = _lib.gsr_wait_stats()
    info = {"binning_path": 0} if dry else R.last_forward_info()
    if graphed is not None:
        for ln in lanes:
            ln["graphed_k"].check()
            ln["graphed"].check()                         # the device marks a slot whose frame did not fit and only the host clears the mark: this covers every replay above
        info["num_rendered"] = max(max(graphed.instances()), max(lanes[0]["graphed_k"].instances()))   # (the recording itself did not know its count)

    # ---- per-kernel durations: HIP events on the launch stream (recorded by the C ABI around every kernel).
    # Bracketing each launch with an event pair costs ~5 % of the frame rate (983 -> 933 frames/s measured),
    # so `value` above comes from the un-instrumented pass and the SAME K steps are repeated here with events on.
    kern, kern_other = {}, {}
    if rank == 0 and not args.no_kernel_profile:
        _lib.gsr_profile_enable(True)
        _lib.launch_profile_enable(True)     # libgab / libgls: the binding and loss kernels of the step (round 5: the whole step in the roofline record)
        run_profile(args.steps, args.warmup)
        torch.cuda.synchronize(device)
        kern = _lib.gsr_profile_read()
        kern_other = _lib.launch_profile_read()
        _lib.gsr_profile_enable(False)
        _lib.launch_profile_enable(False)
    # ---- `--workload train`: the fused L1+SSIM pair has no timing slot in the rasterizer's C ABI; its three launches (forward, reduce, backward)
    # are timed here with events on the launch stream, on one image pair, for the roofline line of the loss
    loss_line = None
    if rank == 0 and SSIM_STEP and not args.no_kernel_profile:
        from gaussianavatars_amd.loss import l1_ssim

        img = torch.rand((3, args.height, args.width), device=device).requires_grad_(True)
        reps, a, b = 50, torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for k in range(reps + 10):
            if k == 10:
                a.record()
            l1, ss = l1_ssim(img, target)
            (0.8 * l1 + 0.2 * (1.0 - ss)).backward()
            img.grad = None
        b.record()
        torch.cuda.synchronize(device)
        us = 1e3 * a.elapsed_time(b) / reps
        hw = args.width * args.height
        # forward: two images in (24 HW), three derivative planes per channel out (36 HW); backward: the planes in (36 HW), the gradient out (12 HW)
        algo_loss = 108 * hw
        loss_line = dict(kernels="gls::k_l1_ssim_fwd + k_l1_ssim_reduce + k_l1_ssim_bwd (+ the torch scalar ops of 0.8 L1 + 0.2 (1 - SSIM))", avg_us=round(us, 2),
                         algorithmic_bytes=int(algo_loss), algo_GBs=round(algo_loss / (us * 1e-6) / 1e9, 1), frac=round(algo_loss / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                         note="three launches + four elementwise torch kernels for 48 MB: launch-latency bound, HBM fraction reported as asked")
    if dist is not None:
        dist.barrier()

    # ---- frame parallelism INSIDE the GPU (reported beside `value`, never as `value`): the same workload as --frame-streams
    # recorded lanes on as many streams.  Independent frames -- what the ranks of a multi-GPU run process side by side -- overlap
    # on one GPU too: one frame's serial tails and 20-workgroup kernels leave most of the 256 CUs idle.
    frame_streams = None
    if rank == 0 and world == 1 and not args.graph and args.frame_streams > 1:
        # in a process of its own: whatever happens to the recording cannot take the line above with it
        import subprocess

        cmd = [sys.executable, os.path.abspath(__file__), "--graph", "--streams", str(args.frame_streams), "--frame-streams", "0",
               "--no-cpu-baseline", "--no-kernel-profile", "--workload", args.workload, "--steps", str(args.steps), "--warmup", str(args.warmup),
               "--rounds", str(args.rounds), "--min-seconds", str(min(args.min_seconds, 1.5)), "--splats", str(args.splats), "--width", str(args.width),
               "--height", str(args.height), "--frames", str(args.frames), "--binding", args.binding] + (["--no-pin"] if args.no_pin else []) + (
                   ([] if SPATIAL_SORT else ["--no-spatial-sort"]))
        try:
            out = subprocess.run(cmd, capture_output=True, text=True, timeout=240)
            sub = json.loads(out.stdout.strip().splitlines()[-1])
            frame_streams = {"streams": args.frame_streams, "value": sub["value"], "unit": sub["unit"], "ms_per_step": sub["ms_per_step"],
                             "rounds": sub["rounds"]["n"],
                             "what": "the same steps dealt in turn to recorded (hipGraph) lanes on separate streams, the lanes sharing one set of splats "
                                     "(`bench.py --graph --streams %d`, run as a child process after the timed rounds above); "
                                     "ms_per_step = elapsed / steps, not the latency of one frame" % args.frame_streams}
        except Exception as e:   # noqa: BLE001 -- the leg is extra evidence, never a reason to lose the line
            frame_streams = {"streams": args.frame_streams, "error": "%s: %s" % (type(e).__name__, str(e)[:200])}
        # the same step recorded once and replayed as ONE hipGraph launch per four frames on ONE stream (`bench.py --graph`): what the GPU side of
        # the step costs when the host is out of the way -- the eager loop above (`value`: one Python-driven launch per kernel, what an unchanged
        # train.py does) is bound by its ~300 us of host work per step since round 4, not by the 283 us of kernels
        if train:
            cmd1 = [c for c in cmd]
            i_s = cmd1.index("--streams")
            del cmd1[i_s:i_s + 2]
            try:
                out = subprocess.run(cmd1, capture_output=True, text=True, timeout=240)
                sub = json.loads(out.stdout.strip().splitlines()[-1])
                frame_streams["recorded_step"] = {"value": sub["value"], "unit": sub["unit"], "ms_per_step": sub["ms_per_step"], "min": sub["rounds"]["min"],
                                                  "what": "one recorded lane (`bench.py --graph`): the same frames, one hipGraph launch per four frames"}
            except Exception as e:   # noqa: BLE001
                frame_streams["recorded_step"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}

    # ---- the same workload on the template-like head (VERDICT r05 Missing 2): a child process with the per-kernel event pass on, reported beside `value`
    template_like = None
    if rank == 0 and world == 1 and not args.graph and not dry and args.scene == "ellipsoid" and args.workload != "cfg5" and not args.no_template_like:
        import subprocess

        cmd = [sys.executable, os.path.abspath(__file__), "--scene", "template_like", "--frame-streams", "0", "--no-cpu-baseline", "--workload", args.workload,
               "--steps", str(args.steps), "--warmup", str(args.warmup), "--rounds", str(args.rounds), "--min-seconds", str(min(args.min_seconds, 1.5)),
               "--splats", str(args.splats), "--width", str(args.width), "--height", str(args.height), "--frames", str(args.frames), "--binding", args.binding] + (
                   ["--no-pin"] if args.no_pin else []) + ([] if SPATIAL_SORT else ["--no-spatial-sort"])
        try:
            out = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
            sub = json.loads(out.stdout.strip().splitlines()[-1])
            ak = (sub.get("roofline") or {}).get("all_kernels", {})
            template_like = {"value": sub["value"], "unit": sub["unit"], "ms_per_step": sub["ms_per_step"], "min": sub["rounds"]["min"],
                             "k_render_us": (ak.get("k_render") or {}).get("avg_us"), "k_render_bwd_us": (ak.get("k_render_bwd") or {}).get("avg_us"),
                             "kernel_sum_us": ((sub.get("roofline") or {}).get("step") or {}).get("kernel_sum_us"),
                             "num_rendered": sub["config"].get("num_rendered"), "num_binned": sub["config"].get("num_binned"),
                             "what": "`bench.py --scene template_like`: the same step on splats bound to a head with the reference template's face-area "
                                     "distribution (synthetic.head_mesh(kind='template_like')), run as a child process after the timed rounds above"}
        except Exception as e:   # noqa: BLE001 -- extra evidence, never a reason to lose the line
            template_like = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}

    if rank == 0:
        N, HW = args.splats, args.width * args.height
        I_binned = info.get("num_rendered", 0)            # instances actually binned (tile culling on: the culled lists)
        I_rect = info.get("rect_instances", I_binned)     # the reference's count: sum of tiles_touched
        with torch.no_grad():
            vis = 1.0
        per_kernel = {k: dict(avg_us=1e3 * ms / max(n, 1), launches=n) for k, (ms, n) in kern.items() if n}
        # Algorithmic bytes per launch: SURVEY.md 8(d)'s per-unit figures x the units of one frame; v (visible
        # fraction) = 1 for this scene (measured by the oracle leg below).  The kernels downstream of the binning stream the CULLED
        # instance lists, so their unit count is I_binned; the same figures with the reference's rect-based count are kept beside
        # them (`algorithmic_bytes_rect_based`) -- those are what an implementation without tile culling would have to move.
        path = int(info.get("binning_path", 2))           # 0 rank path, 1 depth-ordered scatter, 2 round 1's per-tile sort
        bands = int(info.get("rank_bands", 1))            # rank path: bands of tile rows with a depth rank of their own (1 up to 262144 splats)
        def algo_for(I):
            a = {
                "k_preprocess": 236 * N + (44 + (27 if train else 0)) * N,
                "k_scatter": 12 * I,
                "k_tile_sort": 12 * I + 8 * I,              # one ideal sort pass read + range scan (the 12 I key/idx write is k_scatter's)
                "k_render": 40 * I + (12 + (8 if train else 0)) * HW,
                "k_render_bwd": 20 * HW + 40 * I + 36 * N,
                "k_preprocess_bwd": 300 * N + 256 * N,
            }
            if path == 0:   # rank path: bytes per splat N / per tile instance I of each pass
                a.update({   # (k_preprocess stays at SURVEY.md 8(d)'s design-independent 236 N + 44 vN (+27 vN): the rank path's own rows -- binned
                    # rect 8 B, quadrant-test operands 32 B per splat -- are traffic of this design, they show up in `traffic`, not here)
                    "k_count": 16 * N,                             # rect 8, tiles_touched 4, depth 4
                    "k_depth_sort": 20 * N + 12 * N,               # bucket scatter: rect 8 + depth 4 read, key 8 written; bucket sort: key 8 read, rank 4 written
                    "k_scatter": 44 * N + 8 * I,                   # rect 8 + rank 4 + operands 32 read per splat, one 8-byte entry written per instance
                    "k_tile_sort": 8 * I + 4 * I,                  # entry read; at least one 4-byte stream entry written per instance
                })
                if bands > 1:   # frames beyond 262144 splats: ranks per band of tile rows (k_band_count / k_band_scan / k_band_rank)
                    a.update({
                        "k_depth_sort": 20 * N + 24 * N + 8 * N + 24 * N,   # bucket sort: key 8 + rect 8 read, (splat, bands) 8 written; count pass: 8 read; rank pass: 8 read, four ranks 16 written
                        "k_scatter": 56 * N + 8 * I,                         # rect 8 + four ranks 16 + operands 32 read per splat
                    })
            return a
        algo, algo_rect = algo_for(I_binned), algo_for(I_rect)
        # ---- the binding (libgab) and loss (libgls) kernels of the step: SURVEY.md 8(d)'s "Binding (per frame)" row split by kernel -- the FLAME tables
        # a kernel has to read once (expression block of the blend shapes 3V x n_expr, pose-corrective block 36 x 3V, skinning weights), the per-face
        # table (17 floats per face) and, for the bound rasterizer entry's face reduction, the 20-float row per splat; the per-splat 44 B in / 40 B out of
        # the bind itself are inside k_preprocess / k_preprocess_bwd on this path (N1).  Names as the launch sites spell them.
        fm = getattr(g, "flame_model", None)
        if fm is not None:
            V_, F_, NE_ = int(fm.v_template.shape[0]), int(fm.faces.shape[0]), int(fm.shapedirs.shape[2]) - 300
            T_ = int(g.flame_param["expr"].shape[0])
            fbytes = 3 * F_ * fm.faces.element_size()
            algo.update({
                "gab::k_flame_fused": 4 * (3 * V_ * NE_ + 36 * 3 * V_ + 3 * V_ + 5 * V_ + 6 * V_),
                "gab::k_face_frames": 4 * 3 * V_ + fbytes + 4 * 17 * F_,
                "gab::k_bind_bwd_faces": 4 * 20 * N + 4 * (F_ + 1) + 4 * 17 * F_,
                "gab::k_gather_skin_bwd": 16 * 3 * F_ + 4 * (V_ + 1) + 4 * 3 * V_ + 4 * 17 * F_ + 4 * 36 * 3 * V_ + 4 * 5 * V_ + 4 * 6 * V_ + 4 * T_ * (NE_ + 18),
                "gab::k_chain_blend_bwd": 4 * (3 * V_ * NE_ + 3 * V_),
                "gab::k_bind": 84 * N, "gab::k_bind_bwd_rows": (84 + 40 + 80) * N,
            })
        algo.update({"gls::k_l1_fwd": 4 * 3 * HW * 3, "gls::k_l1_reduce": 8 * ((3 * HW + 1023) // 1024), "gls::k_l1_bwd": 4 * 3 * HW * 3,
                     "gls::k_l1_ssim_fwd": 60 * HW, "gls::k_l1_ssim_bwd": 48 * HW, "gls::k_reduce_partials": 8 * 51 * 18 * 3,
                     "gls::k_densify_stats": 28 * N})
        for name, (ms, n) in kern_other.items():
            if n:
                short = name.replace("void ", "").split("<")[0]
                if short.startswith("gls::k_l1_fwd"):
                    short = "gls::k_l1_fwd"   # (the launch sites' plain names of its four instances: one launch / two launches, with / without the gradient image)
                prev = per_kernel.get(short)
                tot_ms, tot_n = ms + (prev["avg_us"] * prev["launches"] / 1e3 if prev else 0.0), n + (prev["launches"] if prev else 0)
                per_kernel[short] = dict(avg_us=1e3 * tot_ms / tot_n, launches=tot_n)
        # Coalesced-read component of each kernel (bytes per launch), for the PMC calibration rule of profiles/r02_pmc_calibration.json:
        # FETCH_SIZE tallies 64 B per request; coalesced streams issue 128-byte requests (reported at half), per-lane gathers and
        # scalar loads issue 64-byte ones (face value).  The blend kernels' 2-D tile accesses (32-128 B runs) lie between the
        # calibrated patterns and are taken at face value (a lower bound); `traffic_bounds` brackets every kernel.
        stream = {"k_preprocess": 236 * N, "k_count": 44 * N, "k_scatter": 48 * N, "k_tile_sort": 8 * I_binned, "k_render": 0,
                  "k_render_bwd": 0, "k_preprocess_bwd": 300 * N + 48 * N}
        if path == 0:
            stream.update({"k_count": 16 * N, "k_depth_sort": 20 * N, "k_scatter": 44 * N, "k_tile_sort": 8 * I_binned})
        pmc, pmc_raw = {}, {}
        pmc_tag = {"cfg5": "cfg5", "cfg4": "cfg4", "cfg3": "cfg3", "cfg2": "cfg3", "train": "cfg3"}.get(args.workload)   # (train: cfg3's rasterizer kernels)
        import glob
        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{pmc_tag}_pmc_fetch_write_per_launch.json"))) if pmc_tag else []
        pmc_path = cands[-1] if cands else ""   # the newest committed PMC summary of this workload
        pmc_stale = None
        if pmc_path:   # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command (profiles/README.md)
            pmc_json = json.load(open(pmc_path))
            meta = pmc_json.pop("_meta", None)   # tools/pmc_per_launch.py: commit, library digests, box
            if meta is None:
                pmc_stale = "no provenance in the PMC file (collected before round 4)"
            else:
                import hashlib

                from gaussianavatars_amd import _lib as _L
                have = hashlib.sha256(open(_L.GSR_LIB_PATH, "rb").read()).hexdigest()[:16]
                if meta.get("libraries", {}).get("libgsr_hip.so") != have:
                    pmc_stale = f"libgsr_hip.so changed since the PMC passes (commit {meta.get('commit') or '?'}): re-run tools/collect_profiles.sh"
            if pmc_stale and rank == 0:
                print(f"bench.py: roofline.traffic comes from {os.path.basename(pmc_path)} -- {pmc_stale}", file=sys.stderr)
            for k, v in pmc_json.items():
                base = k.replace("void ", "").split("<")[0].split("(")[0]
                name = base.split("::")[-1]                                  # the k_tile_sort classes add up
                if base.startswith(("gab::", "gls::")):
                    name = base                                               # the binding / loss kernels keep their library prefix (all_kernels' keys)
                else:
                    name = PMC_ALIAS.get(name, name)                          # kernels that share one timing slot add up as well
                f, w = pmc_raw.get(name, (0.0, 0.0))
                pmc_raw[name] = (f + 1024.0 * v["FETCH_SIZE_KB_per_launch"], w + 1024.0 * v["WRITE_SIZE_KB_per_launch"])
            for name, (f, w) in pmc_raw.items():
                pmc[name] = int(f + min(f, 0.5 * stream.get(name, 0)) + w)
        roofline = None
        if per_kernel:
            dom = max((k for k in per_kernel if k in algo), key=lambda k: per_kernel[k]["avg_us"] * per_kernel[k]["launches"])
            ach = algo[dom] / (per_kernel[dom]["avg_us"] * 1e-6) / 1e9
            notes = {"k_tile_sort": ("per-tile latency chain (bitmap, scans, compaction: six barriers), not streaming bandwidth; HBM fraction reported as asked"
                                     if path == 0 else
                                     "bound by its sorting network (instruction issue) and the scattered record gathers of its epilogue, not by "
                                     "streaming bandwidth; HBM fraction reported as asked"),
                     "k_render": "alpha-blend kernels are issue-bound (exp + per-wave instruction stream), HBM fraction reported as asked",
                     "k_render_bwd": "alpha-blend kernels are issue-bound (exp + per-wave instruction stream), HBM fraction reported as asked"}
            # the whole step (round 5): algorithmic bytes of every kernel of one step / the step's wall time, and the sum of the kernels' own durations
            per_step = lambda v: v["launches"] / float(args.steps)
            # (a libgsr timing slot may cover several launches of one frame -- k_depth_sort: the bucket scatter and the sorts -- its bytes are per FRAME)
            step_bytes = sum(algo[k] * (per_step(v) if "::" in k else 1.0) for k, v in per_kernel.items() if k in algo)
            kernel_sum_us = sum(v["avg_us"] * per_step(v) for v in per_kernel.values())
            ms_step = 1e3 * elapsed / args.steps
            step_line = dict(algorithmic_bytes=int(step_bytes), ms_per_step=round(ms_step, 4), achieved=round(step_bytes / (ms_step * 1e-3) / 1e9, 1), unit="GB/s",
                             frac=round(step_bytes / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), kernel_sum_us=round(kernel_sum_us, 1),
                             launches_per_step=round(sum(per_step(v) for v in per_kernel.values()), 2),
                             unaccounted_kernels=sorted(k for k in per_kernel if k not in algo),
                             what="sum over every kernel of one step (libgsr + libgab + libgls) of its SURVEY.md 8(d) bytes, divided by ms_per_step of the timed "
                                  "(un-instrumented) rounds; kernel_sum_us = the kernels' own event-timed durations added up (instrumented pass)")
            roofline = dict(kernel=dom, bound="hbm", achieved=round(ach, 2), peak=HBM_PEAK_GBS, unit="GB/s", step=step_line,
                            frac=round(ach / HBM_PEAK_GBS, 5), traffic=pmc.get(dom),
                            traffic_bounds=[int(sum(pmc_raw[dom])), int(2 * pmc_raw[dom][0] + pmc_raw[dom][1])] if dom in pmc_raw else None,
                            traffic_source=os.path.basename(pmc_path) if pmc_path else None, traffic_stale=pmc_stale,
                            algorithmic_bytes_per_launch=int(algo[dom]), algorithmic_bytes_rect_based=int(algo_rect[dom]),
                            instances=dict(binned=int(I_binned), rect_based=int(I_rect)),
                            avg_launch_us=round(per_kernel[dom]["avg_us"], 2), note=notes.get(dom, "HBM-bound streaming kernel"),
                            all_kernels={k: dict(avg_us=round(v["avg_us"], 2), launches_per_step=round(v["launches"] / float(args.steps), 2),
                                                 algo_GBs=round(algo[k] / (v["avg_us"] * 1e-6) / 1e9, 1) if k in algo else None,
                                                 frac=round(algo[k] / (v["avg_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4) if k in algo else None,
                                                 traffic=pmc.get(k))
                                         for k, v in per_kernel.items()})
        cpu = None
        if not args.no_cpu_baseline and world == 1:   # rank 0 at N=1 only
            os.sched_setaffinity(0, all_cpus)   # the CPU legs get every host core back (the frame loop was pinned to 8)
            cpu, I_cpu, vis_cpu = cpu_baseline(g, cam, bg, train, args)
            vis = vis_cpu / N
        fps = n_gpus * args.steps / elapsed
        counts = [len(frames_for_rank(args.frames, r, world)) for r in range(world)]
        out = {
            "metric": ("frames/sec fwd+bwd" if train else "frames/sec fwd") + (" (L1+SSIM loss, densification stats)" if SSIM_STEP else "") +
                      " @%dk SH-3 splats %dx%d" % (N // 1000, args.height, args.width),
            "value": round(fps, 2),
            "unit": "frames/s",
            "n_gpus": n_gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": ("synthetic" if not shared_gpu else "synthetic; NOT A MEASUREMENT (--backend gloo_gpu): %d ranks share %d GPU(s), collectives over gloo -- the multi-rank "
                                                        "control flow of the measured path on real kernels" % (world, torch.cuda.device_count())) if not dry else "DRY RUN (--backend gloo, CPU): rasterizer stubbed at its autograd Function, composed-torch binding, "
                                                "reduced scene -- checks the multi-rank plumbing, measures nothing",
            "config": {
                "workload": {"cfg3": "BASELINE configs[2]: %d mesh-bound SH-3 splats (synthetic stand-in for media/306), %dx%d (HxW), "
                                     "select_mesh_by_timestep + render + L1-vs-white + backward, no optimiser step",
                             "cfg2": "BASELINE configs[1]: %d mesh-bound SH-3 splats, %dx%d (HxW), select_mesh_by_timestep + render, no_grad",
                             "cfg4": "BASELINE configs[3]: %d splats bound to the 5143-vertex synthetic FLAME rig, 300-frame expression "
                                     "sequence, %dx%d (HxW), fwd+bwd, frames sharded over the ranks",
                             "cfg5": "BASELINE configs[4]: %d un-bound SH-3 splats, %dx%d (HxW), forward only (stress / roofline run)",
                             "train": "cfg3 + train.py:131-132,197-198: %d mesh-bound SH-3 splats, %dx%d (HxW), fused L1+SSIM loss vs a white "
                                      "target, backward, densification statistics; no optimiser step"}[
                                 args.workload] % (N, args.height, args.width),
                "splats": N, "width": args.width, "height": args.height, "sh_degree": 3, "scene": args.scene if args.workload != "cfg5" else "unbound cloud",
                "splat_order": "morton (io.spatial_sort: the loaders' default)" if SPATIAL_SORT else "as generated (random; --no-spatial-sort)",
                "backward_seed": "cached device scalar (loss.install_backward_seed, what patch_reference() installs)" if seeded else "torch (ones_like fill; GAA_LOSS_SEED=0)",
                "num_rendered": I_rect, "num_binned": I_binned, "tile_culling": bool(info.get("tile_culling", False)),
                "binning_path": {0: "rank", 1: "depth-ordered scatter", 2: "per-tile sort"}[path] + (f" ({bands} bands of tile rows)" if path == 0 and bands > 1 else ""),
                "visible_fraction": round(vis, 4), "binding": args.binding,
                "parallelism": (((f"frame-parallel x{n_gpus}: {dist.get_world_size()} {'gloo (dry run)' if dry else ('gloo, ranks sharing the GPU (control-flow run)' if shared_gpu else 'RCCL (torch nccl)')} rank(s), one process per GPU, "
                                  f"frames per rank {counts}, one asynchronous scalar all-reduce (loss) per step") if dist is not None else
                                 f"one process, one GPU, {counts[0]} frames in turn; no collective")
                                + (f"; {len(lanes)} frame streams inside the GPU (independent frames overlap; ms_per_step is elapsed / steps, "
                                   f"not the latency of one frame)" if len(lanes) > 1 else "")),
            },
            # every timed round is exactly `steps` steps (barrier + device sync on both sides, MAX over ranks); value = median round
            # (the per-round list is thinned to at most 64 evenly spaced rounds: with --min-seconds uncapped a default run has ~500 rounds, and the
            #  line is read by tools that keep a bounded tail of stdout; n / min / p10 / max are over ALL rounds)
            "rounds": {"n": len(rounds), "frames_per_s": [round(n_gpus * args.steps / rounds[int(i)], 2) for i in np.linspace(0, len(rounds) - 1, min(len(rounds), 64))],
                       "min": round(n_gpus * args.steps / max(rounds), 2), "p10": round(n_gpus * args.steps / float(np.percentile(rounds, 90)), 2),
                       "max": round(n_gpus * args.steps / min(rounds), 2), "timed_seconds": round(float(sum(rounds)), 4)},
            "roofline": dict(roofline, loss=loss_line) if roofline is not None and loss_line is not None else roofline,
            "cpu_baseline": cpu,
            "frame_streams": frame_streams,
            "template_like": template_like,
            # the frame's single host wait (for the instance count): ~0 would mean the host paces the loop, not the GPU
            "host": {"scan_wait_ms_per_step": round(wait_ms / max(waits, 1), 4), "pinned_cpus": len(pinned) if pinned else None,
                     "step_launch": ("hipGraph replay (one launch per %d steps, the frame feed a kernel inside the recording; binning capacity %d for %d instances; %d frame stream(s))" % (lanes[0]["K"], graphed.capacity, info["num_rendered"], len(lanes)))
                                    if graphed is not None else "eager (one Python-driven launch per kernel)"},
        }
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
