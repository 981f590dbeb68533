"""The RCCL path, EXECUTED on the one MI355X a test box has (VERDICT r04 "What's missing" 4): a process group of ONE
rank on the `nccl` backend (= RCCL on ROCm) with GAA_COLLECTIVES_AT_WORLD_1=1, so that every collective of gaussianavatars_amd.frame_parallel
-- the scalar all-reduce of the frame-parallel run, the bucketed gradient all-reduce and its reduce-scatter + all-gather form, the replica
fingerprint MIN / MAX, the densification-statistics SUM / MAX, the timestep broadcast -- goes through the backend's kernels instead of
returning early; and bench.py under torch.distributed.run with --dist-at-1, the command line of the driver's N-GPU run at N = 1.

No scaling is measured here (one GPU): this makes the 8-GPU run a measurement instead of a first execution.  The world-size-2 logic of the
same functions is covered on gloo by tests/test_frame_parallel.py and tests/test_dp_training_cpu.py."""
import json
import os
import socket
import subprocess
import sys
import textwrap

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _env(**extra):
    env = dict(os.environ, PYTHONPATH=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.update(extra)
    return env


@pytest.mark.timeout(280)
def test_collectives_of_frame_parallel_execute_on_rccl_at_world_size_one():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    code = textwrap.dedent("""
        import os, torch
        import torch.distributed as dist
        from gaussianavatars_amd import frame_parallel as FP

        rank, world, local = FP.init_process_group("nccl")
        assert (rank, world, local) == (0, 1, 0) and dist.is_initialized() and dist.get_backend() == "nccl"
        assert FP._collectives_active()
        dev = torch.device("cuda", 0)
        # 1) the path's one collective: a scalar
        x = torch.tensor(3.25, device=dev)
        assert float(FP.allreduce_scalar(x)) == 3.25
        assert float(FP.allreduce_scalar(torch.tensor([7.0], device=dev), op="max")) == 7.0
        # 2) run_frames: both all-reduces on the rank's own device
        acc, n = FP.run_frames(lambda t: torch.tensor(float(t), device=dev), 10, 0, 1)
        assert n == 10 and float(acc) == 45.0 and acc.device.type == "cuda"
        # 3) gradient exchange, both forms, several buckets, a parameter without a gradient, an odd total (reduce-scatter pads)
        g = torch.Generator(device="cpu").manual_seed(5)
        shapes = [(100003, 3), (100003, 1, 3), (100003, 15, 3), (100003, 1), (100003, 4), (7,)]
        for method in ("allreduce", "reduce_scatter"):
            params = [torch.nn.Parameter(torch.randn(s, generator=g).to(dev)) for s in shapes]
            want = []
            for i, p in enumerate(params):
                if i != 3:
                    p.grad = torch.randn(p.shape, generator=g).to(dev)
                want.append(torch.zeros_like(p) if p.grad is None else p.grad.clone())
            FP.allreduce_gradients(params, average=True, bucket_bytes=4 << 20, method=method)
            torch.cuda.synchronize()
            for p, w in zip(params, want):
                assert torch.equal(p.grad, w), method          # one rank: sum == average == the gradient itself, bit for bit, through RCCL
        # 4) replica fingerprint (MIN / MAX of int64), densification statistics (SUM / MAX), timestep broadcast
        FP.check_replica_consistency(params + [torch.arange(5, device=dev)])
        class M: pass
        m = M()
        m.xyz_gradient_accum, m.denom, m.max_radii2D = torch.rand(1000, 1, device=dev), torch.ones(1000, 1, device=dev), torch.rand(1000, device=dev)
        before = (m.xyz_gradient_accum.clone(), m.denom.clone(), m.max_radii2D.clone())
        FP.sync_densification_stats(m)
        assert all(torch.equal(a, b) for a, b in zip(before, (m.xyz_gradient_accum, m.denom, m.max_radii2D)))
        m.binding = None
        assert FP.sync_mesh_for_densification(m, 17) == 17
        dist.barrier()
        torch.cuda.synchronize()
        dist.destroy_process_group()
        print("RCCL_OK", torch.cuda.nccl.version() if hasattr(torch.cuda, "nccl") else "")
    """)
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=_env(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", GAA_COLLECTIVES_AT_WORLD_1="1"),
                       capture_output=True, text=True, timeout=260)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.timeout(290)
def test_bench_under_torch_distributed_run_on_nccl_at_one_gpu():
    """The driver's multi-GPU command line (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N --steps K --warmup W`) at N = 1 with --dist-at-1: nccl process group, one asynchronous scalar
    all-reduce per step, barrier + MAX-over-ranks timing, the JSON line."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "10", "--warmup", "3", "--dist-at-1", "--no-cpu-baseline",
           "--frame-streams", "0", "--min-seconds", "0.3", "--frames", "16"]
    r = subprocess.run(cmd, cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=270)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["steps"] == 10
    assert "RCCL" in d["config"]["parallelism"] and "all-reduce" in d["config"]["parallelism"]
    # the instrumented pass (rank 0 only: its runner carries no collective -- one entered by a single rank of N would pair with the others' barrier) ran as well
    assert d["roofline"] is not None and d["roofline"]["step"]["launches_per_step"] >= 14   # (round 6: the L1 forward and its reduction are one launch)


@pytest.mark.timeout(290)
def test_two_ranks_sharing_the_gpu_run_the_multi_rank_control_flow():
    """RCCL refuses two ranks on one device, so a one-GPU box cannot run N = 2 on `nccl`; `--backend gloo_gpu` runs the REAL step (HIP libraries, compiled host)
    on two ranks that share the GPU with the collectives over gloo: the per-step asynchronous all-reduce, the barriers, the MAX-reduced round times and --
    what this test exists for -- the passes only rank 0 makes (the instrumented pass must not enter a collective: round 5 found that it did, which would have
    hung the first N > 1 run) all execute as they will on N GPUs.  The line says it is not a measurement."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo_gpu", "--steps", "10", "--warmup", "3", "--no-cpu-baseline", "--frame-streams", "0",
           "--min-seconds", "0.3", "--frames", "16", "--no-pin"]
    r = subprocess.run(cmd, cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=270)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                     # rank 0 prints ONE line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and "NOT A MEASUREMENT" in d["data"]
    assert "frames per rank [8, 8]" in d["config"]["parallelism"]
    assert d["roofline"] is not None and d["roofline"]["step"]["launches_per_step"] >= 14   # (round 6: the L1 forward and its reduction are one launch)
