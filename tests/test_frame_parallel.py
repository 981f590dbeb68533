"""CPU tests of the multi-GPU path (world_size 2, gloo): frame sharding and the one collective."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def test_frames_for_rank_partition():
    from gaussianavatars_amd.frame_parallel import frames_for_rank

    T, Wn = 300, 8
    parts = [frames_for_rank(T, r, Wn) for r in range(Wn)]
    assert [len(p) for p in parts] == [38, 38, 38, 38, 37, 37, 37, 37]
    assert sorted(sum(parts, [])) == list(range(T))
    assert frames_for_rank(3, 5, 8) == []
    with pytest.raises(ValueError):
        frames_for_rank(10, 8, 8)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import torch.distributed as dist

    from gaussianavatars_amd import frame_parallel as fp

    r, w, _ = fp.init_process_group("gloo")
    assert (r, w) == (rank, world)
    total, count = fp.run_frames(lambda t: torch.tensor(float(t * t)), 21, r, w)
    one = fp.allreduce_scalar(torch.tensor(float(rank + 1)), "max")
    # data-parallel gradient all-reduce (bucketed): two leaves, tiny bucket size to force several buckets
    a, b = torch.nn.Parameter(torch.zeros(5, 3)), torch.nn.Parameter(torch.zeros(7))
    a.grad, b.grad = torch.full((5, 3), float(rank + 1)), torch.arange(7.0) * (rank + 1)
    fp.allreduce_gradients([a, b], average=True, bucket_bytes=32)
    assert torch.allclose(a.grad, torch.full((5, 3), 1.5)) and torch.allclose(b.grad, torch.arange(7.0) * 1.5)
    # reduce-scatter + all-gather form (element count not a multiple of the world size: padded), and a parameter without a
    # gradient on one rank (it contributes zeros; both ranks build the same bucket layout)
    c, d = torch.nn.Parameter(torch.zeros(5, 3)), torch.nn.Parameter(torch.zeros(4))
    c.grad = torch.full((5, 3), float(rank + 1))
    if rank == 0:
        d.grad = torch.arange(4.0)
    fp.allreduce_gradients([c, d], average=False, method="reduce_scatter")
    assert torch.allclose(c.grad, torch.full((5, 3), 3.0)) and torch.allclose(d.grad, torch.arange(4.0))
    # a rank that owns no frame still takes part in the collectives (its accumulator lives on its own device)
    t1, n1 = fp.run_frames(lambda t: torch.tensor(5.0), 1, r, w)
    assert float(t1) == 5.0 and n1 == 1
    # replicas that diverged (one rank densified differently) are detected before gradients are exchanged
    fp.check_replica_consistency([a, b, torch.arange(6)])
    try:
        fp.check_replica_consistency([torch.zeros(5 + rank, 3)])
        diverged = False
    except RuntimeError as e:
        diverged = "replicas diverged" in str(e)
    assert diverged
    # rank-consistent densification statistics + RNG
    class M:
        pass
    m = M()
    m.xyz_gradient_accum, m.denom, m.max_radii2D = torch.full((4, 1), float(rank + 1)), torch.full((4, 1), 1.0), torch.tensor([1.0, 5.0, 2.0, 0.0]) * (rank + 1)
    fp.sync_densification_stats(m)
    assert torch.equal(m.xyz_gradient_accum, torch.full((4, 1), 3.0)) and torch.equal(m.denom, torch.full((4, 1), 2.0))
    assert torch.equal(m.max_radii2D, torch.tensor([2.0, 10.0, 4.0, 0.0]))
    fp.seed_all_ranks(1234)
    draw = torch.rand(3)
    both = [torch.zeros(3) for _ in range(w)]
    dist.all_gather(both, draw)
    assert torch.equal(both[0], both[1])
    # bench.py's frame loop: asynchronous scalar all-reduce ordering, frame wrap-around, MAX-reduced round times
    import bench
    mine = fp.frames_for_rank(7, r, w)                      # rank 0: 0 2 4 6, rank 1: 1 3 5
    seen = []
    def step(t):
        seen.append(t)
        return torch.tensor(float(t))
    run = bench.make_runner(step, mine, dist, torch.device("cpu"), post_step=None)
    total5 = run(5, 2)                                      # 5 steps starting at offset 2: wraps around this rank's list
    exp0 = [[0, 2, 4, 6][(2 + i) % 4] for i in range(5)]
    exp1 = [[1, 3, 5][(2 + i) % 3] for i in range(5)]
    assert seen == (exp0 if r == 0 else exp1)
    assert float(total5) == float(sum(exp0) + sum(exp1))    # every step's scalar was all-reduced exactly once
    rounds = bench.timed_rounds(run, lambda: dist.barrier(), 3, 1, dist, torch.device("cpu"), min_rounds=2, min_seconds=0.0)
    both_r = [None, None]
    dist.all_gather_object(both_r, rounds)
    assert len(rounds) == 2 and both_r[0] == both_r[1]      # MAX over ranks: identical on every rank
    out.put((rank, float(total), count, float(one)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_allreduce():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    expect = float(sum(t * t for t in range(21)))
    for rank, total, count, mx in res:
        assert total == expect and count == 21 and mx == 2.0


def test_cpu_list_parsing_and_pinning_is_a_no_op_without_a_gpu():
    import os

    from gaussianavatars_amd.frame_parallel import _cpulist, pin_to_gpu_numa_node

    assert _cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    assert _cpulist("") == []
    before = os.sched_getaffinity(0)
    if not torch.cuda.is_available():
        assert pin_to_gpu_numa_node(0) is None       # topology unreadable: nothing changes
        assert os.sched_getaffinity(0) == before


def test_pin_host_process_is_a_no_op_without_a_gpu_and_children_get_the_original_mask(monkeypatch):
    """patch_reference() / `python -m gaussianavatars_amd.run` pin by default (frame_parallel.pin_host_process).  Here (no GPU): nothing
    changes.  The fork hook is exercised with a stand-in for the topology: the parent narrows itself to one CPU, a forked child (a
    DataLoader worker of the reference's train.py:55) must see the original mask again."""
    import os

    from gaussianavatars_amd import frame_parallel as FP

    before = os.sched_getaffinity(0)
    if not torch.cuda.is_available():
        assert FP.pin_host_process() is None and os.sched_getaffinity(0) == before
    monkeypatch.setenv("GAA_PIN", "0")
    assert FP.pin_host_process() is None and os.sched_getaffinity(0) == before     # the opt-out
    monkeypatch.delenv("GAA_PIN")
    one = sorted(before)[:1]

    def fake_pin(device_index=0, cores=8):
        os.sched_setaffinity(0, one)
        return one

    monkeypatch.setattr(FP, "pin_to_gpu_numa_node", fake_pin)
    monkeypatch.setattr(FP.torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(FP.torch.cuda, "device_count", lambda: 1)
    monkeypatch.setitem(FP._PIN, "pinned", None)
    try:
        assert FP.pin_host_process() == one and os.sched_getaffinity(0) == set(one)
        assert FP.pin_host_process() == one                                         # idempotent
        r, w = os.pipe()
        pid = os.fork()
        if pid == 0:
            os.write(w, repr(sorted(os.sched_getaffinity(0))).encode())
            os._exit(0)
        os.waitpid(pid, 0)
        child = eval(os.read(r, 1 << 16).decode())
        assert set(child) == set(before), "a forked worker keeps the trainer's narrowed CPU mask"
    finally:
        os.sched_setaffinity(0, before)
        FP._PIN["pinned"] = FP._PIN["original"] = None


@pytest.mark.gpu
def test_pin_host_process_moves_the_process_next_to_the_gpu():
    """On a GPU box the default of patch_reference() narrows the CPU mask to (at most) eight physical cores of the GPU's NUMA node."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import os, json; before = sorted(os.sched_getaffinity(0));\n"
            "from gaussianavatars_amd.frame_parallel import pin_host_process, _gpu_numa_node, _cpulist\n"
            "got = pin_host_process(); node = _gpu_numa_node(0)\n"
            "cpus = _cpulist(open('/sys/devices/system/node/node%d/cpulist' % node).read()) if node >= 0 else []\n"
            "print(json.dumps(dict(before=before, got=got, after=sorted(os.sched_getaffinity(0)), node=node, node_cpus=cpus)))")
    env = {k: v for k, v in os.environ.items() if k != "GAA_PIN"}
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=240, cwd=root, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    import json

    d = json.loads(r.stdout.strip().splitlines()[-1])
    if d["node"] < 0:
        pytest.skip("the GPU reports no NUMA node on this host")
    assert d["got"] is not None and d["after"] == sorted(d["got"]) and 1 <= len(d["got"]) <= 8
    assert set(d["got"]) <= set(d["node_cpus"]) and set(d["got"]) <= set(d["before"])
    r0 = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=240, cwd=root, env=dict(env, GAA_PIN="0"))
    d0 = json.loads(r0.stdout.strip().splitlines()[-1])
    assert d0["got"] is None and d0["after"] == d0["before"]


def _bench(*argv, env_extra=None, timeout=300):
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), *argv], capture_output=True, text=True, timeout=timeout, env=env, cwd=root)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, [json.loads(ln) for ln in lines]


def test_bench_gpus_flag_starts_that_many_ranks():
    """`python bench.py --gpus 2` with no launcher around it (what the driver's scaling run types): the flag itself starts two ranks
    (torch.distributed.run on 127.0.0.1).  Dry path (--backend gloo): the rasterizer Function is a stub, everything around it -- rank
    start-up, frame sharding, the run loop with its asynchronous scalar all-reduce, MAX-reduced rounds, the single JSON line -- is
    the code the GPU run executes."""
    r, lines = _bench("--gpus", "2", "--backend", "gloo", "--steps", "4", "--warmup", "1", "--rounds", "2", "--min-seconds", "0", "--frames", "7")
    assert r.returncode == 0, r.stdout + r.stderr
    assert len(lines) == 1                                   # ONE line, from rank 0
    out = lines[0]
    assert out["n_gpus"] == 2 and out["steps"] == 4 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert "2 gloo (dry run) rank(s)" in out["config"]["parallelism"] and "frames per rank [4, 3]" in out["config"]["parallelism"]
    assert out["rounds"]["n"] == 2 and len(out["rounds"]["frames_per_s"]) == 2
    assert out["data"].startswith("DRY RUN")                 # never mistaken for a measurement
    assert out["roofline"] is None and out["cpu_baseline"] is None
    # value = all ranks' frames / the MAX-reduced median round
    assert abs(out["value"] - 2 * 4 / (out["ms_per_step"] * 4e-3)) / out["value"] < 1e-3


def test_two_ranks_of_two_lanes_dry():
    """lanes x ranks (round 4): `bench.py --gpus 2 --streams 2` -- every rank deals ITS frames (t mod 2) to two frame lanes that share
    one set of splat parameters (graphs.shared_lane_model), the lanes add their losses up and the rank contributes the sum to ONE scalar
    all-reduce per run.  Dry path (--backend gloo): the lanes are plain step functions, the rasterizer Function a stub; the frame
    bookkeeping, the lane models, the collective and the JSON line are the GPU run's."""
    r, lines = _bench("--gpus", "2", "--backend", "gloo", "--streams", "2", "--steps", "6", "--warmup", "2", "--rounds", "2", "--min-seconds", "0", "--frames", "9")
    assert r.returncode == 0, r.stdout + r.stderr
    assert len(lines) == 1
    out = lines[0]
    par = out["config"]["parallelism"]
    assert out["n_gpus"] == 2 and "2 gloo (dry run) rank(s)" in par and "frames per rank [5, 4]" in par
    assert "2 frame streams inside the GPU" in par
    assert out["rounds"]["n"] == 2 and out["value"] > 0 and out["data"].startswith("DRY RUN")
    # the same run on one lane per rank: same frames, same collective result is not observable from the line -- the frame walk of the
    # lanes is checked against the eager loop's in test_recorded_lanes_walk_the_eager_loops_frames; here: the lane models share storage
    from gaussianavatars_amd import synthetic as S
    from gaussianavatars_amd.gaussian_model import FlameGaussianModel
    from gaussianavatars_amd.graphs import LEAF_NAMES, accumulate_lane_grads, shared_lane_model

    g = FlameGaussianModel(1, S.flame_rig(seed=4), binding_impl="unfused", device="cpu")
    g.load_arrays(S.bound_splats(S.FLAME_F + 10, S.FLAME_F, 1, seed=2), device="cpu", requires_grad=True)
    g.load_flame_param(S.flame_sequence(3, seed=4), device="cpu", requires_grad=True)
    a, b = shared_lane_model(g), shared_lane_model(g)
    for n in LEAF_NAMES:
        assert getattr(a, n).data_ptr() == getattr(g, n).data_ptr() and getattr(a, n) is not getattr(b, n) and getattr(a, n).is_leaf
    assert a.flame_param["expr"].data_ptr() != g.flame_param["expr"].data_ptr()      # a lane feeds its own frame rows
    (a._xyz.sum() * 2.0 + a.flame_param["expr"].sum()).backward()
    (b._xyz.sum() * 3.0 + (b._opacity ** 2).sum()).backward()
    assert g._xyz.grad is None
    accumulate_lane_grads(g, [a, b])
    assert torch.equal(g._xyz.grad, torch.full_like(g._xyz, 5.0)) and torch.equal(g._opacity.grad, 2.0 * g._opacity.detach())
    assert torch.equal(g.flame_param["expr"].grad, torch.ones_like(g.flame_param["expr"])) and g._scaling.grad is None
    with torch.no_grad():
        g._xyz.add_(1.0)                       # one optimiser step on the shared storage moves every lane
    assert torch.equal(a._xyz, g._xyz) and torch.equal(b._xyz, g._xyz)


def test_bench_gpus_flag_refuses_what_it_cannot_deliver():
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs present: the refusal does not apply")
    r, lines = _bench("--gpus", "2", "--steps", "2")           # RCCL path on a box without two GPUs: loud, no line
    assert r.returncode != 0 and not lines and "GPU(s) visible" in r.stderr
    # a launcher whose world size contradicts the flag (the line would misreport n_gpus)
    r, lines = _bench("--gpus", "1", "--backend", "gloo", "--steps", "2", env_extra=dict(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"))
    assert r.returncode != 0 and not lines and "must agree" in r.stderr


def test_recorded_lanes_walk_the_eager_loops_frames():
    """bench.py --graph --streams L --graph-frames K: the device-side schedules of the lanes (lane_schedule) and the per-run plan
    (lane_plan: where each lane's cursor starts, which recordings it replays) together render exactly the frames the eager loop renders,
    my_frames[(offset + i) % len] for i < n, each once, whatever n, offset, L and K are."""
    import bench

    for frames in (list(range(12)), [3, 7, 0, 11, 5], list(range(0, 300, 8))):
        for L in (1, 2, 3, 4, 8):
            scheds = [bench.lane_schedule(frames, j, L) for j in range(L)]
            for K in (1, 4, 5):
                offset = 0
                for n in (5, 20, 7, 40, 1):                      # consecutive runs, as timed_rounds issues them (warm-up first)
                    plan = bench.lane_plan(n, offset, L, K)
                    got = {}
                    for j, (start, sizes) in enumerate(plan):
                        assert all(s in (1, K) for s in sizes)
                        cur = start
                        for _ in range(sum(sizes)):
                            g = j + L * cur                          # the run-global index of the lane's cur-th frame
                            assert g not in got
                            got[g] = scheds[j][cur % len(scheds[j])]
                            cur += 1
                    want = {offset + i: frames[(offset + i) % len(frames)] for i in range(n)}
                    assert got == want, (frames[:4], L, K, n, offset)
                    offset += n


def test_pinning_slot_comes_from_the_physical_topology(tmp_path, monkeypatch):
    """N one-GPU processes (HIP_VISIBLE_DEVICES=k each) all see "device 0"; their core groups must still be disjoint.
    _node_gpu_bdfs reads the node's GPUs from sysfs whatever the process may see; _slot_among_node_gpus ranks this GPU among them."""
    from gaussianavatars_amd import frame_parallel as FP

    root = tmp_path / "pci"
    devs = {"0000:05:00.0": ("0x1002", "0x120000", 0), "0000:15:00.0": ("0x1002", "0x120000", 0), "0000:15:00.1": ("0x1002", "0x040300", 0),
            "0000:25:00.0": ("0x1002", "0x030000", 0), "0000:65:00.0": ("0x1002", "0x120000", 1), "0000:03:00.0": ("0x8086", "0x020000", 0)}
    for bdf, (vendor, cls, node) in devs.items():
        d = root / bdf
        d.mkdir(parents=True)
        (d / "vendor").write_text(vendor + "\n")
        (d / "class").write_text(cls + "\n")
        (d / "numa_node").write_text(f"{node}\n")
    assert FP._node_gpu_bdfs(0, str(root)) == ["0000:05:00.0", "0000:15:00.0", "0000:25:00.0"]       # function 0 of AMD display / accelerator devices of node 0
    assert FP._node_gpu_bdfs(1, str(root)) == ["0000:65:00.0"]
    real = FP._node_gpu_bdfs
    monkeypatch.setattr(FP, "_node_gpu_bdfs", lambda node: real(node, str(root)))
    for k, bdf in enumerate(["0000:05:00.0", "0000:15:00.0", "0000:25:00.0"]):
        monkeypatch.setattr(FP, "_gpu_bdf", lambda i, bdf=bdf: bdf)
        assert FP._slot_among_node_gpus(0, 0) == (k, 3)        # every process asks about ITS "device 0": three different slots
