"""Frame-parallel multi-GPU driver: one process per GPU, splat parameters and
FLAME tables replicated, frames (camera, timestep) sharded round-robin, and exactly one collective
on the data path -- the all-reduce of a scalar (loss / metric sum) over RCCL/xGMI.  A single frame
cannot be split across GPUs without coupling every tile through the global depth order, so there
is no tile- or splat-sharded mode (replicas only at frame granularity).

The reference has no multi-GPU path at all; this is an addition, launched with
`python -m torch.distributed.run --nproc-per-node N ...` (rendezvous on 127.0.0.1).
"""
from __future__ import annotations

import os
from typing import List, Optional

import torch


def frames_for_rank(num_frames: int, rank: int, world_size: int) -> List[int]:
    """Frame t goes to rank t mod world_size (300 frames on 8 ranks: 38/38/38/38/37/37/37/37)."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    return list(range(rank, num_frames, world_size))


def _cpulist(spec: str) -> List[int]:
    cpus: List[int] = []
    for part in spec.strip().split(","):
        if part:
            lo, _, hi = part.partition("-")
            cpus.extend(range(int(lo), int(hi or lo) + 1))
    return cpus


def _gpu_bdf(device_index: int) -> str:
    props = torch.cuda.get_device_properties(device_index)
    return "%04x:%02x:%02x.0" % (props.pci_domain_id, props.pci_bus_id, props.pci_device_id)


def _gpu_numa_node(device_index: int) -> int:
    with open("/sys/bus/pci/devices/%s/numa_node" % _gpu_bdf(device_index)) as f:
        return int(f.read().strip())


def _node_gpu_bdfs(node: int, root: str = "/sys/bus/pci/devices") -> List[str]:
    """Every AMD GPU function 0 of NUMA node `node`, in PCI order, read from sysfs -- whatever this process is allowed to SEE.  In the
    documented one-process-per-GPU mode (HIP_VISIBLE_DEVICES=k, utils/general_utils.py:133 pins cuda:0) torch reports one device to every
    process, so the slot among the node's GPUs has to come from the physical topology: vendor 0x1002, class display (0x03xx) or processing
    accelerator (0x12xx)."""
    out = []
    for bdf in sorted(os.listdir(root)):
        if not bdf.endswith(".0"):
            continue
        try:
            with open(os.path.join(root, bdf, "vendor")) as f:
                if f.read().strip().lower() != "0x1002":
                    continue
            with open(os.path.join(root, bdf, "class")) as f:
                cls = f.read().strip().lower()
            if not (cls.startswith("0x03") or cls.startswith("0x12")):
                continue
            with open(os.path.join(root, bdf, "numa_node")) as f:
                if int(f.read().strip()) != node:
                    continue
        except (OSError, ValueError):
            continue
        out.append(bdf)
    return out


def _slot_among_node_gpus(device_index: int, node: int) -> tuple:
    """-> (slot, number of GPUs sharing the node).  Physical topology first (see _node_gpu_bdfs); when sysfs does not list this GPU (a
    container without the PCI tree) the devices torch can see, and LOCAL_RANK when it sees only one of several ranks' GPUs."""
    mine = _gpu_bdf(device_index)
    try:
        bdfs = _node_gpu_bdfs(node)
    except OSError:
        bdfs = []
    if mine in bdfs:
        return bdfs.index(mine), len(bdfs)
    peers = [d for d in range(torch.cuda.device_count()) if _gpu_numa_node(d) == node]
    if len(peers) > 1 or torch.cuda.device_count() > 1:
        return (peers.index(device_index) if device_index in peers else 0), max(len(peers), 1)
    world = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")) or 1)
    rank = int(os.environ.get("LOCAL_RANK", "0") or 0)
    return (rank % max(world, 1)), max(world, 1)


def pin_to_gpu_numa_node(device_index: int = 0, cores: int = 8) -> Optional[List[int]]:
    """Restrict this process to `cores` neighbouring physical cores of the NUMA node the GPU hangs off (one process per GPU:
    each rank calls it with its LOCAL_RANK; GPUs sharing a node get disjoint core groups).

    The frame loop is host-paced: ~45 small kernel launches, one mapped-pinned word polled per frame, and a hand-off to the
    autograd thread and back.  From the far socket of a two-socket host every doorbell write and every poll crosses the
    inter-socket link (0.75 ms instead of 0.55 ms of host time per frame); with the two threads on different core complexes
    of the right socket the hand-offs bounce between L3 slices (0.61 ms).  A group of 8 consecutive physical cores is one
    core complex on the EPYC hosts of the MI355X boxes.  Returns the CPU list, or None when the topology cannot be read
    (then nothing is changed)."""
    try:
        node = _gpu_numa_node(device_index)
        if node < 0:
            return None
        with open("/sys/devices/system/node/node%d/cpulist" % node) as f:
            node_cpus = _cpulist(f.read())
        # one logical CPU per physical core (drop SMT siblings), in core order
        firsts = []
        for c in node_cpus:
            with open("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % c) as f:
                if min(_cpulist(f.read())) == c:
                    firsts.append(c)
        allowed_now = set(os.sched_getaffinity(0))
        firsts = [c for c in sorted(firsts) if c in allowed_now]
        if not firsts:
            return None
        # my slot among the GPUs of this node: from the PHYSICAL topology, so that N one-GPU processes (HIP_VISIBLE_DEVICES=k each, every
        # one of them seeing "device 0") take N disjoint core groups instead of all landing on the node's first eight cores
        slot, n_peers = _slot_among_node_gpus(device_index, node)
        cores = max(1, min(cores, len(firsts) // max(n_peers, 1)))
        slot %= max(1, len(firsts) // cores)
        mine = firsts[slot * cores: slot * cores + cores] or firsts[:cores]
        os.sched_setaffinity(0, mine)
        return mine
    except (OSError, ValueError, AttributeError, RuntimeError):
        return None


_PIN: dict = {"original": None, "pinned": None, "hooked": False}


def _restore_affinity_in_child() -> None:
    original = _PIN["original"]
    if original:
        try:
            os.sched_setaffinity(0, original)
        except OSError:
            pass


def pin_host_process(device_index: Optional[int] = None, cores: int = 8) -> Optional[List[int]]:
    """What `patch_reference()` / `python -m gaussianavatars_amd.run` do for an unchanged entry script by default: the frame loop's two
    threads (main + autograd) go next to their GPU (pin_to_gpu_numa_node), exactly the host conditions bench.py measures under
    (un-pinned, the same loop ran 1 930 instead of 3 044 frames/s on a two-socket box: profiles/r03_l_bench_cfg3_eager_unpinned.json).
    `GAA_PIN=0` opts out.  The device is LOCAL_RANK's when several are visible, else 0 (utils/general_utils.py:133 pins cuda:0; one
    process per GPU selects its device with HIP_VISIBLE_DEVICES).  Forked children -- the reference's DataLoader workers, train.py:55
    -- get the ORIGINAL mask back (os.register_at_fork), so eight of them do not pile onto the trainer's eight cores.  Idempotent;
    returns the CPU list, or None when nothing was changed (opt-out, no GPU, unreadable topology)."""
    if os.environ.get("GAA_PIN", "1") == "0":
        return None
    if _PIN["pinned"] is not None:
        return _PIN["pinned"]
    try:
        if not torch.cuda.is_available():
            return None
        n = torch.cuda.device_count()
        if device_index is None:
            device_index = int(os.environ.get("LOCAL_RANK", "0"))
            if not (0 <= device_index < n):
                device_index = 0
        original = set(os.sched_getaffinity(0))
    except (OSError, ValueError, AttributeError, RuntimeError, AssertionError):
        return None
    mine = pin_to_gpu_numa_node(device_index, cores)
    if mine is None:
        return None
    _PIN["original"], _PIN["pinned"] = original, list(mine)
    if not _PIN["hooked"]:
        os.register_at_fork(after_in_child=_restore_affinity_in_child)
        _PIN["hooked"] = True
    return _PIN["pinned"]


def init_process_group(backend: Optional[str] = None):
    """Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* from the environment (torchrun contract).
    backend: 'nccl' (= RCCL on ROCm) on GPUs, 'gloo' on CPU."""
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
    if (world > 1 or os.environ.get("GAA_COLLECTIVES_AT_WORLD_1", "0") == "1") and not dist.is_initialized():
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local_rank


def _collectives_active() -> bool:
    """True when the collectives below have to run: a process group of more than one rank -- or of ONE rank with GAA_COLLECTIVES_AT_WORLD_1=1,
    which sends every call through the backend anyway (the single MI355X of a test box then executes the RCCL kernels the 8-GPU run
    launches: tests/test_rccl_gpu.py)."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get("GAA_COLLECTIVES_AT_WORLD_1", "0") == "1"


def allreduce_scalar(value: torch.Tensor, op: str = "sum") -> torch.Tensor:
    """In-place all-reduce of a 0-d / 1-element tensor; identity when not distributed."""
    import torch.distributed as dist

    if _collectives_active():
        dist.all_reduce(value, op=dist.ReduceOp.SUM if op == "sum" else dist.ReduceOp.MAX)
    return value


def _collective_device(device=None) -> torch.device:
    """Where a rank's collective buffers must live: the rank's GPU under nccl/RCCL (a CPU tensor there fails on this rank
    while the others block in the collective), the host under gloo."""
    import torch.distributed as dist

    if device is not None:
        return torch.device(device)
    if dist.is_available() and dist.is_initialized() and dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def run_frames(step_fn, num_frames: int, rank: int, world_size: int, device=None):
    """Calls step_fn(t) -> scalar tensor for every frame of this rank and returns the all-reduced sum
    together with the global frame count (every rank gets the same pair).  A rank that owns no frame (num_frames <
    world_size) still takes part in both collectives, with zeros on its own device."""
    mine = frames_for_rank(num_frames, rank, world_size)
    dev = _collective_device(device)
    acc = torch.zeros((), dtype=torch.float32, device=dev)
    for t in mine:
        acc = acc + step_fn(t).detach().reshape(()).to(device=dev, dtype=torch.float32)
    count = torch.tensor(float(len(mine)), device=dev)
    allreduce_scalar(acc)
    allreduce_scalar(count)
    return acc, int(count.item())


def replica_fingerprint(tensors) -> torch.Tensor:
    """A few integers that must agree on every rank before gradients are exchanged: number of tensors, total element count,
    a shape hash, and the checksum of every integer tensor (the `binding` of a mesh-bound model)."""
    n, total, h, chk = 0, 0, 0, 0
    for t in tensors:
        if t is None:
            continue
        n += 1
        total += t.numel()
        for d in t.shape:
            h = (h * 1000003 + int(d) + 1) % 2147483629
        h = (h * 1000003 + 7) % 2147483629
        if not t.is_floating_point():
            chk = (chk + int(t.detach().long().sum().item())) % 2147483629
    return torch.tensor([n, total % 2147483629, h, chk], dtype=torch.int64)


def check_replica_consistency(tensors, device=None) -> None:
    """Raises on every rank if the replicas differ in shape (one rank densified differently) -- BEFORE an all-reduce over
    mismatched buffers corrupts gradients or hangs."""
    import torch.distributed as dist

    if not _collectives_active():
        return
    fp = replica_fingerprint(tensors).to(_collective_device(device))
    lo, hi = fp.clone(), fp.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    if not torch.equal(lo, hi):
        raise RuntimeError(f"replicas diverged: rank {dist.get_rank()} holds fingerprint {fp.tolist()}, ranks span {lo.tolist()} .. {hi.tolist()} "
                           "(tensor count, elements, shape hash, integer checksum)")


def sync_densification_stats(model) -> None:
    """Rank-consistent densification.  The reference decides clone / split / prune from statistics each
    process accumulates over the frames IT rendered (scene/gaussian_model.py:426-519: xyz_gradient_accum / denom /
    max_radii2D); replicas that see different frames would diverge at the first densify_and_prune.  Summing the two
    accumulators and taking the maximum of the radii over all ranks right before that call gives every rank the statistics of
    the whole frame set, hence identical decisions (together with seed_all_ranks for the split sampling and, for a mesh-bound model,
    sync_mesh_for_densification: the decisions also read the current mesh)."""
    import torch.distributed as dist

    if not _collectives_active():
        return
    dist.all_reduce(model.xyz_gradient_accum, op=dist.ReduceOp.SUM)
    dist.all_reduce(model.denom, op=dist.ReduceOp.SUM)
    dist.all_reduce(model.max_radii2D, op=dist.ReduceOp.MAX)


def sync_mesh_for_densification(model, timestep: int, src: int = 0) -> int:
    """densify_and_prune reads the CURRENT mesh: `get_scaling` (= exp(_scaling) * face_scaling of the last select_mesh_by_timestep) decides
    clone vs split and the size prune, and the split writes `selected_scaling / face_scaling` into the new splats
    (scene/gaussian_model.py:462-475, 498-515).  Every replica just rendered a DIFFERENT frame, so the decisions would differ (found by
    check_replica_consistency in tests/test_dp_training_cpu.py): all ranks take rank `src`'s timestep for the densification.
    Returns that timestep."""
    import torch.distributed as dist

    t = int(timestep)
    if _collectives_active():
        buf = torch.tensor([t], dtype=torch.int64, device=_collective_device())
        dist.broadcast(buf, src=src)
        t = int(buf.item())
    if getattr(model, "binding", None) is not None:
        with torch.no_grad():
            model.select_mesh_by_timestep(t)
    return t


def seed_all_ranks(iteration: int, base_seed: int = 0) -> int:
    """The same RNG state on every rank for the sampling inside densify_and_split (scene/gaussian_model.py:462-466
    draws torch.normal): a function of the iteration only.  Returns the seed."""
    seed = (int(base_seed) * 1000003 + int(iteration)) % (2 ** 31 - 1)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    return seed


def allreduce_gradients(params, average: bool = True, bucket_bytes: int = 64 << 20, method: str = "allreduce", check: bool = True) -> None:
    """Data-parallel training on top of frame-parallel rendering: sums (or averages) the
    `.grad` of the given leaf tensors over all ranks.  Gradients are packed into flat buckets so that 100k
    splats (236 B each, ~24 MB) travel as ONE collective: xGMI is point-to-point (7 links x ~153 GB/s per GPU),
    so a few large collectives amortise the per-collective latency that dozens of per-tensor calls would pay.

    method="reduce_scatter": every bucket goes through reduce-scatter + all-gather (each rank reduces 1/N of the bucket, all
    7 links of every GPU carry traffic in both phases) instead of one all-reduce; same result.
    A parameter whose .grad is None on this rank (it rendered no frame touching it) contributes zeros, so every rank builds
    the same bucket layout; with `check`, the replicas' shapes are compared first (check_replica_consistency)."""
    import torch.distributed as dist

    if not _collectives_active():
        return
    world = dist.get_world_size()
    params = [p for p in params if p is not None]
    if check:
        check_replica_consistency(params)
    for p in params:
        if p.grad is None:
            p.grad = torch.zeros_like(p)
    grads = [p.grad for p in params]
    bucket, size = [], 0

    def flush():
        nonlocal bucket, size
        if not bucket:
            return
        n = sum(g.numel() for g in bucket)
        if method == "reduce_scatter":
            pad = (-n) % world
            flat = torch.cat([g.reshape(-1) for g in bucket] + ([bucket[0].new_zeros(pad)] if pad else []))
            shard = torch.empty(flat.numel() // world, dtype=flat.dtype, device=flat.device)
            dist.reduce_scatter_tensor(shard, flat, op=dist.ReduceOp.SUM)
            if average:
                shard /= world
            dist.all_gather_into_tensor(flat, shard)
        else:
            flat = torch.cat([g.reshape(-1) for g in bucket])
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
            if average:
                flat /= world
        off = 0
        for g in bucket:
            k = g.numel()
            g.copy_(flat[off: off + k].view_as(g))
            off += k
        bucket, size = [], 0

    for g in grads:
        nbytes = g.numel() * g.element_size()
        if bucket and (size + nbytes > bucket_bytes or g.dtype != bucket[0].dtype):
            flush()
        bucket.append(g)
        size += nbytes
    flush()
