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


def _gpu_numa_node(device_index: int) -> int:
    props = torch.cuda.get_device_properties(device_index)
    bdf = "%04x:%02x:%02x.0" % (props.pci_domain_id, props.pci_bus_id, props.pci_device_id)
    with open("/sys/bus/pci/devices/%s/numa_node" % bdf) as f:
        return int(f.read().strip())


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
        # my slot among the GPUs of this node
        peers = [d for d in range(torch.cuda.device_count()) if _gpu_numa_node(d) == node]
        slot = peers.index(device_index) if device_index in peers else 0
        cores = max(1, min(cores, len(firsts) // max(len(peers), 1)))
        mine = firsts[slot * cores: slot * cores + cores] or firsts[:cores]
        os.sched_setaffinity(0, mine)
        return mine
    except (OSError, ValueError, AttributeError, RuntimeError):
        return None


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
    if world > 1 and not dist.is_initialized():
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local_rank


def allreduce_scalar(value: torch.Tensor, op: str = "sum") -> torch.Tensor:
    """In-place all-reduce of a 0-d / 1-element tensor; identity when not distributed."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(value, op=dist.ReduceOp.SUM if op == "sum" else dist.ReduceOp.MAX)
    return value


def run_frames(step_fn, num_frames: int, rank: int, world_size: int):
    """Calls step_fn(t) -> scalar tensor for every frame of this rank and returns the all-reduced sum
    together with the global frame count (every rank gets the same pair)."""
    mine = frames_for_rank(num_frames, rank, world_size)
    acc = None
    for t in mine:
        v = step_fn(t).detach().reshape(())
        acc = v.clone() if acc is None else acc + v
    if acc is None:
        acc = torch.zeros(())
    count = torch.tensor(float(len(mine)), device=acc.device)
    allreduce_scalar(acc)
    allreduce_scalar(count)
    return acc, int(count.item())


def allreduce_gradients(params, average: bool = True, bucket_bytes: int = 64 << 20) -> None:
    """Data-parallel training on top of frame-parallel rendering: sums (or averages) the
    `.grad` of the given leaf tensors over all ranks.  Gradients are packed into flat buckets so that 100k
    splats (236 B each, ~24 MB) travel as ONE collective: xGMI is point-to-point (7 links x ~153 GB/s per GPU),
    so a few large all-reduces amortise the per-collective latency that dozens of per-tensor calls would pay.
    Ranks must hold identical parameter shapes (replicated splats, consistent densification)."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    world = dist.get_world_size()
    grads = [p.grad for p in params if p is not None and p.grad is not None]
    bucket, size = [], 0

    def flush():
        nonlocal bucket, size
        if not bucket:
            return
        flat = torch.cat([g.reshape(-1) for g in bucket])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        if average:
            flat /= world
        off = 0
        for g in bucket:
            n = g.numel()
            g.copy_(flat[off: off + n].view_as(g))
            off += n
        bucket, size = [], 0

    for g in grads:
        nbytes = g.numel() * g.element_size()
        if bucket and (size + nbytes > bucket_bytes or g.dtype != bucket[0].dtype):
            flush()
        bucket.append(g)
        size += nbytes
    flush()
