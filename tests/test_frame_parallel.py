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
