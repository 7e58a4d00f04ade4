"""CPU, gloo, world size 2: the multi-process plumbing bench.py and sharded inference use on N GPUs."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from endodav_amd import parallel


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_units, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    parallel.init("gloo")
    mine = parallel.clip_shard(n_units, rank, world)
    # stand-in for "run the forward on my clips": a per-unit array that only depends on the unit index
    results = [np.full((2, 3), float(i), np.float32) for i in mine]
    parallel.barrier()
    elapsed = parallel.max_over_ranks(1.0 + rank)  # rank 1 is the slow one
    gathered = parallel.gather_to_rank0(results)
    if rank == 0:
        merged = parallel.merge_shards(gathered, n_units)
        q.put((elapsed, [float(m[0, 0]) for m in merged]))
    parallel.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_units", [5, 8])
def test_two_rank_sharding_gloo(n_units):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_units, q)) for r in range(2)]
    for p in procs:
        p.start()
    elapsed, order = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert elapsed == 2.0  # MAX over ranks
    assert order == [float(i) for i in range(n_units)]  # every unit exactly once, in order


def test_shard_partition_properties():
    for n in (0, 1, 7, 64):
        for world in (1, 2, 3, 8):
            parts = [parallel.clip_shard(n, r, world) for r in range(world)]
            flat = sorted(i for p in parts for i in p)
            assert flat == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    with pytest.raises(ValueError):
        parallel.clip_shard(4, 2, 2)
    assert parallel.max_over_ranks(3.5) == 3.5 and parallel.gather_to_rank0("x") == ["x"]


def _grad_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    parallel.init("gloo")
    torch.manual_seed(0)
    # the trainable set of a LoRA fine-tune: a few small factors; one frozen tensor; one factor without a gradient on rank 1
    ps = [torch.nn.Parameter(torch.zeros(4, 6)), torch.nn.Parameter(torch.zeros(8, 4)), torch.nn.Parameter(torch.zeros(3))]
    frozen = torch.nn.Parameter(torch.zeros(5), requires_grad=False)
    ps[0].grad = torch.full((4, 6), 1.0 + rank)
    ps[1].grad = torch.arange(32, dtype=torch.float32).reshape(8, 4) * (rank + 1)
    if rank == 0:
        ps[2].grad = torch.tensor([3.0, 6.0, 9.0])
    n = parallel.allreduce_gradients(ps + [frozen])
    if rank == 0:
        q.put((n, [p.grad.clone() for p in ps], frozen.grad))
    parallel.barrier()
    dist.destroy_process_group()


def test_gradient_allreduce_gloo():
    """The one collective of the fine-tune step: flat fp32 buffer, SUM, divided by the world size."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    n, grads, frozen_grad = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert n == 24 + 32 + 3
    assert torch.equal(grads[0], torch.full((4, 6), 1.5))
    assert torch.equal(grads[1], torch.arange(32, dtype=torch.float32).reshape(8, 4) * 1.5)
    assert torch.equal(grads[2], torch.tensor([1.5, 3.0, 4.5]))  # rank 1 had no gradient: zeros in the sum
    assert frozen_grad is None
