"""Clip sharding across GPUs: one process per GPU.  Inference has no data-path collective; the
fine-tune step has exactly one: an all-reduce of the trainable gradients (``allreduce_gradients``).

``forward`` never mixes clips (the encoder treats B*T as a flat batch and the motion modules attend
within a clip only: dpt_pyramid.py:71-74, motion_module.py:232), and the windows of a long video depend on
each other only through *input* key frames (endodav.py:198-199), so inference partitions over independent
units.  The only communication is the bookkeeping around a timed region (barrier, MAX of elapsed time) and
the final gather of results to rank 0 — RCCL on GPUs (backend "nccl"), gloo in the CPU tests.
"""
from __future__ import annotations

import os
import time
from typing import Any, Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def init(backend: str, device: Optional[torch.device] = None) -> None:
    """Join the default process group described by RANK / WORLD_SIZE / MASTER_* (no-op for one process)."""
    rank, world, _ = env_rank_world()
    if world == 1 or dist.is_initialized():
        return
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    kwargs = {"device_id": device} if (backend == "nccl" and device is not None) else {}
    dist.init_process_group(backend, rank=rank, world_size=world, **kwargs)


def clip_shard(n_units: int, rank: int, world: int) -> List[int]:
    """Round-robin assignment of independent units (clips / windows) to ranks."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    return list(range(rank, n_units, world))


def barrier() -> None:
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def max_over_ranks(value: float, device: Optional[torch.device] = None) -> float:
    """MAX-reduction of a host scalar (the elapsed time of a timed region)."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return float(value)
    if dist.get_backend() != "nccl":
        device = None  # gloo (CPU tests, single-GPU rehearsals): a host tensor
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_to_rank0(obj: Any) -> Optional[List[Any]]:
    """Collect one picklable object per rank on rank 0 (per-clip numpy results); None elsewhere."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return [obj]
    out = [None] * dist.get_world_size() if dist.get_rank() == 0 else None
    dist.gather_object(obj, out, dst=0)
    return out


def merge_shards(shards: Sequence[Sequence[Any]], n_units: int) -> List[Any]:
    """Inverse of ``clip_shard``: interleave the per-rank result lists back into unit order."""
    world = len(shards)
    out: List[Any] = [None] * n_units
    for r, items in enumerate(shards):
        idx = clip_shard(n_units, r, world)
        if len(items) != len(idx):
            raise ValueError(f"rank {r} returned {len(items)} results for {len(idx)} units")
        for i, v in zip(idx, items):
            out[i] = v
    return out


def allreduce_gradients(params: Sequence[torch.nn.Parameter], model: Optional[torch.nn.Module] = None) -> int:
    """Data-parallel fine-tune step, one clip per rank (the reference scatters dim 0 of [batch, T, ...] over
    ``nn.DataParallel`` replicas, trainer_end_to_end_video.py:731, and sums their gradients): ONE all-reduce over a flat
    fp32 buffer of every trainable gradient -- a few MB of LoRA factors -- summed, then divided by the world size.

    With ``model`` (an ``endodav_amd.endodav`` whose backward has just run) the gradients already live in one contiguous buffer
    that the engine wrote (``edv_grad_bind_flat``) and every ``p.grad`` is a view of it: the all-reduce and the division run on that
    buffer in place, with no copy.  Otherwise (any module, or gradients that are not such views) the gradients are packed into a
    fresh flat buffer and copied back; a parameter without a gradient on this rank contributes zeros.
    Returns the number of floats reduced."""
    params = [p for p in params if p.requires_grad]
    if not params:
        return 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    n = sum(p.numel() for p in params)
    flat = model.flat_gradients(params) if (model is not None and hasattr(model, "flat_gradients")) else None
    if flat is not None:
        if world > 1:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
            flat /= world
        return n
    if world == 1:
        return n
    flat = torch.zeros(n, dtype=torch.float32, device=params[0].device)
    off = 0
    for p in params:
        if p.grad is not None:
            flat[off:off + p.numel()].copy_(p.grad.reshape(-1))
        off += p.numel()
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat /= world
    off = 0
    for p in params:
        g = flat[off:off + p.numel()].view_as(p)
        if p.grad is None:
            p.grad = g.clone()
        else:
            p.grad.copy_(g)
        off += p.numel()
    return n


def timed_region(step: Callable[[int], Any], steps: int, warmup: int, device: Optional[torch.device] = None) -> Tuple[float, Any]:
    """The measurement protocol of ``bench.py`` for any number of ranks: ``warmup`` untimed calls of ``step(i)``, then EXACTLY ``steps``
    timed calls bracketed on both sides by a device synchronisation + barrier, and the MAX over ranks of the elapsed wall time.
    Returns (seconds, the last step's result).  ``device`` None = a CPU run (the gloo tests): no device synchronisation."""
    def fence() -> None:
        if device is not None and device.type == "cuda":
            torch.cuda.synchronize(device)
        barrier()

    out = None
    for i in range(warmup):
        out = step(i - warmup)
    fence()
    t0 = time.perf_counter()
    for i in range(steps):
        out = step(i)
    fence()
    return max_over_ranks(time.perf_counter() - t0, device), out


def rank_world() -> Tuple[int, int]:
    """(rank, world size) of the default process group; (0, 1) without one."""
    if dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_run(n_units: int, fn: Callable[[int], Any]) -> Optional[List[Any]]:
    """Run ``fn(unit)`` for this rank's round-robin share of ``n_units`` independent units (clips of a dataset, windows of a long
    video) and merge every rank's results back into unit order on rank 0 (None on the other ranks).  No data-path collective:
    the only communication is the final gather of the (picklable, host-side) results."""
    rank, world = rank_world()
    mine = [fn(u) for u in clip_shard(n_units, rank, world)]
    shards = gather_to_rank0(mine)
    return None if shards is None else merge_shards(shards, n_units)
