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


# ---- round 2: the code paths `bench.py --gpus N`, the sharded evaluation and the window-sharded video inference run on N GPUs ----
class _StubDepther:
    """infer_video_depth stand-in: a deterministic function of the clip's frames (no GPU, no model)."""

    def infer_video_depth(self, colors):
        f = colors.astype(np.float32).mean(axis=3) / 255.0
        return (0.1 + 0.8 * f).astype(np.float32)


class _StubRunner:
    """HipWindowRunner stand-in: window k's 32 maps are a deterministic function of its 32 input frames."""

    def __init__(self, frames):
        self.frames = frames
        self.calls = []

    def run(self, sources):
        out = []
        for idx in sources:
            self.calls.append(int(idx[10]))
            win = self.frames[idx].astype(np.float32).mean(axis=3) / 255.0
            out.append((0.2 + win * (1.0 + 0.01 * (idx[10] % 7))).astype(np.float32))
        return out


def _video_frames():
    from endodav_amd import synth

    return (synth.uniform("shard:video", (95, 12, 16, 3), 0.0, 1.0) * 255).astype(np.uint8)  # 95 frames -> 5 windows


def _r2_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from endodav_amd import evaluate as ev
    from endodav_amd import video

    parallel.init("gloo")
    # (1) bench.py's protocol: exactly `steps` timed steps, warm-up untimed, MAX over ranks
    seen = []

    def step(i):
        seen.append(i)
        if i >= 0 and rank == 1:
            import time
            time.sleep(0.02)  # rank 1 is the slow one
        return i

    dt, last = parallel.timed_region(step, steps=5, warmup=2, device=None)
    # (2) clip-sharded evaluation
    res = ev.evaluate_video(_StubDepther(), ev.SyntheticVideos(n_clips=5, n_frames=4, height=12, width=16), depth_align="scale_shift", device=None)
    # (3) one long video, windows sharded over the ranks
    frames = _video_frames()
    runner = _StubRunner(frames)
    out = video.infer_video_depth(None, frames, runner=runner, shard_windows=True)
    # (4) the in-place all-reduce on a flat gradient buffer whose slices are the .grad tensors
    flat = torch.arange(12, dtype=torch.float32) * (rank + 1)
    ps = [torch.nn.Parameter(torch.zeros(2, 4)), torch.nn.Parameter(torch.zeros(4))]
    ps[0].grad, ps[1].grad = flat[:8].view(2, 4), flat[8:].view(4)

    class M:
        def flat_gradients(self, params):
            return flat

    n = parallel.allreduce_gradients(ps, model=M())
    if rank == 0:
        q.put(dict(dt=dt, last=last, seen=seen, res=res, out=out, calls=runner.calls, n=n, flat=flat.clone(), g0=ps[0].grad.clone()))
    else:
        assert res is None and out is None
    parallel.barrier()
    dist.destroy_process_group()


def test_round2_sharded_paths_equal_the_one_rank_run():
    from endodav_amd import evaluate as ev
    from endodav_amd import video

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_r2_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # (1)
    assert got["seen"] == [-2, -1, 0, 1, 2, 3, 4] and got["last"] == 4
    assert got["dt"] >= 5 * 0.02  # rank 0 reports the slow rank's time
    # (2) the two-rank evaluation equals the one-rank one bit for bit (inference_times are wall clocks)
    one = ev.evaluate_video(_StubDepther(), ev.SyntheticVideos(n_clips=5, n_frames=4, height=12, width=16), depth_align="scale_shift", device=None)
    for k in ("errors", "temporal", "aligns", "ratios"):
        assert np.array_equal(got["res"][k], one[k]), k
    assert got["res"]["errors"].shape == (20, 7) and len(got["res"]["inference_times"]) == 5
    # (3)
    frames = _video_frames()
    solo = video.infer_video_depth(None, frames, runner=_StubRunner(frames))
    assert solo.shape == (95, 12, 16) and np.array_equal(got["out"], solo)
    assert got["calls"] == [10, 54, 94]  # rank 0 ran windows 0, 2, 4 (original frame shown in slot 10; 98 is padding -> the last frame)
    # (4)
    assert got["n"] == 12 and torch.equal(got["flat"], torch.arange(12, dtype=torch.float32) * 1.5)
    assert torch.equal(got["g0"], (torch.arange(8, dtype=torch.float32) * 1.5).view(2, 4))  # .grad is a view: reduced in place


class _VideoDepther:
    """What endodav.infer_video_depth does (endodav.py:758): delegate to video.infer_video_depth -- here with the stub runner instead of the GPU one."""

    def infer_video_depth(self, colors):
        from endodav_amd import video

        return video.infer_video_depth(None, colors, runner=_StubRunner(colors))


def _nested_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from endodav_amd import evaluate as ev

    parallel.init("gloo")
    # clips of DIFFERENT lengths on the two ranks (40 / 70 / 25 frames -> 2 / 4 / 2 windows): a window-level collective inside the
    # clip-sharded loop would pair up gathers of different clips (TypeError on rank 1, 'Connection closed by peer' on rank 0)
    res = ev.evaluate_video(_VideoDepther(), _RaggedVideos(), depth_align="scale_shift", device=None)
    if rank == 0:
        q.put(res)
    else:
        assert res is None
    parallel.barrier()
    dist.destroy_process_group()


class _RaggedVideos:
    def __init__(self):
        from endodav_amd import evaluate as ev

        self.clips = [ev.SyntheticVideos(n_clips=1, n_frames=n, height=12, width=16, seed=7 + i)[0] for i, n in enumerate((40, 70, 25))]

    def __len__(self):
        return len(self.clips)

    def __getitem__(self, i):
        return self.clips[i]


def test_clip_sharded_evaluation_of_a_windowing_depther_equals_the_one_rank_run():
    """evaluate_video (clips over ranks) around a depther that runs video.infer_video_depth (windows): the two sharding layers must not nest."""
    from endodav_amd import evaluate as ev

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_nested_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    one = ev.evaluate_video(_VideoDepther(), _RaggedVideos(), depth_align="scale_shift", device=None)
    for k in ("errors", "temporal", "aligns", "ratios"):
        assert np.array_equal(got[k], one[k]), k
    assert got["errors"].shape[0] == 40 + 70 + 25


def test_window_sources_match_the_reference_windows():
    """Padding + key-frame substitution as pure index arithmetic (endodav.py:185-199), against the per-frame means of the window inputs
    the reference's own infer_video_depth built (tests/golden/video_stitch.npz)."""
    from endodav_amd import synth, video
    from tests import helpers as H
    from tests.golden.make_golden import VIDEO_CASE

    g = H.load_golden("video_stitch")
    n, h, w = VIDEO_CASE["n_frames"], VIDEO_CASE["h"], VIDEO_CASE["w"]
    frames = (synth.uniform("video:frames", (n, h, w, 3), 0.0, 1.0) * 255).astype(np.uint8)
    means = (frames.astype(np.float32) / 255.0).astype(np.float64).mean(axis=(1, 2, 3))
    src = video.window_sources(n)
    assert len(src) == g["window_input_means"].shape[0]
    for k, idx in enumerate(src):
        assert idx.shape == (32,) and idx.max() <= n - 1
        assert np.abs(means[idx] - g["window_input_means"][k]).max() < 1e-6
    # slot 0 of window 2 = slot 6 of window 1 = slot 26 of window 0
    assert src[1][0] == src[0][6] and (len(src) < 3 or src[2][0] == src[1][6] == src[0][26])
