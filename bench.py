#!/usr/bin/env python3
"""bench.py — frames/s of EndoDAV's per-clip forward on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--T 8] [--encoder vits]

One "step" = one pass of the hot path (edv_forward) over one synthetic 518x518 clip of T frames that is
already resident in HBM.  N > 1: one process per GPU (torch.distributed.run), every rank runs its own
clip per step, no data-path collective (clips are independent: SURVEY.md §8e) — weak scaling; the only
NCCL(RCCL) traffic is the barrier and the MAX-reduction of the elapsed time around the timed region.

Rank 0 prints ONE JSON line: value = total frames of all ranks / max-over-ranks time, plus
  roofline      the dominant kernel (spatial attention, fp32 MFMA) priced as algorithmic FLOPs per launch over
                its mean launch time, measured with HIP event pairs on the launch stream inside the timed region
  cpu_baseline  the oracle (PyTorch-CPU restatement, kind "port") timed on this box's host cores on a bounded
                sample of the same workload (rank 0, N = 1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

VITS = dict(encoder="vits", features=64, out_channels=[48, 96, 192, 384])
VITB = dict(encoder="vitb", features=128, out_channels=[96, 192, 384, 768])
VITL = dict(encoder="vitl", features=256, out_channels=[256, 512, 1024, 1024])
MODELS = {"vits": VITS, "vitb": VITB, "vitl": VITL}
DIMS = {"vits": (384, 12, 6), "vitb": (768, 12, 12), "vitl": (1024, 24, 16)}
# matmul+conv FLOPs per 518x518 frame, FlopCounterMode over the reference
GFLOP_PER_FRAME = {"vits": 121.1, "vitb": 403.9, "vitl": 1403.8}
LIN_STEPS = 2  # timed steps whose dense-GEMM launches are bracketed with HIP events (see main)
PEAK_F32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--T", type=int, default=8, help="frames per clip (BASELINE headline: 8)")
    ap.add_argument("--encoder", default="vits", choices=sorted(MODELS))
    ap.add_argument("--image", type=int, default=518)
    ap.add_argument("--clips", type=int, default=1, help="clips per GPU per step (a batch [B,T,...] through one forward)")
    ap.add_argument("--conv-head", action="store_true", help="the four HeadDepth heads instead of the VDA head (reference default; with --train "
                    "their convolutions are trainable next to the LoRA factors, endodav/layers.py:5-34)")
    ap.add_argument("--lora", default="dvlora", choices=["none", "lora", "dvlora", "ssb"], help="lora_type (the reference's train_video*.sh use ssb)")
    ap.add_argument("--temporal-lora", action="store_true", help="LoRA on ff.net.2 of the motion modules too (--temporal_lora)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true", help="do not bracket the dominant kernel with HIP events")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU baseline (0 = min(16, cores): the box's CPU share)")
    ap.add_argument("--train", action="store_true", help="time the fine-tune step instead (forward + loss + HIP backward + gradient "
                    "all-reduce + AdamW on the LoRA factors; BASELINE.json config 4 shape with --encoder vitb --T 16); not the headline metric")
    return ap.parse_args()


def measured_traffic(encoder, T, image, clips=1, which="gemm"):
    """HBM-side bytes per launch of a kernel from the committed rocprofv3 --pmc passes (collected in their own runs,
    FETCH_SIZE doubled as the gfx950 guide prescribes); None when no pass matches this workload."""
    path = os.path.join(ROOT, "profiles", f"r01_{which}_traffic.json")
    try:
        with open(path) as f:
            rec = json.load(f)
    except OSError:
        return None
    if rec.get("config") == {"encoder": encoder, "T": T, "image": image} and clips == 1:
        return rec["traffic_bytes_per_launch"]
    return None


def cpu_baseline(kwargs, T, image, threads):
    """Oracle on the host cores: 1 warm-up clip + timed clips until ~20 s or 3 clips."""
    import torch

    import endodav_amd
    from endodav_amd import synth
    from oracle import endodav_oracle as orc

    cores = threads or min(16, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    model = endodav_amd.endodav(**kwargs, image_shape=(image, image), lora_type="dvlora", disable_conv_head=True).eval()
    synth.fill_module_(model)
    sd = {k: v.detach() for k, v in model.state_dict().items()}
    cfg = orc.OracleConfig(encoder=kwargs["encoder"], image_shape=(image, image), lora_type="dvlora", disable_conv_head=True)
    x = torch.from_numpy(synth.synth_clip(1, T, image, image, seed=0))
    with torch.no_grad():
        orc.forward(sd, x, cfg)  # warm-up
        n, t0 = 0, time.perf_counter()
        while n < 3 and (n == 0 or time.perf_counter() - t0 < 20.0):
            orc.forward(sd, x, cfg)
            n += 1
        dt = time.perf_counter() - t0
    return {"value": round(n * T / dt, 3), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{n} clip(s) of T={T} at {image}x{image} after 1 warm-up clip, oracle/endodav_oracle.py (torch {torch.__version__} CPU fp32)"}


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    import endodav_amd
    from endodav_amd import parallel, synth

    rank, world, local = parallel.env_rank_world()
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N > 1 with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N "
                             "--master-addr 127.0.0.1 --master-port P bench.py --gpus N ...")
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    parallel.init("nccl", dev)

    kwargs = MODELS[args.encoder]
    T, S = args.T, args.image
    model = endodav_amd.endodav(**kwargs, image_shape=(S, S), lora_type=args.lora, temporal_lora=args.temporal_lora,
                                disable_conv_head=not args.conv_head).eval()
    synth.fill_module_(model)
    model = model.to(dev)
    Bc = args.clips
    x = torch.from_numpy(synth.synth_clip(Bc, T, S, S, seed=rank)).to(dev)  # resident in HBM before timing

    def sync_all():
        torch.cuda.synchronize(dev)
        parallel.barrier()

    if args.train:
        return train_bench(args, model, x, dev, rank, world, kwargs, sync_all)
    with torch.no_grad():
        for _ in range(max(args.warmup, 1)):
            out = model(x)
        sync_all()
        if not args.no_kernel_events:
            model.profile_enable([])
        sync_all()
        # Kernel timing happens inside the timed region, in its last LIN_STEPS steps: there every dense-GEMM launch and every
        # attention call is bracketed with a HIP event pair on its launch stream, and the encoder runs single-stream so that a
        # bracket times the kernel alone (by default the engine overlaps two frame groups on internal streams for short
        # clips, which is what the other steps run).  Event pairs around ~100 launches cost ~5 % of those steps.
        lin_from = args.steps - min(LIN_STEPS, args.steps)
        t0 = time.perf_counter()
        for i in range(args.steps):
            if i == lin_from and not args.no_kernel_events:
                model.set_encoder_streams(1)
                model.profile_set(["attn_spatial", "linear"])
            out = model(x)
        torch.cuda.synchronize(dev)
        parallel.barrier()
        dt = time.perf_counter() - t0
        if not args.no_kernel_events:
            model.set_encoder_streams(-1)
    dt = parallel.max_over_ranks(dt, dev)

    # Dominant kernel by time (profiles/r01_i_bench_T8_kernel_stats.csv): gemm_dma_kernel, the dense F.linear / 1x1-conv GEMM
    # (49 % of the step over its two epilogue variants); second: the encoder attention call (20 %).  Both are bracketed
    # with HIP event pairs on their launch stream inside the timed region; the engine accounts the algorithmic work of the
    # bracketed linear launches itself (2 M N K and A, W, C (+ residual) once).
    roofline = roofline_attn = None
    if not args.no_kernel_events:
        n_h, ms_h = model.profile_read("linear")          # patch embed + DPT head
        fl_h, by_h = model.profile_work("linear")
        n_e, ms_e = model.profile_read("linear_encoder")  # qkv / proj / fc1 / fc2 of the encoder blocks
        fl_e, by_e = model.profile_work("linear_encoder")
        n_l, ms_l, fl_l, by_l = n_h + n_e, ms_h + ms_e, fl_h + fl_e, by_h + by_e
        if n_l > 0 and ms_l > 0:
            achieved = fl_l / (ms_l * 1e-3) / 1e12
            roofline = {"kernel": "gemm_dma_kernel (every F.linear / 1x1 conv of the step: qkv, proj, fc1, fc2 of the 12 blocks + the head's)",
                        "bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(achieved / PEAK_F32_MFMA_TFLOPS, 4), "traffic": measured_traffic(args.encoder, T, S, Bc, "gemm"),
                        "traffic_unit": "bytes per launch, mean over the step's launches (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, profiles/r01_gemm_traffic.json)",
                        "algorithmic_bytes_per_launch": round(by_l / n_l, 1), "launches": n_l, "avg_launch_ms": round(ms_l / n_l, 4),
                        "flop_per_launch": round(fl_l / n_l, 1), "peak_dtype": "f32 MFMA (v_mfma_f32_32x32x2_f32), dense"}
            if n_e > 0 and ms_e > 0:  # the same kernel on the encoder's four shapes only (the head's small GEMMs are HBM- / launch-bound)
                ach_e = fl_e / (ms_e * 1e-3) / 1e12
                roofline["encoder_launches"] = {"achieved": round(ach_e, 2), "frac": round(ach_e / PEAK_F32_MFMA_TFLOPS, 4), "launches": n_e,
                                                "avg_launch_ms": round(ms_e / n_e, 4), "flop_per_launch": round(fl_e / n_e, 1),
                                                "share_of_linear_flop": round(fl_e / fl_l, 4)}
        n, ms = model.profile_read("attn_spatial")
        D, depth, heads = DIMS[args.encoder]
        ntok = (S // 14) ** 2 + 1
        flops = 4.0 * ntok * ntok * 64 * heads * T * Bc  # QK^T + PV, 2 FLOP per MAC, per launch (one encoder block, all frames)
        if n > 0 and ms > 0:
            achieved = flops / (ms / n * 1e-3) / 1e12
            roofline_attn = {"kernel": "attn_spatial_kernel + attn_combine_kernel (one encoder-block attention call)", "bound": "mfma",
                             "achieved": round(achieved, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                             "frac": round(achieved / PEAK_F32_MFMA_TFLOPS, 4), "traffic": measured_traffic(args.encoder, T, S, Bc, "attn"),
                             "traffic_unit": "bytes per call (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, profiles/r01_attn_traffic.json)",
                             "algorithmic_bytes_per_launch": 4.0 * ntok * T * Bc * heads * 64 * 4,
                             "launches": n, "avg_launch_ms": round(ms / n, 4), "flop_per_launch": flops,
                             "peak_dtype": "f32 MFMA (v_mfma_f32_32x32x2_f32), dense"}
    # PCIe-inclusive variant (never `value`): pinned host clip -> HBM, forward, the four maps -> pinned host
    pcie_value = None
    if world == 1:
        model.profile_enable([])
        xh = x.cpu().pin_memory()
        oh = [torch.empty_like(v, device="cpu").pin_memory() for v in out.values()]
        with torch.no_grad():
            torch.cuda.synchronize(dev)
            n_p = max(args.steps // 2, 1)
            t1 = time.perf_counter()
            for _ in range(n_p):
                xd = xh.to(dev, non_blocking=True)
                o = model(xd)
                for h, v in zip(oh, o.values()):
                    h.copy_(v, non_blocking=True)
            torch.cuda.synchronize(dev)
            pcie_value = Bc * T * n_p / (time.perf_counter() - t1)
    finite = bool(torch.isfinite(out[("disp", 0)]).all().item())
    if rank == 0:
        frames = world * Bc * T * args.steps
        value = frames / dt
        line = {
            "metric": "depth frames/sec (518x518, T=8 clip)" if (S, T) == (518, 8) else f"depth frames/sec ({S}x{S}, T={T} clip)",
            "value": round(value, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"ViT-{args.encoder[-1].upper()} endodav (features {kwargs['features']}, out_channels {kwargs['out_channels']}, "
                                   f"{args.lora} r=4, {'conv head' if args.conv_head else 'VDA head'}), {Bc} synthetic {S}x{S} T={T} clip(s) per GPU per step (BASELINE.json configs[1] shape), "
                                   "hash-initialised weights", "encoder": args.encoder, "T": T, "image": [S, S], "clips_per_gpu_per_step": Bc,
                       "parallelism": f"clip-sharded x{world}, no data-path collective"},
            "model_tflop_per_clip": round(GFLOP_PER_FRAME[args.encoder] * T / 1e3 * (S / 518.0) ** 2, 4),
            "model_tflops": round(GFLOP_PER_FRAME[args.encoder] * (S / 518.0) ** 2 * value / 1e3, 2),
            "launches_per_step": model.launch_count(), "device_mem_mb": round(model.device_bytes() / 2 ** 20, 1), "output_finite": finite,
            "pcie_inclusive_value": None if pcie_value is None else round(pcie_value, 2),
            "roofline": roofline, "roofline_attention": roofline_attn,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(kwargs, T, S, args.cpu_threads)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    if world > 1:
        parallel.barrier()
        dist.destroy_process_group()


def train_bench(args, model, x, dev, rank, world, kwargs, sync_all):
    """One fine-tune step per iteration: HIP forward keeping activations, a PyTorch loss on the four disparity maps (the
    reference's losses stay PyTorch, north_star), HIP backward to the LoRA factors, ONE all-reduce of the flat gradient
    buffer, AdamW.  One clip per GPU per step (the reference's DataParallel split, trainer_end_to_end_video.py:731)."""
    import torch
    import torch.distributed as dist

    import endodav_amd
    from endodav_amd import parallel

    endodav_amd.mark_only_part_as_trainable(model, warm_up=True)
    params = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.AdamW(params, lr=1e-4)
    T, S = args.T, args.image

    def step():
        opt.zero_grad(set_to_none=True)
        out = model(x)
        loss = sum((o - o.detach().mean()).abs().mean() for o in out.values())  # stand-in for the photometric loss: an L1 on every scale
        loss.backward()
        n = parallel.allreduce_gradients(params)
        opt.step()
        return loss, n

    for _ in range(max(args.warmup, 1)):
        loss, nred = step()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, nred = step()
    torch.cuda.synchronize(dev)
    parallel.barrier()
    dt = parallel.max_over_ranks(time.perf_counter() - t0, dev)
    if rank == 0:
        frames = world * args.clips * T * args.steps
        line = {
            "metric": f"fine-tune frames/sec ({S}x{S}, T={T} clip, LoRA factors trainable)", "value": round(frames / dt, 2), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"ViT-{args.encoder[-1].upper()} endodav fine-tune step ({args.lora}{' + temporal_lora' if args.temporal_lora else ''} r=4, {'conv head, conv_depth_* trainable' if args.conv_head else 'VDA head'}): forward + L1 stand-in loss + HIP "
                                   f"backward + gradient all-reduce + AdamW, {args.clips} synthetic {S}x{S} T={T} clip(s) per GPU per step",
                       "encoder": args.encoder, "T": T, "image": [S, S], "clips_per_gpu_per_step": args.clips,
                       "parallelism": f"data-parallel x{world}: one all-reduce of {nred} gradient floats per step"},
            "trainable_floats": nred, "loss": float(loss.item()), "device_mem_mb": round(model.device_bytes() / 2 ** 20, 1),
            "roofline": None, "cpu_baseline": None,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        parallel.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
