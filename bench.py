#!/usr/bin/env python3
"""bench.py — frames/s of EndoDAV's per-clip forward on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--T 8] [--encoder vits] [--train]

With --gpus N > 1 and no WORLD_SIZE in the environment the script starts its N ranks itself (self_launch: a torch.distributed.run child, before
any GPU call here); under `python -m torch.distributed.run ... bench.py --gpus N` it is one of the ranks.

One "step" = one pass of the hot path (edv_forward) over one synthetic 518x518 clip of T frames that is
already resident in HBM.  N > 1: one process per GPU (torch.distributed.run), every rank runs its own
clip per step, no data-path collective (clips are independent: SURVEY.md §8e) — weak scaling; the only
NCCL(RCCL) traffic is the barrier and the MAX-reduction of the elapsed time around the timed region
(``endodav_amd.parallel.timed_region``, the same code the gloo world-2 test runs).

Rank 0 prints ONE JSON line: value = total frames of all ranks / max-over-ranks time, plus
  precision           what "dtype f32" means in the run's products mode (model.products; --products): fp32 tensors / accumulation / softmax / norms / head
                      always; in the default mode ("bf16x6") the encoder's linears and attention form each fp32 product as six bf16 MFMAs on three-term
                      bf16 splits of both operands (error against fp64 no worse than the fp32 matrix pipe's: tests/test_gemm_x6_gpu.py, test_attn_x6_gpu.py)
  other_products      the OTHER products mode timed in the same process on the same clip (value, and how far the two modes' disparities are apart)
  roofline            the dominant kernel by time, priced as FLOP over kernel time measured with HIP event pairs inside the dispatches in the first steps
                      of the timed region: default mode -- gemm_x6_kernel, EXECUTED bf16 FLOP (6 x algorithmic) against the dense bf16 MFMA peak, with the
                      fp32-equivalent rate against the fp32 MFMA peak beside it and the head's fp32 GEMMs as `head_linears_fp32`; --products f32 -- the
                      dense fp32-MFMA GEMM class (every F.linear / 1x1 conv) against the fp32 MFMA peak, its encoder launches as `encoder_launches`
  roofline_attention  the second one: the encoder's fused spatial attention, same method (attn_x6_kernel / attn_lean_kernel)
  roofline_hbm        the bandwidth-bound kernels (LayerNorm, GroupNorm, bilinear resamples, GEGLU, the final 1x1 convolution,
                      patchify): algorithmic bytes over launch time against the 8 TB/s HBM3E peak
  cpu_baseline        the oracle (PyTorch-CPU restatement, kind "port") timed on this box's host cores on a bounded
                      sample of the same workload (rank 0, N = 1 only): median of 5 clips at n = the box's cores and at n = 8.
--train times the fine-tune step (BASELINE.json config 4 with --encoder vitb --T 16 --image 224x280): forward + the trainer's loss
(trainer_end_to_end_video.py:808-971 with synthetic side-network outputs, fused in HIP: edv_trainer_loss; --torch-loss = the same as eager PyTorch
ops, --l1-loss = the round-1 stand-in) + HIP backward + ONE in-place all-reduce of the flat gradient buffer + AdamW + weight refresh.
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
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
LIN_STEPS = 2  # timed steps (the first ones) whose kernels are bracketed with HIP events (see main)
PRECISION_NOTE = {
    "f32": "fp32 tensors, fp32 accumulation, fp32 products (v_mfma_f32_32x32x2_f32) everywhere",
    "bf16x6": "fp32 tensors, fp32 accumulation and fp32 softmax / norms / head everywhere; the encoder's linears and attention compute each fp32 product as six "
              "bf16 MFMAs on three-term bf16 splits of BOTH operands (dropped cross terms <= 2^-26 of a product, below fp32's unit roundoff): error against "
              "fp64 no worse than the fp32-MFMA kernels' (tests/test_gemm_x6_gpu.py, tests/test_attn_x6_gpu.py), every parity test passes in this mode, and "
              "`other_products` times the all-fp32-products mode in the same run",
}
PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense, MI355X_MICROARCH.md (never the 2:1-sparsity figure)
PEAK_F32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_HBM_GBS = 8000.0         # same guide, "HBM3E peak BW" (spec; 6.29 TB/s measured with a float4 copy)
PROFILE_ROUND = "r03"         # profiles/<round>_{gemm,attn,hbm}_traffic.json hold the PMC traffic of this round's kernels


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--T", type=int, default=8, help="frames per clip (BASELINE headline: 8)")
    ap.add_argument("--encoder", default="vits", choices=sorted(MODELS))
    ap.add_argument("--image", default="518", help="image_shape: S (square) or HxW, e.g. 224x280 (the trainer's); frames come in at that size "
                    "(--train with 224x280: at the trainer's 256x320)")
    ap.add_argument("--clips", type=int, default=1, help="clips per GPU per step (a batch [B,T,...] through one forward)")
    ap.add_argument("--conv-head", action="store_true", help="the four HeadDepth heads instead of the VDA head (reference default; with --train "
                    "their convolutions are trainable next to the LoRA factors, endodav/layers.py:5-34)")
    ap.add_argument("--lora", default="dvlora", choices=["none", "lora", "dvlora", "ssb"], help="lora_type (the reference's train_video*.sh use ssb)")
    ap.add_argument("--temporal-lora", action="store_true", help="LoRA on ff.net.2 of the motion modules too (--temporal_lora)")
    ap.add_argument("--in-flight", type=int, default=0, help="clips in flight on the GPU (pipeline.ClipsInFlight): 0 = auto_depth (3 at the headline shape, "
                    "1 for clips that fill the part alone), 1 = one clip at a time")
    ap.add_argument("--products", default=None, choices=["f32", "bf16x6"], help="arithmetic of the encoder's linears and attention in inference (model.products; "
                    "default: the model's, bf16x6 since round 3): bf16x6 = fp32 tensors and accumulation, each product as six bf16 MFMAs on three-term bf16 "
                    "splits of both operands (error no worse than the fp32 pipe's: tests/test_gemm_x6_gpu.py, tests/test_attn_x6_gpu.py); f32 = fp32 products "
                    "on the fp32 matrix pipe everywhere.  The other mode is always timed beside `value` (other_products)")
    ap.add_argument("--no-other-products", action="store_true", help="skip the leg that times the other products mode (profiling passes: one arithmetic per run)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true", help="do not bracket kernels with HIP events")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU baseline (0 = min(16, cores): the box's CPU share)")
    ap.add_argument("--train", action="store_true", help="time the fine-tune step instead (forward + photometric loss + HIP backward + gradient "
                    "all-reduce + AdamW on the LoRA factors; BASELINE.json config 4 shape with --encoder vitb --T 16); not the headline metric")
    ap.add_argument("--l1-loss", action="store_true", help="--train with the round-1 L1 stand-in instead of the trainer's loss (A/B of the loss's share)")
    ap.add_argument("--torch-loss", action="store_true", help="--train with the trainer's loss as eager PyTorch ops (endodav_amd/losses.py::trainer_losses) "
                    "instead of the fused HIP kernels (edv_trainer_loss): what the loss costs on the host framework")
    ap.add_argument("--depth-consistency", action="store_true", help="--train with the depth reprojection / flow consistency terms on (the reference's "
                    "scripts: --depth_reproj 1e-2 / --depth_flow 1e-3 in the temporal tuning phase); default: the options' defaults, both off")
    args = ap.parse_args()
    hw = [int(v) for v in str(args.image).lower().split("x")]
    args.image_hw = (hw[0], hw[0]) if len(hw) == 1 else (hw[0], hw[1])
    return args


def measured_traffic(encoder, T, image_hw, clips=1, which="gemm", products="f32"):
    """HBM-side bytes per launch of a kernel class from the committed rocprofv3 --pmc passes (collected in their own runs, FETCH_SIZE doubled as the
    gfx950 guide prescribes); None when no pass matches this workload.  Round 3: profiles/r03_<config>_traffic.json (scratch/evidence_r03.sh) holds every
    class of one configuration; rounds 1-2 kept one file per class for the headline only."""
    import glob

    if image_hw[0] != image_hw[1] or clips != 1:
        return None, None
    want = {"encoder": encoder, "T": T, "image": image_hw[0]}
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", f"{PROFILE_ROUND}_*_traffic.json"))):
        try:
            with open(path) as f:
                rec = json.load(f)
        except (OSError, ValueError):
            continue
        cfg = dict(rec.get("config") or {})
        if cfg.pop("products", "f32") != products or cfg != want or "per_class" not in rec:
            continue
        pc = rec["per_class"]
        if which == "gemm" and pc.get("gemm"):
            return {"traffic_bytes_per_launch": pc["gemm"]["traffic_bytes_per_launch"]}, os.path.relpath(path, ROOT)
        if which == "gemm_x6" and pc.get("gemm_x6"):
            return {"traffic_bytes_per_launch": pc["gemm_x6"]["traffic_bytes_per_launch"]}, os.path.relpath(path, ROOT)
        if which == "attn" and pc.get("attn"):
            return {"traffic_bytes_per_launch": rec["attn_call_traffic_bytes"]}, os.path.relpath(path, ROOT)
        if which == "hbm":
            return {"traffic_bytes_per_launch": {k: (v["traffic_bytes_per_launch"] if v else None) for k, v in pc.items()}}, os.path.relpath(path, ROOT)
    if products != "f32":
        return None, None
    for rnd in ("r02", "r01"):
        path = os.path.join(ROOT, "profiles", f"{rnd}_{which}_traffic.json")
        try:
            with open(path) as f:
                rec = json.load(f)
        except OSError:
            continue
        if rec.get("config") == want:
            return rec, os.path.relpath(path, ROOT)
    return None, None


def cpu_baseline(kwargs, T, image_hw, threads):
    """Oracle on the host cores: 1 warm-up clip + 5 timed clips, median, at n = the box's cores and at n = 8."""
    import torch

    import endodav_amd
    from endodav_amd import synth
    from oracle import endodav_oracle as orc

    cores = threads or min(16, os.cpu_count() or 1)
    model = endodav_amd.endodav(**kwargs, image_shape=image_hw, lora_type="dvlora", disable_conv_head=True).eval()
    synth.fill_module_(model)
    sd = {k: v.detach() for k, v in model.state_dict().items()}
    cfg = orc.OracleConfig(encoder=kwargs["encoder"], image_shape=image_hw, lora_type="dvlora", disable_conv_head=True)
    x = torch.from_numpy(synth.synth_clip(1, T, image_hw[0], image_hw[1], seed=0))

    def run(n_threads, n_clips=5):
        torch.set_num_threads(n_threads)
        times = []
        with torch.no_grad():
            orc.forward(sd, x, cfg)  # warm-up
            for _ in range(n_clips):
                t0 = time.perf_counter()
                orc.forward(sd, x, cfg)
                times.append(time.perf_counter() - t0)
        return statistics.median(times), times

    med, times = run(cores)
    out = {"value": round(T / med, 3), "unit": "frames/s", "cores": cores, "kind": "port",
           "sample": f"median of {len(times)} clips of T={T} at {image_hw[0]}x{image_hw[1]} after 1 warm-up clip, oracle/endodav_oracle.py "
                     f"(torch {torch.__version__} CPU fp32)", "s_per_clip": round(med, 3)}
    if cores != 8 and (os.cpu_count() or 1) >= 8:
        med8, t8 = run(8)
        out["at_8_threads"] = {"value": round(T / med8, 3), "cores": 8, "s_per_clip": round(med8, 3), "clips": len(t8)}
    return out


def hbm_roofline(model, encoder, T, image_hw, clips):
    """Bandwidth-bound kernel classes bracketed in LIN_STEPS steps of the timed region: algorithmic bytes / launch time against the HBM peak."""
    from endodav_amd import _lib

    rec, src = measured_traffic(encoder, T, image_hw, clips, "hbm")
    rows = {}
    tot_b = tot_ms = 0.0
    for cls in _lib.HBM_CLASSES:
        n, ms = model.profile_read(cls)
        _, by = model.profile_work(cls)
        if n == 0 or ms <= 0:
            continue
        gbs = by / (ms * 1e-3) / 1e9
        rows[cls] = {"launches": n, "avg_launch_us": round(ms / n * 1e3, 2), "algorithmic_bytes_per_launch": round(by / n, 1),
                     "achieved": round(gbs, 1), "frac": round(gbs / PEAK_HBM_GBS, 4),
                     "traffic": None if rec is None else rec.get("traffic_bytes_per_launch", {}).get(cls)}
        tot_b += by
        tot_ms += ms
    if not rows:
        return None
    gbs = tot_b / (tot_ms * 1e-3) / 1e9
    return {"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4),
            "ms_per_step": round(tot_ms / LIN_STEPS, 4), "kernels": rows,
            "traffic_unit": None if rec is None else f"bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, {src})",
            "note": "algorithmic bytes = every tensor a kernel reads or writes, once; launch time by HIP events (kernel + the in-order gap before it); "
                    "groupnorm = statistics + apply launches of one call"}


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment: start the N ranks ourselves -- one child running
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py <the same arguments>`
    (the driver's own launch line) -- BEFORE this process has touched torch or the GPU, pass rank 0's JSON line through and exit with the
    child's status.  The reference's analogue is `--use_dp` starting its per-GPU threads itself (trainer_end_to_end_video.py:269-271,
    scripts/train_dp.sh)."""
    import socket
    import subprocess

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL needs it on this pool
    env.setdefault("OMP_NUM_THREADS", "1")
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, bufsize=1)
    for line in proc.stdout:  # stream through: the driver reads the ONE JSON line from our stdout
        sys.stdout.write(line)
        sys.stdout.flush()
    raise SystemExit(proc.wait())


def stub_bench(args, rank, world):
    """EDV_BENCH_STUB=1: the launch / rendezvous / timing protocol of this file with a sleep in place of the HIP step, on CPU over gloo --
    what tests/test_bench_launch_cpu.py runs to cover `--gpus N` starting itself.  The line says "stub": it is not a measurement."""
    import torch.distributed as dist

    from endodav_amd import parallel

    parallel.init(os.environ.get("EDV_BENCH_BACKEND", "gloo"), None)
    dt, _ = parallel.timed_region(lambda i: time.sleep(0.005), args.steps, max(args.warmup, 1), None)
    if rank == 0:
        print(json.dumps({"metric": "stub", "stub": True, "value": round(world * args.T * args.steps / dt, 2), "unit": "frames/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "scaling": "weak"}), flush=True)
    if world > 1:
        parallel.barrier()
        dist.destroy_process_group()


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)
    import torch
    import torch.distributed as dist

    import endodav_amd
    from endodav_amd import parallel, synth
    from endodav_amd.pipeline import ClipPipeline, ClipsInFlight

    rank, world, local = parallel.env_rank_world()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if os.environ.get("EDV_BENCH_STUB"):
        return stub_bench(args, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # Rehearsal of the N > 1 launch path on a box with ONE GPU (this pool): EDV_BENCH_SHARE_GPU=1 puts every rank on cuda:0 and
    # EDV_BENCH_BACKEND=gloo carries the barrier / MAX / gradient all-reduce.  The JSON line then says "rehearsal": true -- its value is not a
    # scaling number.  The driver's multi-GPU runs set neither: one GPU per rank, RCCL.
    backend = os.environ.get("EDV_BENCH_BACKEND", "nccl")
    if os.environ.get("EDV_BENCH_SHARE_GPU"):
        local = 0
    rehearsal = backend != "nccl" or bool(os.environ.get("EDV_BENCH_SHARE_GPU"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    parallel.init(backend, dev)

    kwargs = MODELS[args.encoder]
    T, (SH, SW) = args.T, args.image_hw
    model = endodav_amd.endodav(**kwargs, image_shape=(SH, SW), lora_type=args.lora, temporal_lora=args.temporal_lora,
                                disable_conv_head=not args.conv_head).eval()
    synth.fill_module_(model)
    model = model.to(dev)
    if args.products:
        model.products = args.products
    Bc = args.clips
    in_hw = (256, 320) if (args.train and (SH, SW) == (224, 280)) else (SH, SW)  # the trainer feeds 256x320 frames (options.py:128-135)
    x = torch.from_numpy(synth.synth_clip(Bc, T, in_hw[0], in_hw[1], seed=rank)).to(dev)  # resident in HBM before timing

    if args.train:
        return train_bench(args, model, x, dev, rank, world, kwargs, rehearsal)
    events = not args.no_kernel_events
    with torch.no_grad():
        model(x)  # creates the context (profile_* need one)
        if events:
            model.profile_enable([])
        # Kernel timing happens inside the timed region, in its FIRST LIN_STEPS steps: there every dense-GEMM launch, every attention call and every
        # bandwidth-bound kernel carries a HIP event pair inside its dispatch (edv_profile_enable: kernel start -> kernel end, the interval rocprofv3
        # reports), one clip at a time and the encoder on one stream, so that a bracket times its kernel alone.  From step LIN_STEPS on the brackets
        # are off and consecutive clips are in flight: they are independent, so the engine keeps several going (pipeline.ClipsInFlight: one context +
        # stream per lane, depth by auto_depth -- 3 at the headline shape, 1 for clips that fill the part alone).  Every step still is one whole
        # forward of one clip, submitted back to back; the closing synchronize of the timed region waits for all of them.  (Rounds 1-2 bracketed
        # the LAST steps; either way one drain separates the bracketed steps from the clips in flight.)
        lin_to = min(LIN_STEPS, args.steps) if events else 0
        out = {}
        n_flight = args.in_flight if args.in_flight > 0 else ClipsInFlight.auto_depth(model, Bc * T)
        flight = ClipsInFlight(model, dev, depth=n_flight) if n_flight > 1 else None
        handles = []

        def step(i):
            if events and i == 0:
                model.set_encoder_streams(1)
                model.profile_set(["attn_spatial", "linear"] + list(endodav_amd._lib.HBM_CLASSES))
            if events and i == lin_to:
                if flight is not None:
                    torch.cuda.synchronize(dev)  # the bracketed steps are still queued (the host runs ahead): clips in flight beside them would share
                                                 # the GPU with the kernels being timed (the first round-3 evidence pass did exactly that: 0.62 instead of 0.74)
                model.profile_set([])  # what was recorded stays
                model.set_encoder_streams(-1)
            if flight is not None and not (0 <= i < lin_to):
                handles.append(flight.submit(x, resident=True))  # x has been in HBM since before the timed region
                if len(handles) > n_flight:
                    handles.pop(0)
            else:
                out["maps"] = model(x)

        dt, _ = parallel.timed_region(step, args.steps, max(args.warmup, 1), dev)
        if events and args.steps <= lin_to:
            model.profile_set([])
            model.set_encoder_streams(-1)
        if handles:
            out["maps"] = handles[-1].result()
        serial_value = None
        if flight is not None:  # the same steps one clip at a time (reported beside `value`, never as it)
            n_s = max(args.steps // 2, 4)
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            for _ in range(n_s):
                model(x)
            torch.cuda.synchronize(dev)
            serial_value = Bc * T * n_s / (time.perf_counter() - t1)
        # The other products mode on the same clip, same clips in flight (reported beside `value`, never as it): how much the bf16 x 6 linears buy at
        # this shape, and how far the two modes' disparities are apart.
        other = None
        if world == 1 and not stub_alt(args):
            first = model.products
            model.products = "bf16x6" if first == "f32" else "f32"
            n_o = max(args.steps // 2, 4)
            run = (lambda: flight.submit(x, resident=True)) if flight is not None else (lambda: model(x))
            for _ in range(3):
                h = run()
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            for _ in range(n_o):
                h = run()
            torch.cuda.synchronize(dev)
            o_value = Bc * T * n_o / (time.perf_counter() - t1)
            o_maps = h.result() if flight is not None else h
            a, b = out["maps"][("disp", 0)], o_maps[("disp", 0)]
            other = {"products": model.products, "value": round(o_value, 2), "steps": n_o,
                     "max_abs_disparity_difference": float((a - b).abs().max()), "disparity_range": [float(a.min()), float(a.max())]}
            model.products = first
            model(x)  # contexts back in the first mode
    out = out["maps"]

    # Dominant kernel by time (profiles/r02_*_kernel_stats.csv): gemm_dma_kernel, the dense F.linear / 1x1-conv GEMM; second: the encoder
    # attention call.  Both are bracketed with HIP event pairs on their launch stream inside the timed region; the engine accounts the
    # algorithmic work of the bracketed linear launches itself (2 M N K and A, W, C (+ residual) once).
    roofline = roofline_attn = roofline_hbm = None
    if events:
        n_h, ms_h = model.profile_read("linear")          # patch embed + DPT head
        fl_h, by_h = model.profile_work("linear")
        n_e, ms_e = model.profile_read("linear_encoder")  # qkv / proj / fc1 / fc2 of the encoder blocks
        fl_e, by_e = model.profile_work("linear_encoder")
        n_l, ms_l, fl_l, by_l = n_h + n_e, ms_h + ms_e, fl_h + fl_e, by_h + by_e
        if n_l > 0 and ms_l > 0:
            achieved = fl_l / (ms_l * 1e-3) / 1e12
            rec, src = measured_traffic(args.encoder, T, (SH, SW), Bc, "gemm", model.products)
            roofline = {"kernel": "gemm_dma_kernel (every F.linear / 1x1 conv of the step: qkv, proj, fc1, fc2 of the encoder blocks + the head's)",
                        "bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(achieved / PEAK_F32_MFMA_TFLOPS, 4), "traffic": None if rec is None else rec["traffic_bytes_per_launch"],
                        "traffic_unit": None if rec is None else f"bytes per launch, mean over the step's launches (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, {src}; "
                                        "collected in separate --pmc passes of this command, not re-measured in this run)",
                        "algorithmic_bytes_per_launch": round(by_l / n_l, 1), "launches": n_l, "avg_launch_ms": round(ms_l / n_l, 4),
                        "flop_per_launch": round(fl_l / n_l, 1), "peak_dtype": "f32 MFMA (v_mfma_f32_32x32x2_f32), dense"}
            if n_e > 0 and ms_e > 0:  # the same kernel on the encoder's four shapes only (the head's small GEMMs are HBM- / launch-bound)
                ach_e = fl_e / (ms_e * 1e-3) / 1e12
                roofline["encoder_launches"] = {"achieved": round(ach_e, 2), "frac": round(ach_e / PEAK_F32_MFMA_TFLOPS, 4), "launches": n_e,
                                                "avg_launch_ms": round(ms_e / n_e, 4), "flop_per_launch": round(fl_e / n_e, 1),
                                                "share_of_linear_flop": round(fl_e / fl_l, 4)}
        if roofline is not None and model.products == "bf16x6" and n_e > 0 and ms_e > 0:
            # The dominant kernel is gemm_x6_kernel: the encoder's linears run their products on the bf16 pipe, six bf16 MFMA FLOP per algorithmic FLOP.
            # The physical bound is the dense bf16 MFMA peak, priced on the EXECUTED FLOP; the fp32-equivalent rate (what the model sees) is given
            # beside it against the fp32 pipe's peak -- above 1.0 there means faster than any fp32-MFMA kernel can be.  The head's fp32 GEMMs keep
            # their own object.
            t_e = ms_e * 1e-3
            ex = 6.0 * fl_e / t_e / 1e12
            rec, src = measured_traffic(args.encoder, T, (SH, SW), Bc, "gemm_x6", "bf16x6")
            head = dict(roofline)
            head.pop("encoder_launches", None)
            if n_h > 0 and ms_h > 0:
                ach_h = fl_h / (ms_h * 1e-3) / 1e12
                head.update({"kernel": "gemm_dma_kernel (patch embed + the DPT head's F.linear / 1x1 convs: fp32 products)", "achieved": round(ach_h, 2),
                             "frac": round(ach_h / PEAK_F32_MFMA_TFLOPS, 4), "launches": n_h, "avg_launch_ms": round(ms_h / n_h, 4),
                             "flop_per_launch": round(fl_h / n_h, 1), "algorithmic_bytes_per_launch": round(by_h / n_h, 1), "traffic": None, "traffic_unit": None})
            roofline = {"kernel": "gemm_x6_kernel (qkv, proj, fc1, fc2 of the encoder blocks: fp32 in / out / accumulate, products as six bf16 MFMAs on three-term "
                                  "bf16 splits of both operands)",
                        "bound": "mfma", "achieved": round(ex, 1), "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(ex / PEAK_BF16_MFMA_TFLOPS, 4),
                        "traffic": None if rec is None else rec["traffic_bytes_per_launch"],
                        "traffic_unit": None if rec is None else f"bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, {src}; separate --pmc passes of this command)",
                        "algorithmic_bytes_per_launch": round(by_e / n_e, 1), "launches": n_e, "avg_launch_ms": round(ms_e / n_e, 4),
                        "flop_per_launch": round(6.0 * fl_e / n_e, 1), "peak_dtype": "bf16 MFMA (v_mfma_f32_32x32x16_bf16), dense (never the 2:1-sparsity figure)",
                        "fp32_equivalent": {"achieved": round(fl_e / t_e / 1e12, 2), "frac_of_fp32_mfma_peak": round(fl_e / t_e / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                                            "flop_per_launch": round(fl_e / n_e, 1)},
                        "head_linears_fp32": head}
        n, ms = model.profile_read("attn_spatial")
        D, depth, heads = DIMS[args.encoder]
        ntok = (SH // 14) * (SW // 14) + 1
        flops = 4.0 * ntok * ntok * 64 * heads * T * Bc  # QK^T + PV, 2 FLOP per MAC, per launch (one encoder block, all frames)
        if n > 0 and ms > 0:
            achieved = flops / (ms / n * 1e-3) / 1e12
            rec, src = measured_traffic(args.encoder, T, (SH, SW), Bc, "attn")  # (the attention kernel is the same in both products modes)
            x6a = model.products == "bf16x6" and os.environ.get("EDV_X6_ATTN", "1") != "0" and ntok > 128
            peak_a = PEAK_BF16_MFMA_TFLOPS if x6a else PEAK_F32_MFMA_TFLOPS
            fp32_eq = achieved
            if x6a:
                achieved *= 6.0  # executed bf16 FLOP
                rec, src = measured_traffic(args.encoder, T, (SH, SW), Bc, "attn", "bf16x6")
            roofline_attn = {"kernel": ("attn_x6_kernel" if x6a else "attn_lean_kernel") + " + attn_combine_kernel (one encoder-block attention call)", "bound": "mfma",
                             "achieved": round(achieved, 2), "peak": peak_a, "unit": "TFLOP/s",
                             "frac": round(achieved / peak_a, 4), "traffic": None if rec is None else rec["traffic_bytes_per_launch"],
                             "traffic_unit": None if rec is None else f"bytes per call (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, {src})",
                             "algorithmic_bytes_per_launch": 4.0 * ntok * T * Bc * heads * 64 * 4,
                             "launches": n, "avg_launch_ms": round(ms / n, 4), "flop_per_launch": flops * (6.0 if x6a else 1.0),
                             "peak_dtype": "bf16 MFMA (v_mfma_f32_32x32x16_bf16), dense" if x6a else "f32 MFMA (v_mfma_f32_32x32x2_f32), dense"}
            if x6a:
                roofline_attn["fp32_equivalent"] = {"achieved": round(fp32_eq, 2), "frac_of_fp32_mfma_peak": round(fp32_eq / PEAK_F32_MFMA_TFLOPS, 4),
                                                    "flop_per_launch": flops}
        roofline_hbm = hbm_roofline(model, args.encoder, T, (SH, SW), Bc)
    # PCIe-inclusive variant (never `value`): pinned host clip -> HBM on a copy stream while the previous clip computes, forward, the four
    # maps -> pinned host memory on a second copy stream (endodav_amd/pipeline.py); the reference does both transfers synchronously
    pcie_value = None
    if world == 1:
        model.profile_enable([])
        pipe = ClipPipeline(model, dev)
        xh = x.cpu().pin_memory()
        n_p = max(args.steps, 4)
        for _ in pipe.run(xh for _ in range(3)):  # allocate the staging buffers
            pass
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        taken = sum(1 for _ in pipe.run(xh for _ in range(n_p)))
        torch.cuda.synchronize(dev)
        pcie_value = Bc * T * taken / (time.perf_counter() - t1)
    finite = bool(torch.isfinite(out[("disp", 0)]).all().item())
    if rank == 0:
        frames = world * Bc * T * args.steps
        value = frames / dt
        px = (SH * SW) / (518.0 * 518.0)
        line = {
            "metric": "depth frames/sec (518x518, T=8 clip)" if (SH, SW, T) == (518, 518, 8) else f"depth frames/sec ({SH}x{SW}, T={T} clip)",
            "value": round(value, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "precision": PRECISION_NOTE[model.products],
            "config": {"workload": f"ViT-{args.encoder[-1].upper()} endodav (features {kwargs['features']}, out_channels {kwargs['out_channels']}, "
                                   f"{args.lora} r=4, {'conv head' if args.conv_head else 'VDA head'}), {Bc} synthetic {SH}x{SW} T={T} clip(s) per GPU per step "
                                   "(BASELINE.json configs[1] shape at the defaults), hash-initialised weights", "encoder": args.encoder, "T": T,
                       "image": [SH, SW], "clips_per_gpu_per_step": Bc, "clips_in_flight": n_flight, "products": model.products,
                       "parallelism": f"clip-sharded x{world}, no data-path collective; {n_flight} consecutive clip(s) in flight per GPU (one engine context and HIP stream each)"},
            "model_tflop_per_clip": round(GFLOP_PER_FRAME[args.encoder] * T / 1e3 * px, 4),
            "model_tflops": round(GFLOP_PER_FRAME[args.encoder] * px * value / 1e3, 2),
            "launches_per_step": model.launch_count(), "device_mem_mb": round(model.device_bytes() / 2 ** 20, 1), "output_finite": finite,
            "one_clip_at_a_time_value": None if serial_value is None else round(serial_value, 2),
            "pcie_inclusive_value": None if pcie_value is None else round(pcie_value, 2),
            "other_products": other,
            "roofline": roofline, "roofline_attention": roofline_attn, "roofline_hbm": roofline_hbm,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(kwargs, T, (SH, SW), args.cpu_threads)
        else:
            line["cpu_baseline"] = None
        if rehearsal:
            line["rehearsal"] = f"{world} ranks on one GPU over {backend}: launch-path check, not a scaling number"
        print(json.dumps(line), flush=True)
    if world > 1:
        parallel.barrier()
        dist.destroy_process_group()


def stub_alt(args):
    """The alternative-products leg is skipped for batched clips (a [B, T] batch through one forward is its own configuration)."""
    return args.clips != 1 or args.no_other_products


def train_bench(args, model, x, dev, rank, world, kwargs, rehearsal=False):
    """One fine-tune step per iteration: HIP forward keeping activations, the photometric loss on the four disparity maps in PyTorch (the
    reference's losses stay PyTorch, north_star: endodav_amd/losses.py restates utils/layers.py's SSIM / back-projection / smoothness and
    the trainer's per-scale sum), HIP backward into ONE flat gradient buffer, ONE in-place all-reduce of it, AdamW.  One clip per GPU per
    step (the reference's DataParallel split, trainer_end_to_end_video.py:731)."""
    import torch
    import torch.distributed as dist

    import endodav_amd
    from endodav_amd import losses, parallel

    endodav_amd.mark_only_part_as_trainable(model, warm_up=True)
    model.train()
    params = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.AdamW(params, lr=1e-4)
    T, (SH, SW) = args.T, args.image_hw
    # The trainer's whole loss (losses.trainer_losses: masked reprojection against `refined`, loss_transform, get_smooth_bright, per-scale smoothness and,
    # with --depth-consistency, the two depth-consistency terms of the reference's scripts) on synthetic stand-ins for the side networks' outputs;
    # every tensor the reference's autograd reaches requires grad (poses, K / inv_K, refined, transform_high), so the fused call produces them all.
    N = x.shape[0] * x.shape[1]
    inp = losses.synthetic_trainer_inputs(N, x.shape[-2], x.shape[-1], device=dev, seed=rank)
    inp[("color", 0, 0)] = x.flatten(0, 1).contiguous()  # the clip the model sees is the loss's frame
    for s_ in range(1, 4):
        inp[("color", 0, s_)] = torch.nn.functional.interpolate(inp[("color", 0, 0)], [x.shape[-2] >> s_, x.shape[-1] >> s_], mode="bilinear", align_corners=False)
    for k in list(inp):
        if k in ("K", "inv_K") or (isinstance(k, tuple) and k[0] in ("cam_T_cam", "refined", "transform")):
            inp[k].requires_grad_(True)
    side_leaves = [v for v in inp.values() if v.requires_grad]
    wts = losses.TrainerLossWeights(depth_reproj=1e-2, depth_flow=1e-3, tune_temporal=True) if args.depth_consistency else losses.TrainerLossWeights()
    events = not args.no_kernel_events
    lin_from = args.steps - min(LIN_STEPS, args.steps)
    state = {}

    def loss_fn(out):
        if args.l1_loss:
            return sum((o - o.detach().mean()).abs().mean() for o in out.values())
        fn = losses.trainer_losses if args.torch_loss else losses.trainer_losses_hip
        return fn(out, inp, wts)["loss"]

    def step(i):
        if i == lin_from and events:
            model.profile_set(["linear", "attn_spatial", "attn_spatial_bwd"] + list(endodav_amd._lib.HBM_CLASSES))
        timed = events and i >= lin_from
        opt.zero_grad(set_to_none=True)
        for t_ in side_leaves:
            t_.grad = None
        out = model(x)
        if timed:
            e0, e1, e2 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        loss = loss_fn(out)
        if timed:
            e1.record()
        loss.backward()
        if timed:
            e2.record()
            state.setdefault("ev", []).append((e0, e1, e2))
        state["n"] = parallel.allreduce_gradients(params, model=model)
        opt.step()
        state["loss"] = loss

    model(x)  # creates the context
    model.profile_enable([])
    dt, _ = parallel.timed_region(step, args.steps, max(args.warmup, 1), dev)
    roofline = loss_share = roofline_hbm = None
    roofline_attn = {}
    if events:
        n_h, ms_h = model.profile_read("linear")
        fl_h, by_h = model.profile_work("linear")
        n_e, ms_e = model.profile_read("linear_encoder")
        fl_e, by_e = model.profile_work("linear_encoder")
        n_l, ms_l, fl_l = n_h + n_e, ms_h + ms_e, fl_h + fl_e
        if n_l > 0 and ms_l > 0:
            ach = fl_l / (ms_l * 1e-3) / 1e12
            roofline = {"kernel": "gemm_dma_kernel (every dense GEMM of the step: the forward's F.linear / 1x1 convs and the backward's input-gradient GEMMs)",
                        "bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4),
                        "traffic": None, "launches": n_l, "avg_launch_ms": round(ms_l / n_l, 4), "flop_per_launch": round(fl_l / n_l, 1),
                        "algorithmic_bytes_per_launch": round((by_h + by_e) / n_l, 1), "ms_per_step": round(ms_l / LIN_STEPS, 3),
                        "peak_dtype": "f32 MFMA (v_mfma_f32_32x32x2_f32), dense"}
        D_, depth_, heads_ = DIMS[args.encoder]
        ntok = (SH // 14) * (SW // 14) + 1
        fl_att = 4.0 * ntok * ntok * 64 * heads_ * T * args.clips  # forward: QK^T + PV per encoder block
        for cls_, mult, label in (("attn_spatial", 1.0, "attn_lean_kernel + attn_combine_kernel (forward of one encoder block)"),
                                  ("attn_spatial_bwd", 3.5, "attn_spatial_bwd_kernel, both passes + combines (backward of one encoder block: seven products against the forward's two)")):
            n_a, ms_a = model.profile_read(cls_)
            if n_a > 0 and ms_a > 0:
                ach = mult * fl_att / (ms_a / n_a * 1e-3) / 1e12
                roofline_attn[cls_] = {"kernel": label, "bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                                       "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4), "launches": n_a, "avg_launch_ms": round(ms_a / n_a, 4),
                                       "flop_per_launch": mult * fl_att}
        roofline_hbm = hbm_roofline(model, args.encoder, T, (SH, SW), args.clips)
        if state.get("ev"):
            fw = [a.elapsed_time(b) for a, b, _ in state["ev"]]
            bw = [b.elapsed_time(c) for _, b, c in state["ev"]]
            loss_share = {"loss_forward_ms": round(statistics.mean(fw), 3), "loss_backward_plus_hip_backward_ms": round(statistics.mean(bw), 3),
                          "note": "torch.cuda events on the caller's stream in the bracketed steps: loss forward alone; autograd's backward of the loss and "
                                  "edv_backward together"}
    if rank == 0:
        frames_n = world * args.clips * T * args.steps
        loss_name = "L1 stand-in" if args.l1_loss else ("the trainer's loss (trainer_end_to_end_video.py:808-971: masked 0.85 SSIM + 0.15 L1 reprojection against refined, "
                                                        "loss_transform, get_smooth_bright, smoothness" + (", depth reprojection + flow consistency" if args.depth_consistency else "") +
                                                        "; 4 scales; side-network outputs synthetic; gradients to disp, poses, K, inv_K, refined, transform_high; " +
                                                        ("eager PyTorch ops)" if args.torch_loss else "fused HIP kernels, edv_trainer_loss)"))
        line = {
            "metric": f"fine-tune frames/sec ({SH}x{SW}, T={T} clip, LoRA factors trainable)", "value": round(frames_n / dt, 2), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "precision": PRECISION_NOTE["f32"] + " (the fine-tune step: forward, loss and backward)",
            "config": {"workload": f"ViT-{args.encoder[-1].upper()} endodav fine-tune step ({args.lora}{' + temporal_lora' if args.temporal_lora else ''} r=4, "
                                   f"{'conv head, conv_depth_* trainable' if args.conv_head else 'VDA head'}): forward + {loss_name} + HIP backward + gradient "
                                   f"all-reduce + AdamW, {args.clips} synthetic clip(s) of T={T} {x.shape[-2]}x{x.shape[-1]} frames -> image_shape {SH}x{SW} per GPU per step",
                       "encoder": args.encoder, "T": T, "image": [SH, SW], "clips_per_gpu_per_step": args.clips,
                       "parallelism": f"data-parallel x{world}: one in-place all-reduce of the flat gradient buffer ({state['n']} gradient floats) per step"},
            "trainable_floats": state["n"], "loss": float(state["loss"].item()), "device_mem_mb": round(model.device_bytes() / 2 ** 20, 1),
            "gradients_in_flat_buffer": model.flat_gradients(params) is not None,
            "roofline": roofline, "roofline_attention": roofline_attn or None, "roofline_hbm": roofline_hbm, "loss_share": loss_share, "cpu_baseline": None,
        }
        if rehearsal:
            line["rehearsal"] = f"{world} ranks on one GPU: launch-path check, not a scaling number"
        print(json.dumps(line), flush=True)
    if world > 1:
        parallel.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
