"""infer_video_depth end to end: a synthetic uint8 video on the host -> depth maps on the host; frames/s in both products modes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import endodav_amd
from endodav_amd import synth

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 232
model = endodav_amd.endodav(encoder="vits", features=64, out_channels=[48, 96, 192, 384], image_shape=(224, 280), lora_type="dvlora", disable_conv_head=True).eval()
synth.fill_module_(model)
model = model.to(dev)
rng = np.random.default_rng(0)
frames = rng.integers(0, 256, size=(n, 256, 320, 3), dtype=np.uint8)
for mode in ("f32", "bf16x6", "f32", "bf16x6"):
    model.products = mode
    out = model.infer_video_depth(frames, device="cuda")  # warm (contexts, planes, workspaces)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = model.infer_video_depth(frames, device="cuda")
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{mode:7s} {n} frames 256x320 -> 224x280 windows of 32: {dt * 1e3:8.1f} ms  {n / dt:8.1f} frames/s  out {np.asarray(out).shape} mean {float(np.asarray(out).mean()):.6f}", flush=True)
