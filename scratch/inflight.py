"""Two clips in flight: consecutive forwards alternate between two engine contexts on two caller streams, so one clip's DPT head
runs beside the next clip's encoder (what a video pipeline does with independent windows).  Compared with the sequential loop."""
import sys, time
import torch
sys.path.insert(0, ".")
import endodav_amd
from endodav_amd import synth

T = int(sys.argv[1]) if len(sys.argv) > 1 else 8
enc = sys.argv[2] if len(sys.argv) > 2 else "vits"
kw = {"vits": dict(encoder="vits", features=64, out_channels=[48, 96, 192, 384]), "vitb": dict(encoder="vitb", features=128, out_channels=[96, 192, 384, 768])}[enc]
dev = torch.device("cuda:0")
models = []
for _ in range(2):
    m = endodav_amd.endodav(**kw, image_shape=(518, 518), lora_type="dvlora", disable_conv_head=True).eval()
    synth.fill_module_(m)
    models.append(m.to(dev))
xs = [torch.from_numpy(synth.synth_clip(1, T, 518, 518, seed=i)).to(dev) for i in range(2)]
streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
K = 40


def run(inflight):
    with torch.no_grad():
        for rep in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(K):
                j = i % 2 if inflight else 0
                with torch.cuda.stream(streams[j]):
                    out = models[j](xs[j])
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
    return K * T / dt, out


for mode in (False, True, False, True):
    fps, out = run(mode)
    print(f"{enc} T={T} {'two clips in flight' if mode else 'sequential        '}: {fps:7.1f} frames/s", flush=True)
a = models[0](xs[0]) if False else None
