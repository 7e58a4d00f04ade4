"""Round 3 lead: consecutive clips are independent, so clip k+1's encoder can fill the ramps / tails / small-grid head kernels of clip k.
K model instances (same weights, one engine context and one stream each) take clips round-robin; frames/s against the serial loop, interleaved rounds."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import endodav_amd
from endodav_amd import synth
dev = torch.device("cuda:0")
T = int(sys.argv[1]) if len(sys.argv) > 1 else 8
enc = sys.argv[2] if len(sys.argv) > 2 else "vits"
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
CFG = {"vits": dict(encoder="vits", features=64, out_channels=[48, 96, 192, 384]), "vitb": dict(encoder="vitb", features=128, out_channels=[96, 192, 384, 768])}[enc]
def make():
    m = endodav_amd.endodav(**CFG, image_shape=(518, 518), lora_type="dvlora", disable_conv_head=True).eval()
    synth.fill_module_(m)
    return m.to(dev), torch.cuda.Stream()
models = [make() for _ in range(3)]
x = torch.from_numpy(synth.synth_clip(1, T, 518, 518, seed=0)).to(dev)
def run(k, n):
    with torch.no_grad():
        for i in range(n):
            m, s = models[i % k]
            with torch.cuda.stream(s):
                m(x)
with torch.no_grad():
    ref = [o.clone() for o in models[0][0](x).values()]
for k in (1, 2, 3):
    run(k, 6)
torch.cuda.synchronize()
for rnd in range(3):
    for k in (1, 2, 3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        run(k, steps)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"round {rnd}: {k} clip(s) in flight: {steps * T / dt:8.1f} frames/s  ({dt / steps * 1e3:.3f} ms per clip)", flush=True)
with torch.no_grad():
    outs = []
    for i in range(6):
        m, s = models[i % 3]
        with torch.cuda.stream(s):
            outs.append([o.clone() for o in m(x).values()])
torch.cuda.synchronize()
bad = sum(not torch.equal(a, b) for o in outs for a, b in zip(o, ref))
print(f"outputs differing from the solo run: {bad} of {len(outs) * 4}")
