"""Why a loop keeps fewer clips in flight than asked: the same depth-3 submission loop with (a) handles dropped as in bench.py, (b) all handles kept,
(c) no cross-stream 'ready' event (resident), against the plain one-at-a-time loop.  frames/s, interleaved rounds."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import endodav_amd
from endodav_amd import synth
from endodav_amd.pipeline import ClipsInFlight
dev = torch.device("cuda:0")
m = endodav_amd.endodav(encoder="vits", features=64, out_channels=[48, 96, 192, 384], image_shape=(518, 518), lora_type="dvlora", disable_conv_head=True).eval()
synth.fill_module_(m); m = m.to(dev)
x = torch.from_numpy(synth.synth_clip(1, 8, 518, 518, seed=0)).to(dev)
fl = ClipsInFlight(m, dev, depth=3)
n = 30
def serial():
    with torch.no_grad():
        for _ in range(n): m(x)
def drop(res):
    hs = []
    for _ in range(n):
        hs.append(fl.submit(x, resident=res))
        if len(hs) > 3: hs.pop(0)
def keep(res):
    hs = [fl.submit(x, resident=res) for _ in range(n)]
    return hs
variants = {"serial": serial, "drop+ready": lambda: drop(False), "keep+ready": lambda: keep(False), "drop+resident": lambda: drop(True), "keep+resident": lambda: keep(True)}
for f in variants.values(): f()
torch.cuda.synchronize()
for rnd in range(3):
    row = []
    for name, f in variants.items():
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        row.append(f"{name} {n * 8 / dt:7.1f}")
        del r
    print("   ".join(row), flush=True)
