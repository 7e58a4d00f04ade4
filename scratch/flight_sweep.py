"""ClipsInFlight depth sweep: frames/s of `steps` clips at depth 1 (plain model(x) loop) / 2 / 3 / 4, interleaved rounds, for one (encoder, T)."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import endodav_amd
from endodav_amd import synth
from endodav_amd.pipeline import ClipsInFlight
dev = torch.device("cuda:0")
enc = sys.argv[1]; T = int(sys.argv[2]); steps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
IH, IW = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (518, 518)
CFG = {"vits": dict(encoder="vits", features=64, out_channels=[48, 96, 192, 384]), "vitb": dict(encoder="vitb", features=128, out_channels=[96, 192, 384, 768]),
       "vitl": dict(encoder="vitl", features=256, out_channels=[256, 512, 1024, 1024])}[enc]
m = endodav_amd.endodav(**CFG, image_shape=(IH, IW), lora_type="dvlora", disable_conv_head=True).eval()
synth.fill_module_(m); m = m.to(dev)
x = torch.from_numpy(synth.synth_clip(1, T, IH, IW, seed=0)).to(dev)
with torch.no_grad():
    ref = [o.clone() for o in m(x).values()]
DEPTHS = (2, 3, 4, 6)
flights = {d: ClipsInFlight(m, dev, depth=d) for d in DEPTHS}
def run(d, n):
    if d == 1:
        with torch.no_grad():
            for _ in range(n): m(x)
        return
    hs = []
    for _ in range(n):
        hs.append(flights[d].submit(x, resident=True))
        if len(hs) > d: hs.pop(0)
    return hs
for d in (1,) + DEPTHS:
    run(d, 5); torch.cuda.synchronize()
torch.cuda.synchronize()
for rnd in range(3):
    row = []
    for d in (1,) + DEPTHS:
        torch.cuda.synchronize(); t0 = time.perf_counter(); run(d, steps); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        row.append(f"depth {d}: {steps * T / dt:8.1f}")
    print(f"{enc} {IH}x{IW} T={T} round {rnd}:  " + "   ".join(row) + "  frames/s", flush=True)
hs = [flights[4].submit(x, resident=True) for _ in range(8)]
bad = 0
for i, h in enumerate(hs):
    for sidx, (a, b) in enumerate(zip(h.result().values(), ref)):
        if not torch.equal(a, b):
            bad += 1
            d = (a - b).abs()
            print(f"  clip {i} (lane {i % 4}) disp{sidx}: max diff {d.max().item():.3e} of scale {b.abs().max().item():.3e}, {int((d > 0).sum())} of {d.numel()} px, frames {sorted(set((d.flatten(1).max(1).values > 0).nonzero().flatten().tolist()))}")
print(f"{enc} T={T}: auto_depth {ClipsInFlight.auto_depth(m, T)}; outputs differing from the solo run: {bad} of 32")
torch.cuda.synchronize()
