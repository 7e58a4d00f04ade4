"""Which lane / output / how much: ClipsInFlight outputs against the lane-0 solo run, (1) every lane alone, (2) all lanes overlapped, repeated."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import endodav_amd
from endodav_amd import synth
from endodav_amd.pipeline import ClipsInFlight
dev = torch.device("cuda:0")
enc = sys.argv[1]; T = int(sys.argv[2]); reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
D = int(sys.argv[4]) if len(sys.argv) > 4 else 4
CFG = {"vits": dict(encoder="vits", features=64, out_channels=[48, 96, 192, 384]), "vitb": dict(encoder="vitb", features=128, out_channels=[96, 192, 384, 768])}[enc]
m = endodav_amd.endodav(**CFG, image_shape=(518, 518), lora_type="dvlora", disable_conv_head=True).eval()
synth.fill_module_(m); m = m.to(dev)
x = torch.from_numpy(synth.synth_clip(1, T, 518, 518, seed=0)).to(dev)
def fwd(lane):
    with torch.no_grad():
        return [o.clone() for o in m(x, lane=lane).values()]
ref = fwd(0); torch.cuda.synchronize()
def cmp(tag, outs):
    for s, (a, b) in enumerate(zip(outs, ref)):
        if not torch.equal(a, b):
            d = (a - b).abs()
            print(f"  {tag}: disp{s} differs: max {d.max().item():.3e} of scale {b.abs().max().item():.3e}, {int((d > 0).sum())} of {d.numel()} pixels, frames {sorted(set((d.flatten(1).max(1).values > 0).nonzero().flatten().tolist()))}", flush=True)
            return 1
    return 0
print(f"== {enc} T={T}: every lane alone (serial)")
for lane in range(4):
    for r in range(3):
        o = fwd(lane); torch.cuda.synchronize()
        cmp(f"lane {lane} run {r}", o)
print(f"== overlapped, depth {D}")
fl = ClipsInFlight(m, dev, depth=D)
bad = 0
for r in range(reps):
    hs = [fl.submit(x, resident=True) for _ in range(2 * D)]
    for i, h in enumerate(hs):
        bad += cmp(f"rep {r} clip {i} (lane {i % D})", list(h.result().values()))
torch.cuda.synchronize()
print(f"overlapped: {bad} of {reps * 2 * D} clips differ")
