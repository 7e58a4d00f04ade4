"""Soak of the round-3 default path: N forwards of the headline clip with clips in flight (bf16 x 6 products), every output compared bit for bit with the
first; then the same with a second model instance's forwards running beside them on its own lanes (different kernels co-resident on the SIMDs)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import endodav_amd
from endodav_amd import synth
from endodav_amd.pipeline import ClipsInFlight

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
kw = dict(encoder="vits", features=64, out_channels=[48, 96, 192, 384], image_shape=(518, 518), lora_type="dvlora", disable_conv_head=True)
m1 = endodav_amd.endodav(**kw).eval(); synth.fill_module_(m1); m1 = m1.to(dev)
m2 = endodav_amd.endodav(**kw).eval(); synth.fill_module_(m2); m2 = m2.to(dev)
assert m1.products == "bf16x6"
x = torch.from_numpy(synth.synth_clip(1, 8, 518, 518, seed=0)).to(dev)
x2 = torch.from_numpy(synth.synth_clip(1, 5, 518, 518, seed=1)).to(dev)
with torch.no_grad():
    ref = {k: v.clone() for k, v in m1(x).items()}
    ref2 = {k: v.clone() for k, v in m2(x2).items()}
    f1, f2 = ClipsInFlight(m1, dev, depth=3), ClipsInFlight(m2, dev, depth=2)
    bad = 0
    t0 = time.perf_counter()
    hs = []
    for i in range(n):
        hs.append((1, f1.submit(x, resident=True)))
        if i % 2 == 0:
            hs.append((2, f2.submit(x2, resident=True)))
        if len(hs) > 8:
            which, h = hs.pop(0)
            out = h.result()
            r = ref if which == 1 else ref2
            bad += int(any(not torch.equal(out[k], r[k]) for k in r))
    for which, h in hs:
        out = h.result()
        r = ref if which == 1 else ref2
        bad += int(any(not torch.equal(out[k], r[k]) for k in r))
    torch.cuda.synchronize()
    print(f"{n} + {(n + 1) // 2} forwards on two models' lanes in {time.perf_counter() - t0:.2f} s: {bad} outputs differ from the first")
sys.exit(1 if bad else 0)
