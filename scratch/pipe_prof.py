"""PCIe-inclusive step: the serial loop of round 1 against ClipPipeline's variants, three interleaved rounds each."""
import os, sys, time
import torch
sys.path.insert(0, ".")
import endodav_amd
from endodav_amd import synth
from endodav_amd.pipeline import ClipPipeline

dev = torch.device("cuda:0")
m = endodav_amd.endodav(encoder="vits", features=64, out_channels=[48, 96, 192, 384], image_shape=(518, 518), lora_type="dvlora", disable_conv_head=True).eval()
synth.fill_module_(m)
m = m.to(dev)
x = torch.from_numpy(synth.synth_clip(1, 8, 518, 518, seed=0))
xh = x.pin_memory()
xd = x.to(dev)
N = 30
print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES"))

def resident():
    with torch.no_grad():
        for _ in range(N): m(xd)

def serial():
    with torch.no_grad():
        for _ in range(N):
            o = m(xh.to(dev, non_blocking=True))
            for h, v in zip(oh, o.values()): h.copy_(v, non_blocking=True)

with torch.no_grad():
    oh = [torch.empty_like(v, device="cpu").pin_memory() for v in m(xd).values()]
pipes = {cs: ClipPipeline(m, dev, copy_streams=cs) for cs in (2, 1, 0)}
variants = {"resident": resident, "serial pcie": serial}
for cs, pp in pipes.items():
    variants["pipeline cs=%d (pinned clip)" % cs] = (lambda pp=pp: sum(1 for _ in pp.run(xh for _ in range(N))))
    variants["pipeline cs=%d (pageable clip)" % cs] = (lambda pp=pp: sum(1 for _ in pp.run(x for _ in range(N))))
res = {k: [] for k in variants}
for rnd in range(4):
    for k, fn in variants.items():
        torch.cuda.synchronize(); t = time.perf_counter()
        fn()
        torch.cuda.synchronize(); res[k].append((time.perf_counter() - t) / N * 1e3)
for k, v in res.items():
    print("%-34s min %.2f  median %.2f ms/clip   (rounds: %s)" % (k, min(v[1:]), sorted(v[1:])[1], " ".join("%.2f" % a for a in v)))
