import sys, ctypes as C, numpy as np, torch
sys.path.insert(0, ".")
import endodav_amd
from endodav_amd import synth, _lib
from oracle import endodav_oracle as orc
from tests.test_backward_gpu import upstream, set_trainable, FACTORS
from tests.helpers import oracle_config
cuda = torch.device("cuda:0")
H, W, T = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
kwargs = dict(encoder="vits", features=64, out_channels=[48, 96, 192, 384], image_shape=(H, W), lora_type="dvlora", disable_conv_head=True)
model = endodav_amd.endodav(**kwargs, pretrained_path=None)
synth.fill_module_(model)
names = set_trainable(model, FACTORS)
x = torch.from_numpy(synth.synth_clip(1, T, H, W, seed=3, kind="tissue"))
model = model.to(cuda).train()
gouts = upstream([(T, 1, h, w) for (h, w) in model.output_shapes()])
# oracle with retained stage grads (fp64)
sd = {k: (v.detach().cpu().clone().double() if v.is_floating_point() else v.cpu()) for k, v in model.state_dict().items()}
for n in names: sd[n].requires_grad_(True)
stages = {}
out = orc.forward(sd, x.double(), oracle_config(kwargs), stages)
for k in ("path1", "path2", "path3", "path4", "tap0", "tap1", "tap2", "tap3", "mm0", "mm1"): stages[k].retain_grad()
loss = sum((out[("disp", s)] * gouts[s].double()).sum() for s in range(4))
loss.backward()
# hip
o = model(x.to(cuda)); l = sum((o[("disp", s)] * gouts[s].to(cuda)).sum() for s in range(4)); l.backward()
lib = _lib.load(); h = C.c_void_p(model._last.handle)
def ws(name, numel):
    n = C.c_size_t(); _lib.check(lib.edv_stage_copy(h, ("ws:" + name).encode(), None, C.byref(n), None))
    t = torch.empty(n.value, device=cuda); _lib.check(lib.edv_stage_copy(h, ("ws:" + name).encode(), t.data_ptr(), C.byref(n), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return t[:numel].cpu().double()
def cmp(name, stage, nchw=True):
    g = stages[stage].grad
    if nchw:
        Fr, Cc, hh, ww = g.shape
        a = ws(name, g.numel()).reshape(Fr, hh, ww, Cc).permute(0, 3, 1, 2)
    else:
        a = ws(name, g.numel()).reshape(g.shape)
    e = (a - g).abs().max().item() / g.abs().max().item()
    print(f"{name:8s} vs d({stage}): {e:.2e}")
cmp("g.p1", "path1"); cmp("g.p2", "path2")
for j in range(4): cmp(f"g.tap{j}", f"tap{j}", nchw=False)
# ---- finer: redo the output head by hand from path1 with retained grads
import torch.nn.functional as F_
p1 = stages["path1"].detach().clone().requires_grad_(True)
s_ = "head.scratch."
o1 = F_.conv2d(p1, sd[s_ + "output_conv1.weight"], sd[s_ + "output_conv1.bias"], padding=1); o1.retain_grad()
up = F_.interpolate(o1, size=(H, W), mode="bilinear", align_corners=True); up.retain_grad()
o2 = F_.relu(F_.conv2d(up, sd[s_ + "output_conv2.0.weight"], sd[s_ + "output_conv2.0.bias"], padding=1)); o2.retain_grad()
d0 = F_.relu(F_.conv2d(o2, sd[s_ + "output_conv2.2.weight"], sd[s_ + "output_conv2.2.bias"])); d0.retain_grad()
outs = [d0]
for k in (1, 2, 3): outs.append(F_.interpolate(outs[-1], scale_factor=0.5, mode="bilinear", align_corners=True))
sum((outs[k] * gouts[k].double()).sum() for k in range(4)).backward()
def cmp2(name, t):
    g = t.grad; Fr, Cc, hh, ww = g.shape
    a = ws(name, g.numel()).reshape(Fr, hh, ww, Cc).permute(0, 3, 1, 2)
    print(f"{name:8s}: {(a - g).abs().max().item() / g.abs().max().item():.2e}")
cmp2("g.d0", d0); cmp2("g.o2", o2); cmp2("g.up", up); cmp2("g.o1", o1); cmp2("g.p1", p1)
# ---- mask agreement of the saved o2 and disp0
a = ws("hd.o2", o2.numel()).reshape(T, H, W, 32).permute(0, 3, 1, 2)
ref = o2.detach()
mm = ((a > 0) != (ref > 0))
print("o2 value err", (a - ref).abs().max().item(), "mask mismatches", int(mm.sum()), "of", ref.numel(), "max |ref| at mismatches", ref[mm].abs().max().item() if mm.any() else 0, "max |hip| at mismatches", a[mm].abs().max().item() if mm.any() else 0)
d0h = o[("disp", 0)].detach().cpu().double()
m0 = ((d0h > 0) != (d0.detach() > 0))
print("disp0 mask mismatches", int(m0.sum()), "of", d0.numel())
g = o2.grad; gh = ws("g.o2", g.numel()).reshape(T, H, W, 32).permute(0, 3, 1, 2)
bad = ((gh - g).abs() > 1e-3 * g.abs().max())
print("g.o2 elements off by >1e-3 of max:", int(bad.sum()), "; of those at a mask mismatch:", int((bad & mm).sum()), "; at disp0 mismatch:", int((bad & m0.expand_as(bad)).sum()))
w1 = sd["head.scratch.output_conv2.2.weight"].reshape(1, 32, 1, 1)
gd0h = ws("g.d0", d0.numel()).reshape(T, 1, H, W)
formula = (d0h > 0) * gd0h * w1 * (a > 0)
print("g.o2 vs formula from HIP tensors:", (gh - formula).abs().max().item(), " oracle grad vs formula:", (g - formula).abs().max().item(), "max|g|", g.abs().max().item())
idx = torch.nonzero(bad)[:5]
for f_, c_, y_, x_ in idx.tolist(): print((f_, c_, y_, x_), "hip", gh[f_, c_, y_, x_].item(), "oracle", g[f_, c_, y_, x_].item(), "formula", formula[f_, c_, y_, x_].item())
