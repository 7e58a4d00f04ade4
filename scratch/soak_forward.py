"""Soak: the same clip through the engine a few hundred times (two encoder frame groups, head side stream, split GEMMs / convolutions);
every output must be bit-identical to the first call's.  Then the same for the fine-tune step's gradients."""
import sys, torch
sys.path.insert(0, ".")
import endodav_amd
from endodav_amd import synth
dev = torch.device("cuda:0")
T = int(sys.argv[1]) if len(sys.argv) > 1 else 8
model = endodav_amd.endodav(encoder="vits", features=64, out_channels=[48, 96, 192, 384], image_shape=(518, 518), lora_type="dvlora", disable_conv_head=True).eval()
synth.fill_module_(model)
model = model.to(dev)
x = torch.from_numpy(synth.synth_clip(1, T, 518, 518, seed=1, kind="tissue")).to(dev)
with torch.no_grad():
    ref = [o.clone() for o in model(x).values()]
    bad = 0
    for it in range(300):
        out = model(x)
        if it % 10 == 0:
            bad += sum(not torch.equal(a, b) for a, b in zip(out.values(), ref))
print(f"inference T={T}: 300 forwards, {bad} differing outputs among the 30 x 4 compared")
endodav_amd.mark_only_part_as_trainable(model, warm_up=True)
model.train()
params = [p for p in model.parameters() if p.requires_grad]
def grads():
    model.zero_grad(set_to_none=True)
    sum(o.mean() for o in model(x).values()).backward()
    return [p.grad.clone() for p in params]
g0 = grads()
badg = 0
for it in range(40):
    g = grads()
    badg += sum(not torch.equal(a, b) for a, b in zip(g, g0))
print(f"fine-tune step: 40 repeats, {badg} differing gradient tensors of {40 * len(params)}")
sys.exit(1 if (bad or badg) else 0)
