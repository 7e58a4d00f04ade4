import sys, numpy as np, torch
sys.path.insert(0, ".")
import endodav_amd
from endodav_amd import synth
from tests.test_backward_gpu import upstream, set_trainable, FACTORS, oracle_grads, hip_grads
cuda = torch.device("cuda:0")
H, W, T = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
kwargs = dict(encoder="vits", features=64, out_channels=[48, 96, 192, 384], image_shape=(H, W), lora_type="dvlora", disable_conv_head=True)
model = endodav_amd.endodav(**kwargs, pretrained_path=None)
synth.fill_module_(model)
names = set_trainable(model, FACTORS)
x = torch.from_numpy(synth.synth_clip(1, T, H, W, seed=3, kind="tissue"))
model = model.to(cuda).train()
shapes = [(T, 1, h, w) for (h, w) in model.output_shapes()]
gouts = [1.0 + 0.5 * g for g in upstream(shapes)] if len(sys.argv) > 4 else upstream(shapes)
ref32, _ = oracle_grads(model, kwargs, x, names, gouts)
ref64, _ = oracle_grads(model, kwargs, x, names, gouts, torch.float64)
hip, _ = hip_grads(model, x, names, gouts, cuda)
for n in names:
    if not (n.endswith("lora_B") and ("blocks.0." in n or "blocks.5." in n or "blocks.11." in n)): continue
    s = ref64[n].abs().max().item()
    eh = (hip[n].cpu().double() - ref64[n]).abs().max().item() / s
    er = (ref32[n].double() - ref64[n]).abs().max().item() / s
    print(f"{n:45s} hip {eh:.2e}  ref32 {er:.2e}")
