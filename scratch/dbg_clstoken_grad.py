"""Per-tensor gradient errors of micro_clstoken_nocls (conv head + use_clstoken + no cls token) against the fp32 and fp64 oracle graphs."""
import sys
sys.path.insert(0, ".")
import torch
from tests.test_backward_gpu import oracle_grads, hip_grads, upstream
from tests.helpers import build_model, case_input

case = sys.argv[1] if len(sys.argv) > 1 else "micro_clstoken_nocls"
cuda = torch.device("cuda:0")
model, kwargs, shape, kind, _ = build_model(case)
x = case_input(case)
names = []
for n, p in model.named_parameters():
    p.requires_grad = (".mlp.fc" in n and n.rsplit(".", 1)[-1] in ("lora_A", "lora_B")) or n.startswith("head.conv_depth_")
    if p.requires_grad:
        names.append(n)
model = model.to(cuda).train()
BT = shape[0] * shape[1]
gouts = upstream([(BT, 1, h, w) for (h, w) in model.output_shapes()])
ref32, _ = oracle_grads(model, kwargs, x, names, gouts)
ref64, _ = oracle_grads(model, kwargs, x, names, gouts, torch.float64)
hip, _ = hip_grads(model, x, names, gouts, cuda)
rows = []
for n in names:
    r64 = ref64[n].double()
    sc = max(r64.abs().max().item(), 1e-30)
    rows.append((n, (hip[n].cpu().double() - r64).abs().max().item() / sc, (ref32[n].double() - r64).abs().max().item() / sc))
rows.sort(key=lambda r: -r[1])
print("tensor, HIP vs fp64, fp32 oracle vs fp64")
for r in rows[:12]:
    print(f"{r[0]:60s} {r[1]:.2e} {r[2]:.2e}")
print("median HIP err", sorted(r[1] for r in rows)[len(rows) // 2], "median fp32-oracle err", sorted(r[2] for r in rows)[len(rows) // 2])
