import sys, torch
sys.path.insert(0, ".")
from tests.test_backward_gpu import *
from tests.helpers import build_model, case_input
cuda = torch.device("cuda:0")
case = sys.argv[1]
model, kwargs, shape, kind, _ = build_model(case)
x = case_input(case)
names = set_trainable(model, FACTORS)
model = model.to(cuda).train()
BT = shape[0] * shape[1]
gouts = upstream([(BT, 1, h, w) for (h, w) in model.output_shapes()])
ref, _ = oracle_grads(model, kwargs, x, names, gouts, torch.float64)
ref32, _ = oracle_grads(model, kwargs, x, names, gouts)
hip, _ = hip_grads(model, x, names, gouts, cuda)
for n in names:
    s = ref[n].abs().max().item()
    e = (hip[n].cpu().double() - ref[n]).abs().max().item() / s
    e32 = (ref32[n].double() - ref[n]).abs().max().item() / s
    if e > 1e-4 or "ff.net" in n: print(f"{n:90s} hip {e:.2e} ref32 {e32:.2e} scale {s:.2e}")
