"""Dump parity metrics of the HIP forward vs the reference goldens for every case (diagnostic)."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from tests import helpers as H
from tests.golden.cases import CASES
from oracle import endodav_oracle as orc
cuda = torch.device("cuda:0")
for name in CASES:
    model, kwargs, shape, kind, store = H.build_model(name)
    model = model.to(cuda)
    x = H.case_input(name).to(cuda)
    with torch.no_grad(): out = model(x)
    g = H.load_golden(name)
    line = [name]
    for s in range(4):
        a = out[("disp", s)].cpu().numpy(); a = a if store == "full" else a[..., ::7, ::7]
        ref = g[f"disp{s}"]
        _, da = orc.disp_to_depth(a.astype(np.float64)); _, db = orc.disp_to_depth(ref.astype(np.float64))
        rel = np.abs(da - db) / db
        absd = np.abs(a.astype(np.float64) - ref) / ref.max()
        for fl in (1e-3, 1e-2):
            m = ref >= fl * ref.max()
            line.append("s%d fl%.0e: %.1e" % (s, fl, rel[m].max()))
        line.append("s%d disp-rel %.1e naive %.1e absrel %.1e | viol(rel>1e-3 & abs>1e-6): %d" % (s, absd.max(), rel.max(), rel.mean(), int(((rel > 1e-3) & (absd > 1e-6)).sum())))
    print(" ; ".join(line), flush=True)
    print("   launches", model.launch_count(), "device MB", model.device_bytes() / 1e6, flush=True)
