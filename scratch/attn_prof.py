"""rocprofv3 target: the T=8 ViT-S attention launch, 50 times."""
import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from endodav_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0")
F, N, heads = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (8, 1370, 6)
qkv = torch.randn(F * N, 3 * heads * 64, device=dev); o = torch.empty(F * N, heads * 64, device=dev)
nb = lib.edv_attn_spatial_workspace(F, N, heads); ws = torch.empty(max(nb // 4, 4), device=dev)
st = _lib.stream_ptr()
for _ in range(50):
    _lib.check(lib.edv_attn_spatial(qkv.data_ptr(), o.data_ptr(), F, N, heads, ws.data_ptr(), nb, None, st))
torch.cuda.synchronize()
