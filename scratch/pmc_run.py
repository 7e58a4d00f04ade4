"""A few single launches for a rocprofv3 --pmc pass (diagnostic)."""
import ctypes as C, sys, torch
sys.path.insert(0, ".")
from endodav_amd import _lib
lib = _lib.load()
import os as _os
import torch as _t
GWS = _t.zeros(lib.edv_gemm_workspace() // 4 if not _os.environ.get('KB_NO_WS') else 4, device='cuda:0'); dev = torch.device("cuda:0")
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
def gemm(M, N, K, reps=3):
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * 0.05; Cm = torch.empty(M, N, device=dev); b = torch.randn(N, device=dev)
    for _ in range(reps): _lib.check(lib.edv_gemm(A.data_ptr(), W.data_ptr(), Cm.data_ptr(), M, N, K, b.data_ptr(), 0, None, None, (GWS.data_ptr() if GWS.numel() > 4 else None), (GWS.numel() * 4 if GWS.numel() > 4 else 0), st()))
    torch.cuda.synchronize()
def attn(F, N, heads, reps=3):
    qkv = torch.randn(F * N, 3 * heads * 64, device=dev); o = torch.empty(F * N, heads * 64, device=dev)
    nb = lib.edv_attn_spatial_workspace(F, N, heads)
    ws = torch.empty(max(nb // 4, 4), device=dev)
    for _ in range(reps): _lib.check(lib.edv_attn_spatial(qkv.data_ptr(), o.data_ptr(), F, N, heads, ws.data_ptr(), nb, None, st()))
    torch.cuda.synchronize()
gemm(8192, 8192, 1024); gemm(10960, 1152, 384); gemm(10960, 1536, 4096)
attn(8, 1370, 6); attn(8, 4096, 6)
