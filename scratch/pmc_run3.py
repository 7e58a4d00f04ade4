import ctypes as C, sys, torch
sys.path.insert(0, ".")
from endodav_amd import _lib
lib = _lib.load()
import os as _os
import torch as _t
GWS = _t.zeros(lib.edv_gemm_workspace() // 4 if not _os.environ.get('KB_NO_WS') else 4, device='cuda:0'); dev = torch.device("cuda:0")
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
M, N, K = 10960, 1152, 384
A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * 0.05; Cm = torch.empty(M, N, device=dev); b = torch.randn(N, device=dev)
planes = torch.empty(3 * N * K, dtype=torch.bfloat16, device=dev)
for _ in range(2): _lib.check(lib.edv_gemm(A.data_ptr(), W.data_ptr(), Cm.data_ptr(), M, N, K, b.data_ptr(), 0, None, None, (GWS.data_ptr() if GWS.numel() > 4 else None), (GWS.numel() * 4 if GWS.numel() > 4 else 0), st()))
for _ in range(2): _lib.check(lib.edv_gemm_sb(A.data_ptr(), W.data_ptr(), planes.data_ptr(), Cm.data_ptr(), M, N, K, b.data_ptr(), 0, None, None, st()))
torch.cuda.synchronize()
