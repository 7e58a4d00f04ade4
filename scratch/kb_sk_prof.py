"""fc2 at T=8 in isolation, plain grid vs stream-K split, for a rocprofv3 --kernel-trace run: are the kernel durations what the
event-timed loop (scratch/kb_sk_ab.py) reports?"""
import ctypes as C
import sys
import torch
sys.path.insert(0, ".")
from endodav_amd import _lib

lib = _lib.load()
dev = torch.device("cuda:0")
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
GWS = torch.zeros(lib.edv_gemm_workspace() // 4, device=dev)
M, N, K = 8 * 1370, 384, 1536
A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * 0.05; Cm = torch.empty(M, N, device=dev)
b = torch.randn(N, device=dev); R = torch.randn(M, N, device=dev)
big = [torch.randn(8192, 1024, device=dev), torch.randn(8192, 1024, device=dev) * 0.05, torch.empty(8192, 8192, device=dev)]
for _ in range(300):
    _lib.check(lib.edv_gemm(big[0].data_ptr(), big[1].data_ptr(), big[2].data_ptr(), 8192, 8192, 1024, None, 0, None, None, None, 0, st()))
for rep in range(3):
    for ws in (False, True):
        for _ in range(50):
            _lib.check(lib.edv_gemm(A.data_ptr(), W.data_ptr(), Cm.data_ptr(), M, N, K, b.data_ptr(), 0, None, R.data_ptr(),
                                    GWS.data_ptr() if ws else None, GWS.numel() * 4 if ws else 0, st()))
torch.cuda.synchronize()
