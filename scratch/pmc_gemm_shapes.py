"""The four encoder GEMM shapes of ViT-S at T=8, three launches each, for a rocprofv3 --pmc pass (MFMA-busy counters)."""
import ctypes as C, sys, torch
sys.path.insert(0, ".")
from endodav_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0")
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
M = 8 * 1370
for (N, K, act, res) in ((1152, 384, 0, 0), (1536, 384, 1, 0), (384, 384, 0, 1), (384, 1536, 0, 1)):
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * 0.05; Cm = torch.empty(M, N, device=dev); b = torch.randn(N, device=dev)
    R = torch.randn(M, N, device=dev) if res else None
    for _ in range(3):
        _lib.check(lib.edv_gemm(A.data_ptr(), W.data_ptr(), Cm.data_ptr(), M, N, K, b.data_ptr(), act, None, _lib.ptr(R), None, 0, st()))
    torch.cuda.synchronize()
