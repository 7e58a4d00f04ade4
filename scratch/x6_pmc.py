"""A few launches of edv_gemm_x6 (and edv_gemm) per shape, for a rocprofv3 --pmc pass:  rocprofv3 --pmc ... -- python3 scratch/x6_pmc.py"""
import ctypes as C
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from endodav_amd import _lib

lib = _lib.load()
dev = torch.device("cuda:0")
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
nbytes = lib.edv_gemm_workspace()
ws = torch.zeros(nbytes // 4, device=dev)
for M, N, K in [(10960, 1152, 384), (10960, 384, 1536), (43840, 1024, 4096)]:
    g = torch.Generator(device=dev).manual_seed(1)
    A = torch.randn(M, K, device=dev, generator=g)
    W = torch.randn(N, K, device=dev, generator=g) / math.sqrt(K)
    Cm = torch.empty(M, N, device=dev)
    planes = torch.empty(3 * N * K, dtype=torch.bfloat16, device=dev)
    _lib.check(lib.edv_gemm_x6_split(W.data_ptr(), planes.data_ptr(), N, K, st()))
    for _ in range(6):
        _lib.check(lib.edv_gemm_x6(A.data_ptr(), planes.data_ptr(), Cm.data_ptr(), M, N, K, None, 0, None, None, ws.data_ptr(), nbytes, st()))
        _lib.check(lib.edv_gemm(A.data_ptr(), W.data_ptr(), Cm.data_ptr(), M, N, K, None, 0, None, None, ws.data_ptr(), nbytes, st()))
    torch.cuda.synchronize()
