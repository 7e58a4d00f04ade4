"""Warm A/B of the plain grid vs the persistent stream-K split of gemm_dma (workspace argument), interleaved, after 0.5 s of warm-up
work: the clocks of an idle MI355X take ~100 ms of load to settle, which biased the first entry of earlier kbench lists."""
import ctypes as C
import sys
import torch
sys.path.insert(0, ".")
from endodav_amd import _lib

lib = _lib.load()
dev = torch.device("cuda:0")
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
GWS = torch.zeros(lib.edv_gemm_workspace() // 4, device=dev)


def run(A, W, Cm, b, R, act, ws):
    M, K = A.shape
    N = W.shape[0]
    _lib.check(lib.edv_gemm(A.data_ptr(), W.data_ptr(), Cm.data_ptr(), M, N, K, b.data_ptr(), act, None, _lib.ptr(R),
                            GWS.data_ptr() if ws else None, GWS.numel() * 4 if ws else 0, st()))


def timed(fn, iters):
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return a.elapsed_time(e) / iters * 1e3


big = [torch.randn(8192, 1024, device=dev), torch.randn(8192, 1024, device=dev) * 0.05, torch.empty(8192, 8192, device=dev), torch.randn(8192, device=dev)]
timed(lambda: run(big[0], big[1], big[2], big[3], None, 0, False), 400)  # ~0.5 s
shapes = [("qkv", 1152, 384, 0, False), ("proj", 384, 384, 0, True), ("fc1", 1536, 384, 1, False), ("fc2", 384, 1536, 0, True)]
import os
cases = [(T * 1370, n, N, K, a, r) for T in (8, 4, 16) for (n, N, K, a, r) in shapes]
cases += [(16 * 1370, "B qkv", 2304, 768, 0, False), (16 * 1370, "B proj", 768, 768, 0, True), (16 * 1370, "B fc1", 3072, 768, 1, False), (16 * 1370, "B fc2", 768, 3072, 0, True),
          (8 * 1370, "B fc2/2", 768, 3072, 0, True), (32 * 1370, "L fc1", 4096, 1024, 1, False), (32 * 1370, "L fc2", 1024, 4096, 0, True), (16 * 1370, "L fc2/2", 1024, 4096, 0, True)]
for M, name, N, K, act, res in cases:
    T = M // 1370
    if True:
        A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * 0.05; Cm = torch.empty(M, N, device=dev)
        b = torch.randn(N, device=dev); R = torch.randn(M, N, device=dev) if res else None
        ts = {False: [], True: []}
        for rep in range(3):
            for ws in (False, True):
                ts[ws].append(timed(lambda: run(A, W, Cm, b, R, act, ws), 100))
        fl = 2.0 * M * N * K
        print(f"T={T:2d} {name:5s} tiles {(M + 63) // 64 * ((N + 63) // 64):5d}  plain {min(ts[False]):7.1f} us ({fl / min(ts[False]) / 1e6:6.1f} TF)   "
              f"stream-K {min(ts[True]):7.1f} us ({fl / min(ts[True]) / 1e6:6.1f} TF)   all: {[round(v, 1) for v in ts[False]]} {[round(v, 1) for v in ts[True]]}", flush=True)
