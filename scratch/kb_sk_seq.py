"""Is the stream-K gain of the isolated loop (same launch repeated back to back) still there when the launch sits between other
kernels, as in the forward?  Event-timed sequences: [fc1, fc2] x 50 with fc2 plain / split, and fc2 alone x 50."""
import ctypes as C
import sys
import torch
sys.path.insert(0, ".")
from endodav_amd import _lib

lib = _lib.load()
dev = torch.device("cuda:0")
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
GWS = torch.zeros(lib.edv_gemm_workspace() // 4, device=dev)
M, D = 8 * 1370, 384
x = torch.randn(M, D, device=dev); w1 = torch.randn(4 * D, D, device=dev) * 0.05; h = torch.empty(M, 4 * D, device=dev)
w2 = torch.randn(D, 4 * D, device=dev) * 0.05; y = torch.empty(M, D, device=dev); b1 = torch.randn(4 * D, device=dev); b2 = torch.randn(D, device=dev)
big = [torch.randn(8192, 1024, device=dev), torch.randn(8192, 1024, device=dev) * 0.05, torch.empty(8192, 8192, device=dev)]


def gemm(A, W, Cm, bias, act, R, ws):
    Mm, K = A.shape
    _lib.check(lib.edv_gemm(A.data_ptr(), W.data_ptr(), Cm.data_ptr(), Mm, W.shape[0], K, _lib.ptr(bias), act, None, _lib.ptr(R),
                            GWS.data_ptr() if ws else None, GWS.numel() * 4 if ws else 0, st()))


def timed(fn, iters=50):
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return a.elapsed_time(e) / iters * 1e3


for _ in range(300):
    gemm(big[0], big[1], big[2], None, 0, None, False)
for rep in range(3):
    for ws in (False, True):
        alone = timed(lambda: gemm(h, w2, y, b2, 0, x, ws))
        pair = timed(lambda: (gemm(x, w1, h, b1, 1, None, False), gemm(h, w2, y, b2, 0, x, ws)))
        fc1 = timed(lambda: gemm(x, w1, h, b1, 1, None, False))
        print(f"fc2 {'split' if ws else 'plain'}: alone {alone:6.1f} us   [fc1, fc2] {pair:6.1f} us   fc1 alone {fc1:6.1f} us   -> fc2 inside the pair {pair - fc1:6.1f} us", flush=True)
