"""Small grids with deep K (the C = 384 motion module's ff.net.2 and friends): plain grid vs the K split."""
import ctypes as C, sys, torch
sys.path.insert(0, ".")
from endodav_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0")
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
GWS = torch.zeros(lib.edv_gemm_workspace() // 4, device=dev)
def run(A, W, Cm, b, ws):
    M, K = A.shape
    _lib.check(lib.edv_gemm(A.data_ptr(), W.data_ptr(), Cm.data_ptr(), M, W.shape[0], K, b.data_ptr(), 0, None, None, GWS.data_ptr() if ws else None, GWS.numel() * 4 if ws else 0, st()))
def timed(fn, iters=200):
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / iters * 1e3
big = [torch.randn(8192, 1024, device=dev), torch.randn(8192, 1024, device=dev) * 0.05, torch.empty(8192, 8192, device=dev), torch.randn(8192, device=dev)]
timed(lambda: run(big[0], big[1], big[2], big[3], False), 300)
for (M, N, K, what) in ((8 * 361, 384, 1536, "mm1 ff.net.2"), (8 * 361, 384, 768, "K=768"), (8 * 1369, 192, 768, "mm0 ff.net.2"), (8 * 361, 768, 768, ""), (16 * 361, 768, 3072, "ViT-B mm1 ff.net.2 T=16"), (8 * 361, 64, 1536, "")):
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * 0.05; Cm = torch.empty(M, N, device=dev); b = torch.randn(N, device=dev)
    ts = {False: [], True: []}
    for rep in range(3):
        for ws in (False, True): ts[ws].append(timed(lambda: run(A, W, Cm, b, ws)))
    print(f"M={M:6d} N={N:4d} K={K:4d} tiles {(M + 63) // 64 * ((N + 63) // 64):4d}  plain {min(ts[False]):6.1f} us  split {min(ts[True]):6.1f} us   {what}", flush=True)
