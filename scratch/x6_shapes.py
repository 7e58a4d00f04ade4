"""Per-shape timing of edv_gemm (fp32 MFMA) against edv_gemm_x6 (bf16 x 6), both with the stream-K workspace, on the encoder's shapes."""
import ctypes as C
import math
import sys

import torch

from endodav_amd import _lib

lib = _lib.load()
dev = torch.device("cuda:0")
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
nbytes = lib.edv_gemm_workspace()
ws = torch.zeros(nbytes // 4, device=dev)
SHAPES = [(10960, 1152, 384, "S qkv"), (10960, 384, 384, "S proj"), (10960, 1536, 384, "S fc1"), (10960, 384, 1536, "S fc2"),
          (21920, 2304, 768, "B qkv"), (21920, 768, 768, "B proj"), (21920, 3072, 768, "B fc1"), (21920, 768, 3072, "B fc2"),
          (43840, 3072, 1024, "L qkv"), (43840, 1024, 4096, "L fc2")]
if len(sys.argv) > 1:
    SHAPES = SHAPES[:int(sys.argv[1])]


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


tot32 = tot6 = 0
for M, N, K, name in SHAPES:
    g = torch.Generator(device=dev).manual_seed(1)
    A = torch.randn(M, K, device=dev, generator=g)
    W = torch.randn(N, K, device=dev, generator=g) / math.sqrt(K)
    b = torch.randn(N, device=dev, generator=g)
    Cm = torch.empty(M, N, device=dev)
    planes = torch.empty(3 * N * K, dtype=torch.bfloat16, device=dev)
    _lib.check(lib.edv_gemm_x6_split(W.data_ptr(), planes.data_ptr(), N, K, st()))
    act = 1 if "fc1" in name else 0
    t32 = timeit(lambda: lib.edv_gemm(A.data_ptr(), W.data_ptr(), Cm.data_ptr(), M, N, K, b.data_ptr(), act, None, None, ws.data_ptr(), nbytes, st()))
    t6 = timeit(lambda: lib.edv_gemm_x6(A.data_ptr(), planes.data_ptr(), Cm.data_ptr(), M, N, K, b.data_ptr(), act, None, None, ws.data_ptr(), nbytes, st()))
    t6p = timeit(lambda: lib.edv_gemm_x6(A.data_ptr(), planes.data_ptr(), Cm.data_ptr(), M, N, K, b.data_ptr(), act, None, None, None, 0, st()))
    gf = 2.0 * M * N * K * 1e-9
    print(f"{name:7s} {M:6d} x {N:5d} x {K:5d}   f32 {t32:8.1f} us {gf / t32 * 1e3:6.1f} TF/s   x6 stream-K {t6:8.1f} us {gf / t6 * 1e3:6.1f} TF/s   x6 plain {t6p:8.1f} us   speed-up {t32 / min(t6, t6p):.2f}", flush=True)
    if name.startswith("S "):
        tot32 += t32
        tot6 += min(t6, t6p)
print(f"ViT-S block total: f32 {tot32:.1f} us, x6 {tot6:.1f} us")
