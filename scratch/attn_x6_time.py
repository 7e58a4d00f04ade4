"""edv_attn_spatial (fp32 MFMA) against edv_attn_spatial_x6 on the encoder's attention shapes."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from endodav_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0")
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for Fr, N, heads, name in [(8, 1370, 6, "ViT-S T=8"), (16, 1370, 12, "ViT-B T=16"), (32, 1370, 16, "ViT-L T=32")]:
    D = heads * 64
    g = torch.Generator(device=dev).manual_seed(1)
    qkv = torch.randn(Fr * N, 3 * D, device=dev, generator=g)
    o = torch.empty(Fr * N, D, device=dev)
    nb = lib.edv_attn_spatial_workspace(Fr, N, heads); ws = torch.zeros(max(nb // 4, 4), device=dev)
    nb6 = lib.edv_attn_spatial_x6_workspace(Fr, N, heads); ws6 = torch.zeros(max(nb6 // 4, 4), device=dev)
    t32 = timeit(lambda: lib.edv_attn_spatial(qkv.data_ptr(), o.data_ptr(), Fr, N, heads, ws.data_ptr(), nb, None, st()))
    t6 = timeit(lambda: lib.edv_attn_spatial_x6(qkv.data_ptr(), o.data_ptr(), Fr, N, heads, ws6.data_ptr(), nb6, st()))
    fl = 4.0 * N * N * 64 * heads * Fr
    print(f"{name:11s} fp32 {t32:8.1f} us {fl / t32 * 1e-6:6.1f} TF/s    bf16x6 {t6:8.1f} us {fl / t6 * 1e-6:6.1f} TF-eq/s   speed-up {t32 / t6:.2f}", flush=True)
