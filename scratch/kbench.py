"""Kernel micro-benchmarks through the C ABI (diagnostic): TFLOP/s per shape."""
import ctypes as C
import sys

import torch

sys.path.insert(0, ".")
from endodav_amd import _lib

lib = _lib.load()
import os as _os
import torch as _t
GWS = _t.zeros(lib.edv_gemm_workspace() // 4 if not _os.environ.get('KB_NO_WS') else 4, device='cuda:0')
dev = torch.device("cuda:0")
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


def gemm(M, N, K, act=0, res=False, label=""):
    A = torch.randn(M, K, device=dev)
    W = torch.randn(N, K, device=dev) * 0.05
    Cm = torch.empty(M, N, device=dev)
    b = torch.randn(N, device=dev)
    R = torch.randn(M, N, device=dev) if res else None
    t = timeit(lambda: _lib.check(lib.edv_gemm(A.data_ptr(), W.data_ptr(), Cm.data_ptr(), M, N, K, b.data_ptr(), act, None, _lib.ptr(R), (GWS.data_ptr() if GWS.numel() > 4 else None), (GWS.numel() * 4 if GWS.numel() > 4 else 0), st())))
    print(f"gemm {label:10s} M={M:6d} N={N:5d} K={K:5d} act={act} res={int(res)}: {t*1e6:8.1f} us  {2*M*N*K/t/1e12:6.1f} TF", flush=True)


def gemm_sb(M, N, K, act=0, res=False, label=""):
    A = torch.randn(M, K, device=dev)
    W = torch.randn(N, K, device=dev) * 0.05
    planes = torch.empty(3 * N * K, dtype=torch.bfloat16, device=dev)
    Cm = torch.empty(M, N, device=dev)
    b = torch.randn(N, device=dev)
    R = torch.randn(M, N, device=dev) if res else None
    t = timeit(lambda: _lib.check(lib.edv_gemm_sb(A.data_ptr(), W.data_ptr(), planes.data_ptr(), Cm.data_ptr(), M, N, K, b.data_ptr(), act, None, _lib.ptr(R), st())))
    print(f"gemm_sb {label:10s} M={M:6d} N={N:5d} K={K:5d} act={act} res={int(res)}: {t*1e6:8.1f} us  {2*M*N*K/t/1e12:6.1f} TF-eq (incl. W split)", flush=True)


def conv(F, H, W, Cin, Cout, stride=1, pre=0, post=0, label=""):
    x = torch.randn(F, H, W, Cin, device=dev); w = torch.randn(Cout, 9 * Cin, device=dev) * 0.05; b = torch.randn(Cout, device=dev)
    OH, OW = (H - 1) // stride + 1, (W - 1) // stride + 1
    y = torch.empty(F, OH, OW, Cout, device=dev)
    t = timeit(lambda: _lib.check(lib.edv_conv3x3(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), F, H, W, Cin, Cout, stride, pre, post, None, None, st())))
    print(f"conv {label:12s} F={F} {H}x{W} {Cin}->{Cout} s{stride}: {t*1e6:8.1f} us  {2*F*OH*OW*Cout*9*Cin/t/1e12:6.1f} TF", flush=True)


def attn(F, N, heads):
    qkv = torch.randn(F * N, 3 * heads * 64, device=dev)
    o = torch.empty(F * N, heads * 64, device=dev)
    nb = lib.edv_attn_spatial_workspace(F, N, heads)
    ws = torch.empty(max(nb // 4, 4), device=dev)
    t = timeit(lambda: _lib.check(lib.edv_attn_spatial(qkv.data_ptr(), o.data_ptr(), F, N, heads, ws.data_ptr(), nb, None, st())))
    print(f"attn F={F} N={N} heads={heads}: {t*1e6:8.1f} us  {4*N*N*64*heads*F/t/1e12:6.1f} TF", flush=True)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "sb":
    import os
    print("EDV_SB_TILE", os.environ.get("EDV_SB_TILE"))
    for T in (8, 32):
        M = T * 1370
        gemm_sb(M, 1152, 384, label=f"qkv T{T}")
        gemm_sb(M, 1536, 384, act=1, label=f"fc1 T{T}")
        gemm_sb(M, 384, 384, res=True, label=f"proj T{T}")
        gemm_sb(M, 384, 1536, res=True, label=f"fc2 T{T}")
    gemm_sb(8192, 8192, 1024, label="8k8k1k")
    gemm_sb(4096, 4096, 4096, label="4096^3")
elif __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "conv":
    import os
    print("EDV_CONV_DMA", os.environ.get("EDV_CONV_DMA"))
    conv(8, 518, 518, 32, 32, post=1, label="out_conv2.0")
    conv(8, 296, 296, 64, 32, label="out_conv1")
    conv(8, 296, 296, 64, 64, pre=1, label="rcu @296")
    conv(8, 148, 148, 64, 64, pre=1, label="rcu @148")
    conv(8, 74, 74, 64, 64, pre=1, label="rcu @74")
    conv(8, 37, 37, 64, 64, pre=1, label="rcu @37")
    conv(8, 74, 74, 96, 64, label="layer2_rn")
    conv(8, 37, 37, 192, 64, label="layer3_rn")
    conv(8, 37, 37, 384, 384, stride=2, label="resize3 s2")
elif __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "attn":
    import os
    print("EDV_ATTN_PLAIN", os.environ.get("EDV_ATTN_PLAIN"), "EDV_ATTN_KT", os.environ.get("EDV_ATTN_KT"))
    for F, N, h in ((8, 1370, 6), (32, 1370, 6), (1, 1370, 6), (2, 1370, 6), (4, 1370, 6), (16, 1370, 12), (32, 1370, 16), (8, 4096, 6)):
        attn(F, N, h)
elif __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "sweep":
    import os
    print("EDV_GEMM_TILE", os.environ.get("EDV_GEMM_TILE"), "EDV_ATTN_WAVES", os.environ.get("EDV_ATTN_WAVES"))
    for T in (8, 4):
        M = T * 1370
        gemm(M, 1152, 384, label=f"qkv T{T}")
        gemm(M, 1536, 384, act=1, label=f"fc1 T{T}")
        gemm(M, 384, 384, res=True, label=f"proj T{T}")
        gemm(M, 384, 1536, res=True, label=f"fc2 T{T}")
        attn(T, 1370, 6)
elif __name__ == "__main__" and not (len(sys.argv) > 1 and sys.argv[1] in ("attnbwd", "tattn", "head", "gn", "tattn16")):
    M = 8 * 1370
    gemm(M, 1152, 384, label="qkv")
    gemm(M, 1536, 384, act=1, label="fc1+gelu")
    gemm(M, 1536, 384, act=0, label="fc1 noact")
    gemm(M, 384, 384, res=True, label="proj")
    gemm(M, 384, 1536, res=True, label="fc2")
    gemm(M, 1536, 4096, label="bigK")
    gemm(4096, 4096, 4096, label="4096^3")
    gemm(8192, 8192, 1024, label="8k8k1k")
    gemm(32 * 1370, 1152, 384, label="qkv T=32")
    gemm(32 * 1370, 384, 1536, res=True, label="fc2 T=32")
    attn(8, 1370, 6)
    attn(32, 1370, 6)
    attn(8, 4096, 6)
    attn(2, 1370, 6)


def attn_bwd(F, N, heads):
    D = heads * 64
    qkv = torch.randn(F * N, 3 * D, device=dev); o = torch.empty(F * N, D, device=dev); g = torch.randn(F * N, D, device=dev)
    lse = torch.empty(F * heads * N, device=dev); delta = torch.empty(F * heads * N, device=dev); dq = torch.empty(F * N, 3 * D, device=dev)
    nb = lib.edv_attn_spatial_workspace(F, N, heads); ws = torch.empty(max(nb // 4, 4), device=dev)
    _lib.check(lib.edv_attn_spatial(qkv.data_ptr(), o.data_ptr(), F, N, heads, ws.data_ptr(), nb, lse.data_ptr(), st()))
    nbb = lib.edv_attn_spatial_bwd_workspace(F, N, heads); wsb = torch.empty(max(nbb // 4, 4), device=dev)
    t = timeit(lambda: _lib.check(lib.edv_attn_spatial_bwd(qkv.data_ptr(), o.data_ptr(), g.data_ptr(), lse.data_ptr(), delta.data_ptr(), dq.data_ptr(), F, N, heads, wsb.data_ptr(), nbb, st())))
    print(f"attn_bwd F={F} N={N} heads={heads}: {t*1e6:8.1f} us  {14*N*N*64*heads*F/t/1e12:6.1f} TF (7 products)", flush=True)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "attnbwd":
    import os
    print("EDV_ATTN_BWD_WAVES", os.environ.get("EDV_ATTN_BWD_WAVES"))
    attn_bwd(8, 1370, 6); attn_bwd(32, 1370, 6); attn_bwd(16, 1370, 12); attn_bwd(16, 321, 12)


def tattn(B, T, P, C):
    qkv = torch.randn(B * T * P, 3 * C, device=dev); o = torch.empty(B * T * P, C, device=dev)
    t = timeit(lambda: _lib.check(lib.edv_attn_temporal(qkv.data_ptr(), o.data_ptr(), B, T, P, C, 8, st())))
    print(f"attn_temporal B={B} T={T} P={P} C={C}: {t*1e6:8.1f} us  {(qkv.numel()+o.numel())*4/t/1e12:5.2f} TB/s", flush=True)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "tattn":
    import os
    print("EDV_TATTN_PER_QUERY", os.environ.get("EDV_TATTN_PER_QUERY"))
    tattn(1, 8, 1369, 192); tattn(1, 8, 361, 384); tattn(1, 8, 1369, 64); tattn(1, 8, 5476, 64); tattn(1, 4, 5476, 64)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "head":
    for (M, Cc, tag) in ((10952, 192, "mm0"), (2888, 384, "mm1"), (10952, 64, "mm2"), (43808, 64, "mm3")):
        gemm(M, Cc, Cc, label=f"{tag} proj_in")
        gemm(M, 3 * Cc, Cc, label=f"{tag} qkv")
        gemm(M, Cc, Cc, res=True, label=f"{tag} to_out")
        gemm(M, 8 * Cc, Cc, label=f"{tag} ff1")
        gemm(M, Cc, 4 * Cc, res=True, label=f"{tag} ff2")
    for (N, tag) in ((48, "proj0"), (96, "proj1"), (192, "proj2"), (384, "proj3")):
        gemm(10952, N, 384, label=tag)
    gemm(10952, 768, 48, label="convT4")
    gemm(10952, 384, 96, label="convT2")
    for (M, tag) in ((2888, "oc4"), (10952, "oc3"), (43808, "oc2"), (175232, "oc1")):
        gemm(M, 64, 64, label=tag)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "gn":
    for (F_, P, Cc) in ((8, 1369, 192), (8, 361, 384), (8, 1369, 64), (8, 5476, 64)):
        x = torch.randn(F_, P, Cc, device=dev); w = torch.randn(Cc, device=dev); b = torch.randn(Cc, device=dev)
        y = torch.empty_like(x); stats = torch.empty(F_ * 64, device=dev)
        t = timeit(lambda: _lib.check(lib.edv_groupnorm(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), stats.data_ptr(), F_, P, Cc, 32, 1e-6, None, 0, st())))
        nb = lib.edv_groupnorm_workspace(F_, P, Cc); gws = torch.empty(nb // 4, device=dev)
        t2 = timeit(lambda: _lib.check(lib.edv_groupnorm(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), stats.data_ptr(), F_, P, Cc, 32, 1e-6, gws.data_ptr(), nb, st())))
        print(f"   two-stage statistics: {t2*1e6:8.1f} us")
        print(f"groupnorm F={F_} P={P} C={Cc}: {t*1e6:7.1f} us (stats + apply)  {x.numel()*4*2/t/1e12:5.2f} TB/s", flush=True)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "tattn16":
    import os
    print("EDV_TATTN_PER_QUERY", os.environ.get("EDV_TATTN_PER_QUERY"))
    tattn(1, 16, 1369, 384); tattn(1, 16, 361, 768); tattn(1, 16, 1369, 128); tattn(1, 16, 5476, 128)
    tattn(1, 32, 1369, 1024); tattn(1, 32, 361, 1024); tattn(1, 32, 1369, 256); tattn(1, 32, 5476, 256); tattn(1, 32, 5476, 64)
