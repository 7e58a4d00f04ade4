"""edv_attn_spatial on one stream while edv_gemm / edv_layernorm run on another (the two-frame-group encoder): every result must equal the solo run."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from endodav_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0")
F, N, heads = 4, 1370, 6
D = heads * 64
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
g = torch.Generator(device=dev).manual_seed(1)
qkv = torch.randn(F * N, 3 * D, device=dev, generator=g) * 1.5
o = torch.empty(F * N, D, device=dev)
nb = lib.edv_attn_spatial_workspace(F, N, heads); ws = torch.zeros(max(nb // 4, 4), device=dev)
M, Ng, K = F * N, 1536, 384
A = torch.randn(M, K, device=dev, generator=g); W = torch.randn(Ng, K, device=dev, generator=g) * 0.05; bias = torch.randn(Ng, device=dev, generator=g)
C = torch.empty(M, Ng, device=dev)
gws_b = lib.edv_gemm_workspace(); gws = torch.zeros(gws_b // 4, device=dev)
def attn(): _lib.check(lib.edv_attn_spatial(qkv.data_ptr(), o.data_ptr(), F, N, heads, ws.data_ptr(), nb, None, s1.cuda_stream))
def gemm(act): _lib.check(lib.edv_gemm(A.data_ptr(), W.data_ptr(), C.data_ptr(), M, Ng, K, bias.data_ptr(), act, None, None, gws.data_ptr(), gws_b, s2.cuda_stream))
torch.cuda.synchronize()
attn(); torch.cuda.synchronize(); ref_o = o.clone()
res = {}
for act in (0, 1):
    gemm(act); torch.cuda.synchronize(); ref_c = C.clone()
    bad_o = bad_c = 0
    for it in range(150):
        o.fill_(float("nan")); C.fill_(float("nan")); torch.cuda.synchronize()
        for _ in range(3):
            attn(); gemm(act); gemm(act)
        torch.cuda.synchronize()
        bad_o += not torch.equal(o, ref_o); bad_c += not torch.equal(C, ref_c)
    print(f"attention || gemm(act={act}): of 150 rounds, attention differs {bad_o}, gemm differs {bad_c}; max |diff| attention {float((o-ref_o).abs().max()):.3e} gemm {float((C-ref_c).abs().max()):.3e}", flush=True)
