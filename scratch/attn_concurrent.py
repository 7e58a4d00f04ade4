"""Two edv_attn_spatial launches running concurrently on two streams (the two-frame-group encoder): every result must equal the one computed alone."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from endodav_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0")
F, N, heads = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (4, 1370, 6)
D = heads * 64
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def mk(seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    qkv = torch.randn(F * N, 3 * D, device=dev, generator=g) * 1.5
    nb = lib.edv_attn_spatial_workspace(F, N, heads)
    return qkv, torch.empty(F * N, D, device=dev), torch.zeros(max(nb // 4, 4), device=dev), nb
A, B = mk(1), mk(2)
def run(x, stream):
    qkv, o, ws, nb = x
    _lib.check(lib.edv_attn_spatial(qkv.data_ptr(), o.data_ptr(), F, N, heads, ws.data_ptr(), nb, None, stream.cuda_stream))
torch.cuda.synchronize()
run(A, s1); torch.cuda.synchronize(); refA = A[1].clone()
run(B, s2); torch.cuda.synchronize(); refB = B[1].clone()
bad = 0
for it in range(200):
    A[1].fill_(float("nan")); B[1].fill_(float("nan"))
    torch.cuda.synchronize()
    for _ in range(3):
        run(A, s1); run(B, s2)
    torch.cuda.synchronize()
    bad += (not torch.equal(A[1], refA)) + (not torch.equal(B[1], refB))
print(f"F={F} N={N} heads={heads}: 200 rounds of 3 concurrent pairs, {bad} results differ from the solo run; max |diff| A {float((A[1]-refA).abs().max()):.3e}")
