"""Two encoder 'lanes' (LN -> qkv GEMM -> attention -> proj GEMM with residual) of one block on two streams, as the two-frame-group encoder runs them:
every stage buffer of every round must equal the solo run's.  Prints which stage differs first."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from endodav_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0")
F, N, heads = 2, 1370, 6
D = heads * 64; M = F * N
g = torch.Generator(device=dev).manual_seed(3)
Wq = torch.randn(3 * D, D, device=dev, generator=g) * 0.05; bq = torch.randn(3 * D, device=dev, generator=g) * 0.1
Wp = torch.randn(D, D, device=dev, generator=g) * 0.05; bp = torch.randn(D, device=dev, generator=g) * 0.1
lw = torch.randn(D, device=dev, generator=g) * 0.1 + 1; lb = torch.randn(D, device=dev, generator=g) * 0.1
nb = lib.edv_attn_spatial_workspace(F, N, heads); gb = lib.edv_gemm_workspace()
class Lane:
    def __init__(self, seed):
        gg = torch.Generator(device=dev).manual_seed(seed)
        self.x0 = torch.randn(M, D, device=dev, generator=gg)
        self.x = self.x0.clone(); self.xn = torch.empty(M, D, device=dev); self.qkv = torch.empty(M, 3 * D, device=dev); self.att = torch.empty(M, D, device=dev)
        self.ws = torch.zeros(max(nb // 4, 4), device=dev); self.gws = torch.zeros(gb // 4, device=dev); self.s = torch.cuda.Stream()
    def run(self, blocks=3):
        st = self.s.cuda_stream
        with torch.cuda.stream(self.s):
            self.x.copy_(self.x0)
        for _ in range(blocks):
            _lib.check(lib.edv_layernorm(self.x.data_ptr(), lw.data_ptr(), lb.data_ptr(), self.xn.data_ptr(), M, D, 1e-6, None, 0, 0, st))
            _lib.check(lib.edv_gemm(self.xn.data_ptr(), Wq.data_ptr(), self.qkv.data_ptr(), M, 3 * D, D, bq.data_ptr(), 0, None, None, self.gws.data_ptr(), gb, st))
            _lib.check(lib.edv_attn_spatial(self.qkv.data_ptr(), self.att.data_ptr(), F, N, heads, self.ws.data_ptr(), nb, None, st))
            _lib.check(lib.edv_gemm(self.att.data_ptr(), Wp.data_ptr(), self.x.data_ptr(), M, D, D, bp.data_ptr(), 0, None, self.x.data_ptr(), self.gws.data_ptr(), gb, st))
    def snap(self): return [t.clone() for t in (self.xn, self.qkv, self.att, self.x)]
A, B = Lane(1), Lane(2)
torch.cuda.synchronize()
A.run(); torch.cuda.synchronize(); rA = A.snap()
B.run(); torch.cuda.synchronize(); rB = B.snap()
names = ["xn (LN)", "qkv (GEMM)", "att (attention)", "x (proj + residual)"]
bad = [0] * 4
for it in range(200):
    A.run(); B.run()
    torch.cuda.synchronize()
    for ref, lane in ((rA, A), (rB, B)):
        for i, (r, t) in enumerate(zip(ref, lane.snap())):
            bad[i] += not torch.equal(r, t)
print("F per lane", F, "| of 400 lane-runs, buffers differing from the solo run:", dict(zip(names, bad)))

# ---- one block only: where does the attention output differ?
import collections
A.run(1); torch.cuda.synchronize(); r1 = A.snap()
B.run(1); torch.cuda.synchronize(); r1b = B.snap()
seen = 0
for it in range(400):
    A.run(1); B.run(1); torch.cuda.synchronize()
    for ref, lane, nm in ((r1, A, "A"), (r1b, B, "B")):
        cur = lane.snap()
        if not torch.equal(cur[1], ref[1]): print("qkv differs", nm); 
        if not torch.equal(cur[2], ref[2]) and seen < 6:
            seen += 1
            d = (cur[2] - ref[2]).abs()
            idx = torch.nonzero(d > 0)
            rows = idx[:, 0]; cols = idx[:, 1]
            fr = (rows // N).unique().tolist(); tok = rows % N
            print(f"[{nm} it {it}] att differs at {idx.shape[0]} elements: frames {fr}, token range {int(tok.min())}..{int(tok.max())} ({tok.unique().numel()} rows), heads {(cols // 64).unique().tolist()}, "
                  f"max |diff| {float(d.max()):.3e} of scale {float(ref[2].abs().max()):.3e}; qkv equal: {torch.equal(cur[1], ref[1])}")
print("one-block rounds with a differing attention output:", seen)
