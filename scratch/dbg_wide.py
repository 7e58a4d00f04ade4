import torch, math, ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from endodav_amd import _lib
lib=_lib.load(); cuda=torch.device("cuda:0")
def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed); return (torch.rand(*shape, generator=g) * 2 - 1) * scale
M,N,K=1370,1536,384
A,W,b=rnd(M,K,seed=1),rnd(N,K,seed=2,scale=1/math.sqrt(K)),rnd(N,seed=3,scale=0.1)
z=A.double()@W.double().T+b.double()
ref=torch.nn.functional.gelu(z)
Ad,Wd,bd=A.to(cuda),W.to(cuda),b.to(cuda)
st=C.c_void_p(torch.cuda.current_stream().cuda_stream)
for rep in range(3):
    Cd=torch.full((M,N),float("nan"),device=cuda)
    lib.edv_gemm(Ad.data_ptr(),Wd.data_ptr(),Cd.data_ptr(),M,N,K,bd.data_ptr(),1,None,None,None,0,st)
    torch.cuda.synchronize()
    got=Cd.double().cpu()
    err=(got-ref).abs()
    bad=(err>1e-4).nonzero()
    print("rep",rep,"bad elements",len(bad))
    for (r,c) in bad[:12].tolist():
        print(f"  [{r},{c}] tile({r//64},{c//64}) in-tile row {r%64} col {c%64}: got {got[r,c]:.5f} ref {ref[r,c]:.5f} z {z[r,c]:.5f}  got==z? {abs(got[r,c]-z[r,c])<1e-4}  gelu(z) at [r,c+1..3] {[round(float(ref[r,c+k]),4) for k in (1,2,3)]} got there {[round(float(got[r,c+k]),4) for k in (1,2,3)]}")
# where does a wrong value come from?  search the same 64x64 tile of ref (post-GELU) and z (pre-activation, with bias) and z - bias
zb = (A.double()@W.double().T)
for (r,c) in bad[:8].tolist():
    tr, tc = r//64*64, c//64*64
    g = got[r,c]
    for name, T in (("gelu", ref), ("z", z), ("acc", zb)):
        blk = T[tr:tr+64, tc:tc+64]
        hit = ((blk - g).abs() < 2e-5).nonzero()
        if len(hit): print(f"  [{r%64},{c%64}] value {g:.5f} == {name} at in-tile {hit[:3].tolist()}")
print("hazard check: got[r,c] vs 0.70710678 * z[r+8,c]")
for (r,c) in bad[:10].tolist():
    if r + 8 < M: print(f"  [{r},{c}] got {got[r,c]:.6f}   0.7071*z[r+8,c] = {0.70710678*float(z[r+8,c]):.6f}")
