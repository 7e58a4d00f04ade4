import torch, math, ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from endodav_amd import _lib
lib=_lib.load(); cuda=torch.device("cuda:0")
def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed); return (torch.rand(*shape, generator=g) * 2 - 1) * scale
st=C.c_void_p(torch.cuda.current_stream().cuda_stream)
for (M,N,K) in [(1370,1536,384),(1370,384,384),(2740,1152,384),(1370,1536,64)]:
    A,W,b=rnd(M,K,seed=1),rnd(N,K,seed=2,scale=1/math.sqrt(K)),rnd(N,seed=3,scale=0.1)
    z=A.double()@W.double().T+b.double()
    Ad,Wd,bd=A.to(cuda),W.to(cuda),b.to(cuda)
    for act in (0,1,2):
        ref=[z, torch.nn.functional.gelu(z), torch.relu(z)][act]
        nbad=[]
        for rep in range(4):
            Cd=torch.full((M,N),float("nan"),device=cuda)
            lib.edv_gemm(Ad.data_ptr(),Wd.data_ptr(),Cd.data_ptr(),M,N,K,bd.data_ptr(),act,None,None,None,0,st)
            torch.cuda.synchronize()
            nbad.append(int(((Cd.double().cpu()-ref).abs()>1e-4).sum()))
        print(f"M={M} N={N} K={K} act={act}: bad elements per launch {nbad}")
