import ctypes as C, sys, numpy as np, torch
sys.path.insert(0, ".")
from endodav_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0")
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
def interp_matrix(n_in, n_out):
    Wm = np.zeros((n_out, n_in)); ratio = np.float32(n_in - 1) / np.float32(n_out - 1)
    for o in range(n_out):
        src = np.float32(ratio * np.float32(o)); i0 = int(src); i1 = i0 + (1 if i0 < n_in - 1 else 0)
        lam = float(np.float32(src - np.float32(i0))); Wm[o, i0] += 1.0 - lam; Wm[o, i1] += lam
    return Wm
for n_in, n_out in ((148, 259), (518, 259)):
    g = torch.eye(n_out, device=dev).reshape(n_out, n_out, 1, 1).contiguous()   # F = n_out frames, each one-hot along y
    dx = torch.empty(n_out, n_in, 1, 1, device=dev)
    _lib.check(lib.edv_bilinear_bwd(g.data_ptr(), dx.data_ptr(), n_out, n_in, 1, 1, n_out, 1, 0, st()))
    K = dx.reshape(n_out, n_in).cpu().double().numpy()
    A = interp_matrix(n_in, n_out)
    d = np.abs(K - A)
    print(n_in, n_out, "max diff", d.max())
    idx = np.argwhere(d > 1e-6)[:6]
    for o, i in idx: print("  o", o, "i", i, "kernel", K[o, i], "ref", A[o, i])
