"""Backward kernels on MI355X against torch autograd in fp64 on the CPU: each C-ABI entry point
gets the same upstream gradient as the autograd graph of the op the reference calls."""
import ctypes as C
import math

import pytest
import torch
import torch.nn.functional as F

from endodav_amd import _lib

pytestmark = pytest.mark.gpu


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


def st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def close(a, b, rtol, what=""):
    a, b = a.double().cpu(), b.double().cpu()
    err = (a - b).abs().max().item() / max(b.abs().max().item(), 1e-30)
    assert err <= rtol, f"{what}: scale-relative error {err:.3e} > {rtol:.1e}"
    return err


_KEEP = []


def keep(t):
    """Device pointer of a tensor that must outlive the (asynchronous) call it is passed to."""
    _KEEP.append(t)
    if len(_KEEP) > 64:
        torch.cuda.synchronize()
        del _KEEP[:32]
    return t.data_ptr()


def grad_of(fn, inputs, gout, dtype=torch.float64):
    """d(sum(fn(*inputs) * gout)) / d(inputs), in fp64 unless the op's index arithmetic is dtype-dependent."""
    xs = [t.to(dtype).requires_grad_(True) for t in inputs]
    y = fn(*xs)
    return torch.autograd.grad(y, xs, gout.to(dtype))


@pytest.mark.parametrize("rows,dim,acc", [(1000, 384, False), (777, 1024, True), (129, 32, False), (5, 768, True), (300, 192, False)])
def test_layernorm_bwd(lib, cuda, rows, dim, acc):
    x, w, g = rnd(rows, dim, seed=1, scale=3) + 0.5, rnd(dim, seed=2) + 1.0, rnd(rows, dim, seed=3)
    (ref,) = grad_of(lambda xx: F.layer_norm(xx, (dim,), w.double(), None, 1e-6), [x], g)
    base = rnd(rows, dim, seed=4)
    dx = base.to(cuda) if acc else torch.full((rows, dim), float("nan"), device=cuda)
    _lib.check(lib.edv_layernorm_bwd(keep(x.to(cuda)), keep(w.to(cuda)), keep(g.to(cuda)), dx.data_ptr(), rows, dim, 1e-6, int(acc), st()))
    close(dx, ref + (base.double() if acc else 0), 3e-6, "layernorm_bwd")


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_ew_bwd(lib, cuda, mode):
    n = 4 * 12345
    d, src, add = rnd(n, seed=1), rnd(n, seed=2, scale=3), rnd(n, seed=3)
    if mode == 1:
        (ref,) = grad_of(lambda s: F.gelu(s), [src], d)
    elif mode == 2:
        ref = torch.where(src > 0, d, torch.zeros_like(d)).double()
    else:
        ref = d.double()
    ref = ref + add.double()
    out = torch.empty(n, device=cuda)
    _lib.check(lib.edv_ew_bwd(keep(d.to(cuda)), keep(src.to(cuda)), keep(add.to(cuda)), out.data_ptr(), n, mode, st()))
    close(out, ref, 2e-6, f"ew_bwd mode {mode}")


def test_geglu_bwd(lib, cuda):
    M, inner = 333, 256
    x, g = rnd(M, 2 * inner, seed=1, scale=2), rnd(M, inner, seed=2)
    (ref,) = grad_of(lambda xx: xx[:, :inner] * F.gelu(xx[:, inner:]), [x], g)
    dx = torch.empty(M, 2 * inner, device=cuda)
    _lib.check(lib.edv_geglu_bwd(keep(x.to(cuda)), keep(g.to(cuda)), dx.data_ptr(), M, inner, st()))
    close(dx, ref, 2e-6, "geglu_bwd")


def test_transpose_scale_feeds_the_dx_gemm(lib, cuda):
    """dX = (dY * gamma) W through edv_gemm with the transposed, gamma-scaled weight."""
    M, N, K = 500, 96, 160
    W, gamma, dY = rnd(N, K, seed=1), rnd(N, seed=2) + 1.5, rnd(M, N, seed=3)
    ref = (dY.double() * gamma.double()) @ W.double()
    Wt = torch.empty(K, N, device=cuda)
    _lib.check(lib.edv_transpose_scale(keep(W.to(cuda)), keep(gamma.to(cuda)), Wt.data_ptr(), N, K, st()))
    assert torch.equal(Wt.cpu(), (W * gamma[:, None]).T.contiguous())
    dX = torch.empty(M, K, device=cuda)
    _lib.check(lib.edv_gemm(keep(dY.to(cuda)), Wt.data_ptr(), dX.data_ptr(), M, K, N, None, 0, None, None, None, 0, st()))
    close(dX, ref, 3e-6, "dX gemm")


@pytest.mark.parametrize("M,nin,nout,r,dv,use_gamma", [(2740, 384, 1536, 4, True, False), (2740, 1536, 384, 4, True, True), (1000, 64, 256, 8, False, True),
                                                        (70, 256, 64, 2, True, False)])
def test_lora_grads(lib, cuda, M, nin, nout, r, dv, use_gamma):
    x, G = rnd(M, nin, seed=1), rnd(M, nout, seed=2)
    A, Bm, U, V = rnd(r, nin, seed=3), rnd(nout, r, seed=4), rnd(r, 1, seed=5) + 1.2, rnd(nout, 1, seed=6) + 1.2  # DVLinear: U [r,1], V [out,1]
    W, gamma, s = rnd(nout, nin, seed=7, scale=0.05), (rnd(nout, seed=8) + 1.5 if use_gamma else None), 0.75

    def fwd(a, b, u, v):
        weff = W.double() + s * ((b * v) if dv else b) @ ((a * u) if dv else a)
        y = x.double() @ weff.T
        return y * gamma.double() if use_gamma else y

    if dv:
        refs = grad_of(fwd, [A, Bm, U, V], G)
    else:
        refs = grad_of(lambda a, b: fwd(a, b, None, None), [A, Bm], G) + (None, None)
    d = lambda t: None if t is None else t.to(cuda)
    nb = lib.edv_lora_grads_workspace(M, nin, nout, r)
    ws = torch.full((nb // 4,), float("nan"), device=cuda)
    outs = [torch.full_like(t, float("nan")).to(cuda) for t in (A, Bm, U, V)]
    Ud, Vd = (d(U), d(V)) if dv else (None, None)
    _lib.check(lib.edv_lora_grads(keep(d(x)), keep(d(G)), M, nin, nout, r, keep(d(A)), keep(d(Bm)), _lib.ptr(Ud), _lib.ptr(Vd), s,
                                  (keep(d(gamma)) if gamma is not None else None), ws.data_ptr(), nb, outs[0].data_ptr(), outs[1].data_ptr(),
                                  outs[2].data_ptr() if dv else None, outs[3].data_ptr() if dv else None, st()), "edv_lora_grads")
    for name, o, ref in zip("ABUV", outs, refs):
        if name in "UV" and not dv:
            continue
        close(o, ref, 1e-5, f"lora d{name}")


# (8, 1370, 6): 528 tasks on 512 resident slots = one whole round + 16 tasks split along the streamed axis
@pytest.mark.parametrize("Fr,N,heads", [(2, 1370, 6), (1, 64, 2), (3, 10, 1), (1, 129, 12), (2, 321, 6), (9, 200, 3), (8, 1370, 6), (40, 300, 16)])
def test_attn_spatial_bwd(lib, cuda, Fr, N, heads):
    D = heads * 64
    qkv, g = rnd(Fr * N, 3 * D, seed=1, scale=2.0), rnd(Fr * N, D, seed=2)

    def fwd(t):
        t = t.reshape(Fr, N, 3, heads, 64).permute(2, 0, 3, 1, 4)
        q, k, v = t[0] * 64 ** -0.5, t[1], t[2]
        return ((q @ k.transpose(-2, -1)).softmax(-1) @ v).transpose(1, 2).reshape(Fr * N, D)

    (ref,) = grad_of(fwd, [qkv], g)
    qd, gd = qkv.to(cuda), g.to(cuda)
    o = torch.empty(Fr * N, D, device=cuda)
    lse = torch.full((Fr * heads * N,), float("nan"), device=cuda)
    nb = lib.edv_attn_spatial_workspace(Fr, N, heads)
    ws = torch.empty(max(nb // 4, 4), device=cuda)
    _lib.check(lib.edv_attn_spatial(qd.data_ptr(), o.data_ptr(), Fr, N, heads, ws.data_ptr(), nb, lse.data_ptr(), st()), "edv_attn_spatial")
    # the log-sum-exp the backward relies on, in base 2
    t = qkv.double().reshape(Fr, N, 3, heads, 64).permute(2, 0, 3, 1, 4)
    lse_ref = torch.logsumexp((t[0] * 0.125) @ t[1].transpose(-2, -1), -1) / math.log(2.0)  # [F, heads, N]
    assert (lse.cpu().double().reshape(Fr, heads, N) - lse_ref).abs().max().item() < 2e-5
    delta = torch.empty(Fr * heads * N, device=cuda)
    dqkv = torch.full((Fr * N, 3 * D), float("nan"), device=cuda)
    nbb = lib.edv_attn_spatial_bwd_workspace(Fr, N, heads)
    wsb = torch.full((max(nbb // 4, 4),), float("nan"), device=cuda)
    _lib.check(lib.edv_attn_spatial_bwd(qd.data_ptr(), o.data_ptr(), gd.data_ptr(), lse.data_ptr(), delta.data_ptr(), dqkv.data_ptr(), Fr, N, heads,
                                        wsb.data_ptr(), nbb, st()), "edv_attn_spatial_bwd")
    for j, name in enumerate("qkv"):
        close(dqkv[:, j * D:(j + 1) * D], ref[:, j * D:(j + 1) * D], 1e-5, f"attn_spatial_bwd d{name}")


@pytest.mark.parametrize("Fr,ih,iw,Cc,oh,ow,acc", [(2, 37, 37, 64, 74, 74, False), (1, 148, 148, 32, 259, 259, False), (2, 74, 50, 1, 37, 25, True),
                                                   (1, 19, 19, 64, 37, 37, True), (3, 5, 7, 4, 5, 7, False), (1, 1, 1, 4, 3, 3, False),
                                                   (1, 518, 518, 1, 259, 259, False), (2, 128, 160, 32, 224, 280, False), (2, 224, 280, 1, 112, 140, True),
                                                   (2, 56, 70, 1, 28, 35, True), (1, 64, 80, 64, 128, 160, False)])
def test_bilinear_bwd(lib, cuda, Fr, ih, iw, Cc, oh, ow, acc):
    """Adjoint of the forward the kernels actually compute: source coordinates in fp32 as ATen's forward (and CUDA
    backward) derives them.  (ATen's CPU backward uses a double-precision scale, 1e-5 .. 5e-5 away at these sizes.)"""
    import numpy as np

    def interp_matrix(n_in, n_out):
        Wm = np.zeros((n_out, n_in), dtype=np.float64)
        ratio = np.float32(n_in - 1) / np.float32(n_out - 1) if n_out > 1 else np.float32(0)
        for o in range(n_out):
            if n_in == n_out:
                Wm[o, o] = 1.0
                continue
            src = np.float32(ratio * np.float32(o))
            i0 = int(src)
            i1 = i0 + (1 if i0 < n_in - 1 else 0)
            lam = float(np.float32(src - np.float32(i0)))
            Wm[o, i0] += 1.0 - lam
            Wm[o, i1] += lam
        return torch.from_numpy(Wm)

    x, g = rnd(Fr, Cc, ih, iw, seed=1), rnd(Fr, Cc, oh, ow, seed=2)
    Wy, Wx = interp_matrix(ih, oh), interp_matrix(iw, ow)
    fwd = torch.einsum("oi,fcij,pj->fcop", Wy, x.double(), Wx)
    torch_fwd = F.interpolate(x, (oh, ow), mode="bilinear", align_corners=True)
    assert (fwd - torch_fwd.double()).abs().max().item() < 2e-6  # the matrices ARE the fp32 forward
    ref = torch.einsum("oi,fcop,pj->fcij", Wy, g.double(), Wx)
    base = rnd(Fr, ih, iw, Cc, seed=3)
    gd = g.permute(0, 2, 3, 1).contiguous().to(cuda)
    dx = base.to(cuda) if acc else torch.full((Fr, ih, iw, Cc), float("nan"), device=cuda)
    _lib.check(lib.edv_bilinear_bwd(gd.data_ptr(), dx.data_ptr(), Fr, ih, iw, Cc, oh, ow, int(acc), st()))
    close(dx.permute(0, 3, 1, 2), ref + (base.permute(0, 3, 1, 2).double() if acc else 0), 3e-6, "bilinear_bwd")


def test_dot_channels_bwd(lib, cuda):
    npix, Cc = 5000, 32
    o2pre, w, b, g = rnd(npix, Cc, seed=1), rnd(Cc, seed=2), torch.tensor([0.1]), rnd(npix, seed=3)
    (ref,) = grad_of(lambda z: F.relu(F.relu(z) @ w.double() + b.double()), [o2pre], g)
    o2 = F.relu(o2pre)
    disp = F.relu(o2 @ w + b)
    d = torch.empty(npix, Cc, device=cuda)
    _lib.check(lib.edv_dot_channels_bwd(keep(g.to(cuda)), keep(disp.to(cuda)), keep(w.to(cuda)), keep(o2.to(cuda)), d.data_ptr(), None, npix, Cc, 0,
                                        st()))
    close(d, ref, 2e-6, "dot_channels_bwd")


@pytest.mark.parametrize("neg", [False, True], ids=["sigmoid", "inv_sigmoid"])
def test_dot_channels_bwd_sigmoid_head(lib, cuda, neg):
    """HeadDepth's last two layers + the sigmoid of dpt_pyramid.py:103-109: input gradient, and the 1x1 conv's own weight / bias
    gradient through colsum_rows with the kept dL/dz as the row scale."""
    npix, Cc = 6000, 32
    o2pre, w, b, g = rnd(npix, Cc, seed=1), rnd(Cc, seed=2), rnd(1, seed=4, scale=0.1), rnd(npix, seed=3)
    sign = -1.0 if neg else 1.0
    fwd = lambda z, ww, bb: torch.sigmoid(sign * (F.relu(z) @ ww + bb))
    ref_z, ref_w, ref_b = grad_of(fwd, [o2pre, w, b], g)
    o2 = F.relu(o2pre)
    disp = fwd(o2pre, w, b)
    d, gz = torch.empty(npix, Cc, device=cuda), torch.empty(npix, device=cuda)
    o2d = o2.to(cuda)
    _lib.check(lib.edv_dot_channels_bwd(keep(g.to(cuda)), keep(disp.to(cuda)), keep(w.to(cuda)), o2d.data_ptr(), d.data_ptr(), gz.data_ptr(), npix, Cc,
                                        2 if neg else 1, st()))
    close(d, ref_z, 2e-6, "dot_channels_bwd sigmoid")
    nb = lib.edv_colsum_workspace(Cc)
    ws = torch.empty(nb // 4, device=cuda)
    dw, db = torch.full((Cc,), float("nan"), device=cuda), torch.full((1,), float("nan"), device=cuda)
    _lib.check(lib.edv_colsum_rows(o2d.data_ptr(), gz.data_ptr(), npix, Cc, ws.data_ptr(), nb, dw.data_ptr(), 0, st()))
    _lib.check(lib.edv_colsum_rows(gz.data_ptr(), None, npix, 1, ws.data_ptr(), nb, db.data_ptr(), 0, st()))
    close(dw, ref_w, 3e-6, "1x1 weight gradient")
    close(db, ref_b, 3e-6, "1x1 bias gradient")


@pytest.mark.parametrize("M,N,acc", [(100000, 32, False), (777 * 4, 64, True), (1369 * 8, 128, False), (5, 4, False), (8 * 518 * 518, 32, False)])
def test_colsum_rows(lib, cuda, M, N, acc):
    P = rnd(M, N, seed=1) + 0.3
    base = rnd(N, seed=2)
    nb = lib.edv_colsum_workspace(N)
    ws = torch.empty(nb // 4, device=cuda)
    out = base.to(cuda) if acc else torch.full((N,), float("nan"), device=cuda)
    _lib.check(lib.edv_colsum_rows(keep(P.to(cuda)), None, M, N, ws.data_ptr(), nb, out.data_ptr(), int(acc), st()))
    close(out, P.double().sum(0) + (base.double() if acc else 0), 3e-6, "colsum_rows")


@pytest.mark.parametrize("Fr,H,W,Cin,Cout,acc", [(2, 9, 12, 32, 32, False), (1, 37, 37, 64, 32, False), (3, 20, 26, 32, 32, True), (1, 7, 5, 128, 64, False),
                                                 (2, 74, 74, 64, 32, False), (8, 148, 148, 32, 32, False), (1, 16, 16, 256, 128, False),
                                                 (2, 12, 16, 16, 32, False), (1, 6, 8, 48, 20, True), (2, 11, 7, 4, 8, False)])
def test_conv3x3_wgrad(lib, cuda, Fr, H, W, Cin, Cout, acc):
    """dW of a 3x3 / stride 1 / padding 1 convolution (the HeadDepth / output_conv layers) against autograd in fp64."""
    x, wt, g = rnd(Fr, Cin, H, W, seed=1), rnd(Cout, Cin, 3, 3, seed=2, scale=0.1), rnd(Fr, Cout, H, W, seed=3)
    (ref,) = grad_of(lambda ww: F.conv2d(x.double(), ww, padding=1), [wt], g)
    base = rnd(Cout, Cin, 3, 3, seed=4)
    xd, gd = x.permute(0, 2, 3, 1).contiguous().to(cuda), g.permute(0, 2, 3, 1).contiguous().to(cuda)
    nb = lib.edv_conv3x3_wgrad_workspace(Fr, H, W, Cin, Cout)
    assert nb > 0
    ws = torch.full((nb // 4,), float("nan"), device=cuda)
    dw = base.to(cuda) if acc else torch.full((Cout, Cin, 3, 3), float("nan"), device=cuda)
    _lib.check(lib.edv_conv3x3_wgrad(xd.data_ptr(), gd.data_ptr(), dw.data_ptr(), Fr, H, W, Cin, Cout, ws.data_ptr(), nb, int(acc), st()), "edv_conv3x3_wgrad")
    close(dw, ref + (base.double() if acc else 0), 5e-6, "conv3x3_wgrad")


@pytest.mark.parametrize("Fr,P,Cc,acc", [(3, 37 * 37, 192, False), (2, 100, 64, True), (1, 9, 384, False)])
def test_groupnorm_bwd(lib, cuda, Fr, P, Cc, acc):
    x, w, b, g = rnd(Fr, P, Cc, seed=1, scale=2) + 0.3, rnd(Cc, seed=2) + 1, rnd(Cc, seed=3), rnd(Fr, P, Cc, seed=4)
    (ref,) = grad_of(lambda xx: F.group_norm(xx.permute(0, 2, 1), 32, w.double(), b.double(), 1e-6).permute(0, 2, 1), [x], g)
    xd, wd, bd = x.to(cuda), w.to(cuda), b.to(cuda)
    y, stats, sums = torch.empty_like(xd), torch.empty(Fr * 32 * 2, device=cuda), torch.empty(Fr * 32 * 2, device=cuda)
    _lib.check(lib.edv_groupnorm(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), y.data_ptr(), stats.data_ptr(), Fr, P, Cc, 32, 1e-6, None, 0, st()))
    base = rnd(Fr, P, Cc, seed=5)
    dx = base.to(cuda) if acc else torch.full((Fr, P, Cc), float("nan"), device=cuda)
    _lib.check(lib.edv_groupnorm_bwd(xd.data_ptr(), stats.data_ptr(), wd.data_ptr(), keep(g.to(cuda)), sums.data_ptr(), dx.data_ptr(), Fr, P, Cc, 32,
                                     int(acc), st()))
    close(dx, ref + (base.double() if acc else 0), 5e-6, "groupnorm_bwd")


@pytest.mark.parametrize("Bc,T,P,Cc", [(1, 8, 300, 192), (2, 3, 50, 64), (1, 16, 61, 384), (1, 32, 41, 32), (1, 1, 9, 64), (1, 8, 361, 384), (2, 5, 77, 64),
                                       (1, 8, 1500, 64), (1, 16, 361, 768), (1, 16, 300, 128), (1, 32, 60, 1024), (2, 12, 50, 512), (1, 32, 200, 256)])
def test_attn_temporal_bwd(lib, cuda, Bc, T, P, Cc):
    heads, d = 8, Cc // 8
    rows = Bc * T * P
    qkv, g = rnd(rows, 3 * Cc, seed=1), rnd(rows, Cc, seed=2)

    def fwd(t):
        t = t.reshape(Bc, T, P, 3, heads, d).permute(3, 0, 2, 4, 1, 5)  # [3, B, P, heads, T, d]
        att = ((t[0] @ t[1].transpose(-2, -1)) * d ** -0.5).softmax(-1) @ t[2]
        return att.permute(0, 3, 1, 2, 4).reshape(rows, Cc)

    (ref,) = grad_of(fwd, [qkv], g)
    dq = torch.full((rows, 3 * Cc), float("nan"), device=cuda)
    _lib.check(lib.edv_attn_temporal_bwd(keep(qkv.to(cuda)), keep(g.to(cuda)), dq.data_ptr(), Bc, T, P, Cc, heads, st()))
    close(dq, ref, 5e-6, "attn_temporal_bwd")


@pytest.mark.parametrize("Fr,H,W,Cin,Cout", [(2, 37, 37, 64, 64), (1, 20, 28, 48, 32), (1, 5, 6, 32, 4), (2, 224, 280, 32, 32), (2, 128, 160, 64, 32),
                                             (1, 64, 80, 48, 64)])
def test_conv3x3_bwd_data_stride1(lib, cuda, Fr, H, W, Cin, Cout):
    x, w, g = rnd(Fr, Cin, H, W, seed=1), rnd(Cout, Cin, 3, 3, seed=2, scale=0.1), rnd(Fr, Cout, H, W, seed=3)
    (ref,) = grad_of(lambda xx: F.conv2d(xx, w.double(), None, padding=1), [x], g)
    wp = torch.empty(Cin * 9 * Cout, device=cuda)
    _lib.check(lib.edv_pack_conv3x3_bwd(keep(w.to(cuda)), wp.data_ptr(), Cout, Cin, st()))
    gd = g.permute(0, 2, 3, 1).contiguous().to(cuda)
    dx = torch.full((Fr, H, W, Cin), float("nan"), device=cuda)
    _lib.check(lib.edv_conv3x3(gd.data_ptr(), wp.data_ptr(), None, dx.data_ptr(), Fr, H, W, Cout, Cin, 1, 0, 0, None, None, st()))
    close(dx.permute(0, 3, 1, 2), ref, 3e-6, "conv3x3 bwd-data")


@pytest.mark.parametrize("Fr,H,W,Cc", [(2, 19, 19, 64), (1, 16, 20, 32), (1, 4, 5, 8)])
def test_conv3x3_s2_bwd(lib, cuda, Fr, H, W, Cc):
    x, w = rnd(Fr, Cc, H, W, seed=1), rnd(Cc, Cc, 3, 3, seed=2, scale=0.1)
    OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    g = rnd(Fr, Cc, OH, OW, seed=3)
    (ref,) = grad_of(lambda xx: F.conv2d(xx, w.double(), None, stride=2, padding=1), [x], g)
    wp = torch.empty(Cc * 9 * Cc, device=cuda)
    _lib.check(lib.edv_pack_conv3x3(keep(w.to(cuda)), wp.data_ptr(), Cc, Cc, st()))
    gd = g.permute(0, 2, 3, 1).contiguous().to(cuda)
    dx = torch.full((Fr, H, W, Cc), float("nan"), device=cuda)
    _lib.check(lib.edv_conv3x3_s2_bwd(gd.data_ptr(), wp.data_ptr(), dx.data_ptr(), Fr, H, W, Cc, Cc, st()))
    close(dx.permute(0, 3, 1, 2), ref, 3e-6, "conv3x3 s2 bwd-data")
    # the engine's route: zero insertion + the stride-1 input-gradient convolution on the MFMA path
    z = torch.full((Fr, H, W, Cc), float("nan"), device=cuda)
    _lib.check(lib.edv_dilate2(gd.data_ptr(), z.data_ptr(), Fr, H, W, Cc, st()))
    wb = torch.empty(Cc * 9 * Cc, device=cuda)
    _lib.check(lib.edv_pack_conv3x3_bwd(keep(w.to(cuda)), wb.data_ptr(), Cc, Cc, st()))
    dx2 = torch.full((Fr, H, W, Cc), float("nan"), device=cuda)
    _lib.check(lib.edv_conv3x3(z.data_ptr(), wb.data_ptr(), None, dx2.data_ptr(), Fr, H, W, Cc, Cc, 1, 0, 0, None, None, st()))
    close(dx2.permute(0, 3, 1, 2), ref, 3e-6, "conv3x3 s2 bwd-data via dilation")


@pytest.mark.parametrize("Fr,h,w,Cc,s", [(2, 37, 37, 48, 4), (1, 16, 20, 96, 2)])
def test_conv_transpose_bwd_data(lib, cuda, Fr, h, w, Cc, s):
    """ConvTranspose2d(k = s) input gradient = GEMM of the pixel-unshuffled dy with the transposed packed weight."""
    x, wt, b = rnd(Fr, Cc, h, w, seed=1), rnd(Cc, Cc, s, s, seed=2, scale=1 / math.sqrt(Cc)), rnd(Cc, seed=3, scale=0.1)
    g = rnd(Fr, Cc, h * s, w * s, seed=4)
    (ref,) = grad_of(lambda xx: F.conv_transpose2d(xx, wt.double(), b.double(), stride=s), [x], g)
    xd, wd, bd = x.permute(0, 2, 3, 1).contiguous().to(cuda), wt.to(cuda), b.to(cuda)
    wp, bp = torch.empty(s * s * Cc * Cc, device=cuda), torch.empty(s * s * Cc, device=cuda)
    y = torch.empty(Fr, h * s, w * s, Cc, device=cuda)
    _lib.check(lib.edv_conv_transpose(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), wp.data_ptr(), bp.data_ptr(), y.data_ptr(), Fr, h, w, Cc, s, st()))
    gd = g.permute(0, 2, 3, 1).contiguous().to(cuda)
    A = torch.empty(Fr * h * w, s * s * Cc, device=cuda)
    _lib.check(lib.edv_pixel_unshuffle(gd.data_ptr(), A.data_ptr(), Fr, h, w, Cc, s, st()))
    wpt = torch.empty(Cc, s * s * Cc, device=cuda)  # packed weight [s*s*C, C] -> [C, s*s*C]
    _lib.check(lib.edv_transpose_scale(wp.data_ptr(), None, wpt.data_ptr(), s * s * Cc, Cc, st()))
    dx = torch.empty(Fr * h * w, Cc, device=cuda)
    _lib.check(lib.edv_gemm(A.data_ptr(), wpt.data_ptr(), dx.data_ptr(), Fr * h * w, Cc, s * s * Cc, None, 0, None, None, None, 0, st()))
    close(dx.reshape(Fr, h, w, Cc).permute(0, 3, 1, 2), ref, 3e-6, "convT bwd-data")
