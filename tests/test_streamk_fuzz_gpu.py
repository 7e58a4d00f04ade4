"""Seeded shape sweep of the stream-K split (gemm_dma.hip, conv_dma.hip): every launch with a workspace must equal the plain launch
up to fp32 summation order, leave the arrival counters at zero, and be reproducible -- whatever the grid / run / piece geometry."""
import ctypes as C
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from endodav_amd import _lib

pytestmark = pytest.mark.gpu
COUNTER_FLOATS = 4096  # MAX_COUNTERS of gemm_dma.hip


def st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def shapes_gemm(n=36, seed=7):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        K = int(rng.choice([768, 1024, 1536, 2048, 3072, 4096]))
        N = int(rng.choice([64, 192, 384, 768, 1024]))
        tiles_target = int(rng.choice([20, 50, 200, 600, 770, 1030, 1600, 2300, 3100]))
        M = max(1, tiles_target * 64 // max(1, (N + 63) // 64)) + int(rng.integers(-40, 41))
        out.append((max(M, 5), N, K, bool(rng.integers(0, 2)), int(rng.integers(0, 3))))
    return out


@pytest.fixture(scope="module")
def ws(lib, cuda):
    nbytes = lib.edv_gemm_workspace()
    w = torch.full((nbytes // 4,), float("nan"), device=cuda)
    w[:COUNTER_FLOATS] = 0
    return w, nbytes


@pytest.mark.parametrize("M,N,K,res,act", shapes_gemm())
def test_gemm_split_equals_plain(lib, cuda, ws, M, N, K, res, act):
    w, nbytes = ws
    g = torch.Generator().manual_seed(M * 31 + N * 7 + K)
    A = torch.randn(M, K, generator=g).to(cuda)
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(cuda)
    b = torch.randn(N, generator=g).to(cuda)
    R = torch.randn(M, N, generator=g).to(cuda) if res else None
    outs = []
    for use_ws in (False, True, True):
        Cd = torch.full((M, N), float("nan"), device=cuda)
        _lib.check(lib.edv_gemm(A.data_ptr(), W.data_ptr(), Cd.data_ptr(), M, N, K, b.data_ptr(), act, None, _lib.ptr(R),
                                w.data_ptr() if use_ws else None, nbytes if use_ws else 0, st()), "edv_gemm")
        torch.cuda.synchronize()
        outs.append(Cd)
    assert int(w[:COUNTER_FLOATS].view(torch.int32).abs().sum()) == 0
    assert torch.equal(outs[1], outs[2]), "the split launch is not reproducible"
    scale = outs[0].abs().max().item()
    assert (outs[0] - outs[1]).abs().max().item() <= 3e-6 * scale, (M, N, K)


@pytest.mark.parametrize("Fr,H,W,Cin,Cout,stride", [(8, 19, 19, 384, 64, 1), (3, 19, 19, 384, 384, 1), (8, 37, 37, 384, 384, 2), (5, 37, 37, 192, 64, 1),
                                                  (2, 24, 31, 256, 128, 1), (16, 19, 19, 768, 128, 1), (1, 33, 47, 192, 32, 1), (4, 19, 19, 64, 64, 1),
                                                  (7, 21, 13, 512, 64, 2), (2, 37, 37, 1024, 256, 1)])
def test_conv_split_equals_plain(lib, cuda, ws, Fr, H, W, Cin, Cout, stride):
    w, nbytes = ws
    g = torch.Generator().manual_seed(Fr * 131 + H * 17 + Cin)
    x = torch.randn(Fr, H, W, Cin, generator=g).to(cuda)
    wt = (torch.randn(Cout, 9 * Cin, generator=g) / math.sqrt(9 * Cin)).to(cuda)
    b = torch.randn(Cout, generator=g).to(cuda)
    OH, OW = (H - 1) // stride + 1, (W - 1) // stride + 1
    R = torch.randn(Fr, OH, OW, Cout, generator=g).to(cuda)
    outs = []
    for use_ws in (False, True, True):
        y = torch.full((Fr, OH, OW, Cout), float("nan"), device=cuda)
        if use_ws:
            _lib.check(lib.edv_conv3x3_ws(x.data_ptr(), wt.data_ptr(), b.data_ptr(), y.data_ptr(), Fr, H, W, Cin, Cout, stride, 1, 0, R.data_ptr(), None,
                                          w.data_ptr(), nbytes, st()), "edv_conv3x3_ws")
        else:
            _lib.check(lib.edv_conv3x3(x.data_ptr(), wt.data_ptr(), b.data_ptr(), y.data_ptr(), Fr, H, W, Cin, Cout, stride, 1, 0, R.data_ptr(), None, st()),
                       "edv_conv3x3")
        torch.cuda.synchronize()
        outs.append(y)
    assert int(w[:COUNTER_FLOATS].view(torch.int32).abs().sum()) == 0
    assert torch.equal(outs[1], outs[2])
    scale = outs[0].abs().max().item()
    assert (outs[0] - outs[1]).abs().max().item() <= 3e-6 * scale


# ---- spatial attention, forward and backward: the task / run / piece geometry depends on (frames, tokens, heads) --------------------
def shapes_attn(n=24, seed=11):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        heads = int(rng.choice([1, 2, 3, 6, 12, 16]))
        N = int(rng.choice([33, 64, 65, 127, 200, 257, 321, 400, 577, 700]))
        want_tasks = int(rng.choice([3, 40, 130, 260, 500, 520, 530, 800, 1100]))
        Fr = max(1, min(48, want_tasks // (heads * ((N + 127) // 128))))
        out.append((Fr, N, heads))
    return sorted(set(out))


@pytest.mark.parametrize("Fr,N,heads", shapes_attn())
def test_attention_forward_backward_sweep(lib, cuda, Fr, N, heads):
    D = heads * 64
    g = torch.Generator().manual_seed(Fr * 1000 + N * 10 + heads)
    qkv = torch.randn(Fr * N, 3 * D, generator=g) * 1.5
    gout = torch.randn(Fr * N, D, generator=g)
    x = qkv.double().requires_grad_(True)
    t = x.reshape(Fr, N, 3, heads, 64).permute(2, 0, 3, 1, 4)
    ref = ((t[0] * 0.125) @ t[1].transpose(-2, -1)).softmax(-1) @ t[2]
    ref = ref.transpose(1, 2).reshape(Fr * N, D)
    (dref,) = torch.autograd.grad(ref, x, gout.double())
    qd, gd = qkv.to(cuda), gout.to(cuda)
    o = torch.full((Fr * N, D), float("nan"), device=cuda)
    lse = torch.full((Fr * heads * N,), float("nan"), device=cuda)
    nb = lib.edv_attn_spatial_workspace(Fr, N, heads)
    wsf = torch.full((max(nb // 4, 4),), float("nan"), device=cuda)
    _lib.check(lib.edv_attn_spatial(qd.data_ptr(), o.data_ptr(), Fr, N, heads, wsf.data_ptr(), nb, lse.data_ptr(), st()), "edv_attn_spatial")
    ref_d = ref.detach()
    assert (o.cpu().double() - ref_d).abs().max().item() <= 5e-6 * ref_d.abs().max().item()
    delta = torch.empty(Fr * heads * N, device=cuda)
    dqkv = torch.full((Fr * N, 3 * D), float("nan"), device=cuda)
    nbb = lib.edv_attn_spatial_bwd_workspace(Fr, N, heads)
    wsb = torch.full((max(nbb // 4, 4),), float("nan"), device=cuda)
    _lib.check(lib.edv_attn_spatial_bwd(qd.data_ptr(), o.data_ptr(), gd.data_ptr(), lse.data_ptr(), delta.data_ptr(), dqkv.data_ptr(), Fr, N, heads,
                                        wsb.data_ptr(), nbb, st()), "edv_attn_spatial_bwd")
    got = dqkv.cpu().double()
    for j in range(3):
        a, b = got[:, j * D:(j + 1) * D], dref[:, j * D:(j + 1) * D]
        assert (a - b).abs().max().item() <= 1.5e-5 * b.abs().max().item(), "qkv"[j]


# ---- temporal attention (pixel-per-workgroup kernels with head groups): clip length x pixels x width ----------------------------------
def shapes_temporal(n=20, seed=13):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        T = int(rng.choice([1, 2, 3, 5, 8, 9, 12, 16, 17, 24, 32]))
        Cc = int(rng.choice([32, 64, 128, 192, 256, 384, 768, 1024]))
        P = int(rng.choice([1, 7, 50, 361, 500]))
        Bc = int(rng.choice([1, 2]))
        if Bc * T * P * Cc <= 12_000_000:
            out.append((Bc, T, P, Cc))
    return sorted(set(out))


@pytest.mark.parametrize("Bc,T,P,Cc", shapes_temporal())
def test_temporal_attention_forward_backward_sweep(lib, cuda, Bc, T, P, Cc):
    heads, d = 8, Cc // 8
    g = torch.Generator().manual_seed(T * 100 + P + Cc)
    qkv = torch.randn(Bc * T * P, 3 * Cc, generator=g) * 1.2
    gout = torch.randn(Bc * T * P, Cc, generator=g)
    x = qkv.double().requires_grad_(True)
    t = x.reshape(Bc, T, P, 3, heads, d).permute(3, 0, 2, 4, 1, 5)  # [3, B, P, h, T, d]
    a = ((t[0] @ t[1].transpose(-1, -2)) * d ** -0.5).softmax(-1)
    ref = (a @ t[2]).permute(0, 3, 1, 2, 4).reshape(Bc * T * P, Cc)
    (dref,) = torch.autograd.grad(ref, x, gout.double())
    qd, gd = qkv.to(cuda), gout.to(cuda)
    o = torch.full((Bc * T * P, Cc), float("nan"), device=cuda)
    _lib.check(lib.edv_attn_temporal(qd.data_ptr(), o.data_ptr(), Bc, T, P, Cc, heads, st()), "edv_attn_temporal")
    ref_d = ref.detach()
    assert (o.cpu().double() - ref_d).abs().max().item() <= 4e-6 * max(ref_d.abs().max().item(), 1e-30)
    dqkv = torch.full((Bc * T * P, 3 * Cc), float("nan"), device=cuda)
    _lib.check(lib.edv_attn_temporal_bwd(qd.data_ptr(), gd.data_ptr(), dqkv.data_ptr(), Bc, T, P, Cc, heads, st()), "edv_attn_temporal_bwd")
    assert (dqkv.cpu().double() - dref).abs().max().item() <= 1e-5 * dref.abs().max().item()


def test_piece_exchange_contract_under_uneven_load(lib, cuda, ws):
    """The contract the in-kernel piece exchange of the split GEMM rests on (gemm_dma.hip, "piece hand-off"): pieces are written with
    agent-scope (sc1, write-through) stores, each storing wave drains them (s_waitcnt vmcnt(0)) before the workgroup barrier behind which
    ONE lane bumps the tile's agent-scope counter, and the last arriver reads the pieces with agent-scope (L2-served, L1-bypassing) loads
    behind a workgroup barrier that follows its own counter add.  That is a row of the guide's measured hand-off table, not a C++ memory-model
    guarantee, so this test is the gate to re-run after any ROCm / hipcc change.  It stresses what hides a broken hand-off: UNEVEN load (a
    second stream keeps part of the chip busy with launches of other sizes, so pieces of a tile arrive far apart and their producers sit on
    different XCDs), a consumer that has the piece slots warm in its caches (the workspace is poisoned with NaN between launches through
    the same CUs, and every launch re-uses the same slots), and a check of every output word of 200 launches against the first.
    If it ever fails: switch the hand-off to the fenced form (lane 0: __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent") + s_waitcnt vmcnt(0)
    before the counter add; last arriver: __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent") + s_waitcnt vmcnt(0) + barrier before plain loads),
    or set EDV_GEMM_STREAMK=0 (one workgroup per tile, no exchange) until it is."""
    w, nbytes = ws
    g = torch.Generator().manual_seed(11)
    M, N, K = 10960, 384, 1536  # fc2 of ViT-S at T=8: 1032 tiles on 768 persistent workgroups, 264 leftover tiles split along K
    A = torch.randn(M, K, generator=g).to(cuda)
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(cuda)
    b = torch.randn(N, generator=g).to(cuda)
    R = torch.randn(M, N, generator=g).to(cuda)
    ref = (A.double() @ W.double().T + b.double() + R.double())
    side = torch.cuda.Stream(device=cuda)
    noise_a = [torch.randn(m, 512, device=cuda) for m in (300, 1700, 5000, 900)]
    noise_w = torch.randn(512, 512, device=cuda) / math.sqrt(512)
    first = None
    for it in range(200):
        w[COUNTER_FLOATS:] = float("nan")  # poison the piece slots (the counters stay: every launch must leave them at zero)
        with torch.cuda.stream(side):      # uneven background load of other shapes on another stream
            for a in noise_a[it % 4:] + noise_a[:it % 4]:
                torch.matmul(a, noise_w)
        Cd = torch.empty((M, N), device=cuda)
        _lib.check(lib.edv_gemm(A.data_ptr(), W.data_ptr(), Cd.data_ptr(), M, N, K, b.data_ptr(), 0, None, R.data_ptr(), w.data_ptr(), nbytes, st()), "edv_gemm")
        if first is None:
            torch.cuda.synchronize()
            first = Cd
            err = (Cd.double() - ref).abs().max().item() / ref.abs().max().item()
            assert err <= 3e-6, err
        else:
            assert torch.equal(Cd, first), f"launch {it}: a merged tile differs (torn or stale piece)"
    torch.cuda.synchronize()
    assert int(w[:COUNTER_FLOATS].view(torch.int32).abs().sum()) == 0
