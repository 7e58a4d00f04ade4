"""The "bf16 x 6" GEMM (endodav_amd/csrc/gemm_x6.hip) through the C-ABI: fp32 in, fp32 out, products on the bf16 matrix pipe.

Its claim is an accuracy claim -- every product term carries an error below fp32's unit roundoff -- so the gates are against an fp64 product and
against the fp32-MFMA kernel on the same operands, not only "close to the reference"."""
import ctypes as C
import math

import pytest
import torch
import torch.nn.functional as F

from endodav_amd import _lib

from .test_kernels_gpu import COUNTER_FLOATS, close, gemm_ws, rnd, st

pytestmark = pytest.mark.gpu


def planes_of(lib, Wd, N, K):
    nbytes = lib.edv_gemm_x6_planes_bytes(N, K)
    assert nbytes == 3 * N * K * 2
    planes = torch.empty(nbytes // 2, dtype=torch.bfloat16, device=Wd.device)
    _lib.check(lib.edv_gemm_x6_split(Wd.data_ptr(), planes.data_ptr(), N, K, st()), "edv_gemm_x6_split")
    return planes


def test_split_is_an_exact_three_term_expansion(lib, cuda):
    """w0 + w1 + w2 reproduces w to 2^-24 relative (mostly exactly), w0 is the round-to-nearest bf16 of w, each term is the bf16 of the remainder."""
    N, K = 200, 96
    W = (rnd(N, K, seed=3) * torch.logspace(-6, 3, K)).to(cuda)
    p = planes_of(lib, W, N, K).view(3, N, K)
    w0, w1, w2 = p[0].float(), p[1].float(), p[2].float()
    assert torch.equal(p[0], W.to(torch.bfloat16))
    assert torch.equal(p[1], (W - w0).to(torch.bfloat16))
    assert torch.equal(p[2], ((W - w0) - w1).to(torch.bfloat16))
    back = w0.double() + w1.double() + w2.double()
    assert ((back - W.double()).abs() <= W.double().abs() * 2.0 ** -24).all()


@pytest.mark.parametrize("M,N,K,act,use_bias,use_gamma,use_res", [
    (300, 384, 384, 0, True, False, False),
    (1370, 1152, 384, 0, True, False, False),         # qkv of one frame
    (8 * 1370, 1152, 384, 0, True, False, False),     # qkv at T=8: one whole round + 262 tiles split along k
    (8 * 1370, 384, 384, 0, True, True, True),        # proj at T=8 with LayerScale and the residual: every tile split
    (8 * 1370, 1536, 384, 1, True, False, False),     # fc1 + GELU, applied by the last piece to arrive
    (8 * 1370, 384, 1536, 0, True, True, True),       # fc2 at T=8: 96 k-steps per tile
    (4 * 1370, 768, 768, 0, True, False, True),
    (515, 200, 48, 2, True, True, False),             # ragged rows AND columns (N % 128 != 0), 3 k-steps, ReLU
    (129, 64, 16, 0, False, False, False),            # one k-step: the prologue alone
    (257, 320, 32, 0, True, False, True),             # two k-steps
    (1000, 1024, 4096, 0, True, False, False),        # ViT-L fc2 depth
])
@pytest.mark.parametrize("split", [False, True], ids=["plain", "streamk"])
def test_gemm_x6(lib, cuda, M, N, K, act, use_bias, use_gamma, use_res, split):
    A, W = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=1 / math.sqrt(K))
    bias = rnd(N, seed=3, scale=0.1) if use_bias else None
    gamma = rnd(N, seed=4) + 1.2 if use_gamma else None
    R = rnd(M, N, seed=5) if use_res else None
    ref = A.double() @ W.double().T
    if bias is not None:
        ref = ref + bias.double()
    if act == 1:
        ref = F.gelu(ref)
    elif act == 2:
        ref = F.relu(ref)
    if gamma is not None:
        ref = ref * gamma.double()
    if R is not None:
        ref = ref + R.double()
    d = lambda t: None if t is None else t.to(cuda)
    Ad, Wd, bd, gd, Rd = d(A), d(W), d(bias), d(gamma), d(R)
    planes = planes_of(lib, Wd, N, K)
    Cd = torch.full((M, N), float("nan"), device=cuda)
    ws, nbytes = gemm_ws(lib, cuda) if split else (None, 0)
    _lib.check(lib.edv_gemm_x6(Ad.data_ptr(), planes.data_ptr(), Cd.data_ptr(), M, N, K, _lib.ptr(bd), act, _lib.ptr(gd), _lib.ptr(Rd), _lib.ptr(ws), nbytes, st()),
               "edv_gemm_x6")
    torch.cuda.synchronize()
    if split:
        assert int(ws[:COUNTER_FLOATS].view(torch.int32).abs().sum()) == 0  # every launch leaves the arrival counters at zero
    close(Cd, ref, 3e-6, f"gemm_x6 {M}x{N}x{K}")  # the gate edv_gemm is held to


@pytest.mark.parametrize("M,N,K", [(8 * 1370, 1152, 384), (8 * 1370, 384, 1536), (2048, 1024, 4096)])
def test_gemm_x6_is_as_accurate_as_the_fp32_pipe(lib, cuda, M, N, K):
    """Error against fp64 relative to sum_k |a_k w_k| (what a dot product's rounding is judged by): the bf16 x 6 result must be no worse than the
    fp32-MFMA kernel's on the same operands, in rms and in the maximum, and both within a few fp32 unit roundoffs (2^-24 = 5.96e-8)."""
    A, W = (rnd(M, K, seed=11) * 2).to(cuda), rnd(N, K, seed=12, scale=0.05).to(cuda)
    planes = planes_of(lib, W, N, K)
    ws, nbytes = gemm_ws(lib, cuda)
    C6, C32 = torch.empty(M, N, device=cuda), torch.empty(M, N, device=cuda)
    _lib.check(lib.edv_gemm_x6(A.data_ptr(), planes.data_ptr(), C6.data_ptr(), M, N, K, None, 0, None, None, ws.data_ptr(), nbytes, st()), "edv_gemm_x6")
    _lib.check(lib.edv_gemm(A.data_ptr(), W.data_ptr(), C32.data_ptr(), M, N, K, None, 0, None, None, ws.data_ptr(), nbytes, st()), "edv_gemm")
    rows = torch.arange(0, M, 7, device=cuda)
    ref = A[rows].double() @ W.double().T
    mag = A[rows].double().abs() @ W.double().abs().T
    e6 = ((C6[rows].double() - ref).abs() / mag)
    e32 = ((C32[rows].double() - ref).abs() / mag)
    rms6, rms32 = e6.pow(2).mean().sqrt().item(), e32.pow(2).mean().sqrt().item()
    assert rms6 <= rms32 * 1.05, (rms6, rms32)
    assert e6.max().item() <= max(e32.max().item() * 1.25, 4 * 2.0 ** -24), (e6.max().item(), e32.max().item())
    assert e6.max().item() <= 8 * 2.0 ** -24


def test_gemm_x6_streamk_is_reproducible(lib, cuda):
    ws, nbytes = gemm_ws(lib, cuda)
    for M, N, K in ((8 * 1370, 384, 1536), (8 * 1370, 1152, 384)):
        A, W = rnd(M, K, seed=1).to(cuda), rnd(N, K, seed=2, scale=1 / math.sqrt(K)).to(cuda)
        planes = planes_of(lib, W, N, K)
        outs = []
        for _ in range(3):
            Cd = torch.full((M, N), float("nan"), device=cuda)
            _lib.check(lib.edv_gemm_x6(A.data_ptr(), planes.data_ptr(), Cd.data_ptr(), M, N, K, None, 0, None, None, ws.data_ptr(), nbytes, st()), "edv_gemm_x6")
            outs.append(Cd)
        torch.cuda.synchronize()
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


def test_gemm_x6_inplace_residual(lib, cuda):
    """proj / fc2 write the residual stream in place (C aliases R)."""
    M, N, K = 8 * 1370, 384, 384
    ws, nbytes = gemm_ws(lib, cuda)
    A, W, X = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.05), rnd(M, N, seed=3)
    ref = X.double() + A.double() @ W.double().T
    Ad, Wd, Xd = A.to(cuda), W.to(cuda), X.to(cuda)
    planes = planes_of(lib, Wd, N, K)
    _lib.check(lib.edv_gemm_x6(Ad.data_ptr(), planes.data_ptr(), Xd.data_ptr(), M, N, K, None, 0, None, Xd.data_ptr(), ws.data_ptr(), nbytes, st()))
    close(Xd, ref, 3e-6, "in-place residual")


def test_gemm_x6_rejects_what_it_does_not_cover(lib, cuda):
    a = torch.zeros(256, 24, device=cuda)
    p = torch.zeros(3 * 256 * 24, dtype=torch.bfloat16, device=cuda)
    assert lib.edv_gemm_x6(a.data_ptr(), p.data_ptr(), a.data_ptr(), 256, 256, 24, None, 0, None, None, None, 0, st()) != 0  # K % 16
    assert b"K % 16" in lib.edv_last_error()
    assert lib.edv_gemm_x6(a.data_ptr(), p.data_ptr(), a.data_ptr(), 256, 32, 16, None, 0, None, None, None, 0, st()) != 0  # N < 64


def test_gemm_x6_random_geometries(lib, cuda):
    """Forty seeded random shapes (ragged M and N, 1 .. 40 k-steps, every epilogue combination, with and without the stream-K workspace) against fp64."""
    g = torch.Generator().manual_seed(2024)
    ws, nbytes = gemm_ws(lib, cuda)
    for case in range(40):
        M = int(torch.randint(1, 3000, (1,), generator=g))
        N = int(torch.randint(64, 700, (1,), generator=g))
        K = 16 * int(torch.randint(1, 41, (1,), generator=g))
        act = int(torch.randint(0, 3, (1,), generator=g))
        flags = [bool(torch.randint(0, 2, (1,), generator=g)) for _ in range(4)]  # bias, gamma, residual, workspace
        A, W = rnd(M, K, seed=100 + case), rnd(N, K, seed=200 + case, scale=1 / math.sqrt(K))
        bias = rnd(N, seed=300 + case, scale=0.1) if flags[0] else None
        gamma = rnd(N, seed=400 + case) + 1.2 if flags[1] else None
        R = rnd(M, N, seed=500 + case) if flags[2] else None
        ref = A.double() @ W.double().T
        if bias is not None:
            ref = ref + bias.double()
        ref = F.gelu(ref) if act == 1 else (F.relu(ref) if act == 2 else ref)
        if gamma is not None:
            ref = ref * gamma.double()
        if R is not None:
            ref = ref + R.double()
        d = lambda t: None if t is None else t.to(cuda)
        Ad, Wd, bd, gd, Rd = d(A), d(W), d(bias), d(gamma), d(R)
        planes = planes_of(lib, Wd, N, K)
        guard = 64
        buf = torch.full((M + 2 * guard, N), float("nan"), device=cuda)  # NaN guard bands of 64 rows on both sides
        Cd = buf[guard:guard + M]
        w, nb = (ws, nbytes) if flags[3] else (None, 0)
        _lib.check(lib.edv_gemm_x6(Ad.data_ptr(), planes.data_ptr(), Cd.data_ptr(), M, N, K, _lib.ptr(bd), act, _lib.ptr(gd), _lib.ptr(Rd), _lib.ptr(w), nb, st()),
                   f"edv_gemm_x6 case {case}: {M}x{N}x{K}")
        torch.cuda.synchronize()
        assert torch.isnan(buf[:guard]).all() and torch.isnan(buf[guard + M:]).all(), f"case {case}: {M}x{N}x{K} wrote outside its rows"
        close(Cd, ref, 3e-6, f"gemm_x6 case {case}: {M}x{N}x{K} act {act} {flags}")
