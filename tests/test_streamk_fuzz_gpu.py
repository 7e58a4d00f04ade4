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
