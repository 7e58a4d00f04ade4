"""The encoder attention with both products on the bf16 matrix pipe (attn_x6_kernel, EDV_PRODUCTS_BF16X6) through the C-ABI: the fp32 kernel's tests,
plus the accuracy claim -- against an fp64 attention it is no worse than the fp32-MFMA kernel on the same q | k | v."""
import pytest
import torch

from endodav_amd import _lib

from .test_kernels_gpu import attn_spatial, close, rnd, st

pytestmark = pytest.mark.gpu


def attn_x6(lib, cuda, qd, o, Fr, N, heads):
    nbytes = lib.edv_attn_spatial_x6_workspace(Fr, N, heads)
    ws = torch.full((max(nbytes // 4, 4),), float("nan"), device=cuda)  # poisoned: a piece nobody wrote would show
    _lib.check(lib.edv_attn_spatial_x6(qd.data_ptr(), o.data_ptr(), Fr, N, heads, ws.data_ptr(), nbytes, st()), "edv_attn_spatial_x6")
    return nbytes


def reference(qkv, Fr, N, heads):
    t = qkv.double().reshape(Fr, N, 3, heads, 64).permute(2, 0, 3, 1, 4)
    q, k, v = t[0] * 64 ** -0.5, t[1], t[2]
    return ((q @ k.transpose(-2, -1)).softmax(-1) @ v).transpose(1, 2).reshape(Fr * N, heads * 64)


# (8, 1370, 6): the bench shape, 288 tasks of 256 queries on 256 resident workgroups = one whole round + 32 tasks split by keys; (3, 1370, 1): 18 tasks,
# all split; (1, 129, 12) and shorter run the fp32 kernel (one 256-query block would be mostly empty); (2, 321, 6): ragged last query block AND key tile
@pytest.mark.parametrize("Fr,N,heads", [(2, 1370, 6), (1, 129, 12), (1, 64, 2), (2, 321, 6), (1, 1369, 16), (8, 1370, 6), (3, 1370, 1), (40, 300, 16),
                                        (1, 4096, 1), (1, 257, 1), (5, 200, 3)])
def test_attn_x6(lib, cuda, Fr, N, heads):
    D = heads * 64
    qkv = rnd(Fr * N, 3 * D, seed=1, scale=2.0)
    ref = reference(qkv, Fr, N, heads)
    qd = qkv.to(cuda)
    o = torch.full((Fr * N, D), float("nan"), device=cuda)
    _lib.check(lib.edv_debug_fill_lds(float("nan"), st()), "edv_debug_fill_lds")  # rows past the sequence end must never reach a product as stale LDS
    attn_x6(lib, cuda, qd, o, Fr, N, heads)
    close(o, ref, 5e-6, "attn_x6")  # the gate edv_attn_spatial is held to


def test_attn_x6_peaked_rows(lib, cuda):
    """Online-softmax rescale path: one key dominates, and it sits in a LATE tile for some rows."""
    Fr, N, heads, D = 1, 300, 1, 64
    qkv = rnd(Fr * N, 3 * D, seed=7, scale=0.5)
    qkv[:, :64] *= 6.0
    qkv[250, 64:128] *= 12.0
    ref = reference(qkv, Fr, N, heads)
    qd = qkv.to(cuda)
    o = torch.empty((Fr * N, D), device=cuda)
    attn_x6(lib, cuda, qd, o, Fr, N, heads)
    close(o, ref, 5e-6, "attn_x6 peaked")


@pytest.mark.parametrize("Fr,N,heads", [(8, 1370, 6), (2, 1370, 16)])
def test_attn_x6_is_as_accurate_as_the_fp32_kernel(lib, cuda, Fr, N, heads):
    D = heads * 64
    qkv = rnd(Fr * N, 3 * D, seed=3, scale=2.0)
    ref = reference(qkv, Fr, N, heads)
    qd = qkv.to(cuda)
    o6, o32 = torch.empty(Fr * N, D, device=cuda), torch.empty(Fr * N, D, device=cuda)
    attn_x6(lib, cuda, qd, o6, Fr, N, heads)
    attn_spatial(lib, cuda, qd, o32, Fr, N, heads)
    scale = ref.abs().max().item()
    e6, e32 = (o6.double().cpu() - ref).abs() / scale, (o32.double().cpu() - ref).abs() / scale
    assert e6.pow(2).mean().sqrt().item() <= e32.pow(2).mean().sqrt().item() * 1.10, (e6.pow(2).mean().sqrt().item(), e32.pow(2).mean().sqrt().item())
    assert e6.max().item() <= max(e32.max().item() * 1.5, 1e-6), (e6.max().item(), e32.max().item())


def test_attn_x6_is_reproducible(lib, cuda):
    Fr, N, heads = 8, 1370, 6
    qd = rnd(Fr * N, 3 * heads * 64, seed=5, scale=2.0).to(cuda)
    outs = []
    for _ in range(3):
        o = torch.full((Fr * N, heads * 64), float("nan"), device=cuda)
        attn_x6(lib, cuda, qd, o, Fr, N, heads)
        outs.append(o)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


def test_attn_x6_workspace_contract(lib, cuda):
    Fr, N, heads = 8, 1370, 6
    need = lib.edv_attn_spatial_x6_workspace(Fr, N, heads)
    assert need > 0 and need % 16 == 0
    qd = torch.zeros(Fr * N, 3 * heads * 64, device=cuda)
    o = torch.empty(Fr * N, heads * 64, device=cuda)
    assert lib.edv_attn_spatial_x6(qd.data_ptr(), o.data_ptr(), Fr, N, heads, None, 0, st()) != 0
    assert "workspace" in lib.edv_last_error().decode()


def test_attn_x6_random_geometries(lib, cuda):
    """Twenty seeded random (frames, tokens, heads) with ragged query blocks and key tiles, NaN guard bands around the output."""
    g = torch.Generator().manual_seed(77)
    for case in range(20):
        Fr = int(torch.randint(1, 7, (1,), generator=g))
        N = int(torch.randint(129, 900, (1,), generator=g))
        heads = int(torch.randint(1, 9, (1,), generator=g))
        D = heads * 64
        qkv = rnd(Fr * N, 3 * D, seed=600 + case, scale=2.0)
        ref = reference(qkv, Fr, N, heads)
        qd = qkv.to(cuda)
        guard = 32
        buf = torch.full((Fr * N + 2 * guard, D), float("nan"), device=cuda)
        o = buf[guard:guard + Fr * N]
        attn_x6(lib, cuda, qd, o, Fr, N, heads)
        torch.cuda.synchronize()
        assert torch.isnan(buf[:guard]).all() and torch.isnan(buf[guard + Fr * N:]).all(), f"case {case}: F={Fr} N={N} heads={heads} wrote outside its rows"
        close(o, ref, 5e-6, f"attn_x6 case {case}: F={Fr} N={N} heads={heads}")
