"""edv_trainer_loss (csrc/loss_trainer.hip) on MI355X: the trainer's whole loss and every gradient its autograd reaches, against
(1) the known answers captured from the reference's OWN generate_images_pred / compute_losses (tests/golden/trainer_loss_kat.npz) and
(2) the PyTorch definition endodav_amd/losses.py::trainer_losses evaluated in float64 on other shapes and weight settings.

Gradient gate (no dropped percentiles).  The loss has kinks -- |.| at zero, SSIM's clamp, the border clip of the sampling
coordinates, the cell boundaries of bilinear sampling, the > 1e-3 masks -- where a float32 evaluation may legitimately take the other branch than
float64.  Pixels within an fp32-ulp band of a kink are found from the float64 graph itself: the gradient is recomputed at inputs perturbed by a few
fp32 ulps, and how far the float64 gradient of an element moves under that perturbation bounds what the kernel may differ by there (3x); every
other element must agree to 2e-4 of the tensor's scale -- every element, no dropped percentile."""
import numpy as np
import pytest
import torch

from endodav_amd import losses
from tests import helpers as H
from tests.test_losses_cpu import kat_leaves, kat_name

pytestmark = pytest.mark.gpu


def _run_hip(inp, disps, wts, cuda, want):
    inp_d = {k: v.to(cuda) for k, v in inp.items()}
    leaves = {k: v.to(cuda).clone().requires_grad_(k in want) for k, v in kat_leaves(inp, disps).items()}
    inp_d.update({k: v for k, v in leaves.items() if not (isinstance(k, tuple) and k[0] == "disp")})
    out = losses.trainer_losses_hip({k: v for k, v in leaves.items() if isinstance(k, tuple) and k[0] == "disp"}, inp_d, wts)
    out["loss"].backward()
    torch.cuda.synchronize()
    return out, leaves


def _ref64(inp, disps, wts, eps_scale=0.0, seed=0):
    """float64 values and gradients; eps_scale > 0 perturbs every leaf by that many fp32 ulps (uniform in +-), for the kink band."""
    g = torch.Generator().manual_seed(seed)
    def pert(v):
        v = v.double()
        if eps_scale:
            v = v + (torch.rand(v.shape, generator=g, dtype=torch.float64) * 2 - 1) * eps_scale * 1.2e-7 * v.abs().clamp_min(1e-3)
        return v
    inp64 = {k: pert(v) if k not in [("occu_mask_backward", 0, -1), ("occu_mask_backward", 0, 1)] else v.double() for k, v in inp.items()}
    leaves = {k: (inp64[k] if not (isinstance(k, tuple) and k[0] == "disp") else pert(disps[k])).clone().requires_grad_(True) for k in kat_leaves(inp, disps)}
    inp64.update({k: v for k, v in leaves.items() if not (isinstance(k, tuple) and k[0] == "disp")})
    out = losses.trainer_losses({k: v for k, v in leaves.items() if isinstance(k, tuple) and k[0] == "disp"}, inp64, wts)
    out["loss"].backward()
    return out, leaves


def test_trainer_loss_matches_the_references_known_answers(lib, cuda):
    from tests.golden.make_golden import TRAINER_LOSS_CASE as C

    g = H.load_golden("trainer_loss_kat")
    inp = losses.synthetic_trainer_inputs(C["n"], C["H"], C["W"], seed=C["seed"])
    disps = losses.synthetic_disps(C["n"], C["disp_sizes"], seed=C["seed"])
    wts = losses.TrainerLossWeights(**C["weights"])
    out, leaves = _run_hip(inp, disps, wts, cuda, set(kat_leaves(inp, disps)))
    for k in [k[len("value:"):] for k in g if k.startswith("value:")]:
        ref = float(g["value:" + k])
        assert abs(float(out[k]) - ref) <= 3e-6 * abs(ref) + 1e-9, (k, float(out[k]), ref)
    # gradients: bulk agreement here (the kink-aware per-pixel gate is the next test, against float64)
    for k, v in leaves.items():
        ref = torch.from_numpy(g[kat_name(k)])
        got = v.grad.cpu()
        l2 = float((got - ref).norm() / ref.norm())
        assert l2 <= 2e-3, (k, l2)


@pytest.mark.parametrize("n,Hh,W,sizes,wkw", [
    (4, 32, 48, [(28, 42), (14, 21), (7, 10), (4, 5)], dict(depth_reproj=1e-2, depth_flow=1e-3, tune_temporal=True)),
    (3, 64, 80, [(64, 80), (32, 40), (16, 20), (8, 10)], dict()),                                            # no resize anywhere; the options' defaults
    (5, 70, 98, [(70, 98), (35, 49), (17, 24), (8, 12)], dict(depth_reproj=1e-2, tune_temporal=True)),        # odd sizes, tile edges
    (2, 48, 40, [(42, 35), (21, 17), (10, 8), (5, 4)], dict(depth_flow=1e-2, tune_temporal=True, disparity_smoothness=1e-2)),
])
def test_trainer_loss_gradients_per_pixel_outside_the_kink_band(lib, cuda, n, Hh, W, sizes, wkw):
    inp = losses.synthetic_trainer_inputs(n, Hh, W, seed=11)
    disps = losses.synthetic_disps(n, sizes, seed=11)
    wts = losses.TrainerLossWeights(**wkw)
    out, leaves = _run_hip(inp, disps, wts, cuda, set(kat_leaves(inp, disps)))
    ref, rl = _ref64(inp, disps, wts)
    for k, v in ref.items():
        assert abs(float(out[k]) - float(v)) <= 5e-6 * abs(float(v)) + 1e-9, (k, float(out[k]), float(v))
    # kink band from the float64 graph: gradients at inputs moved by +-8 fp32 ulps, six draws
    moved = [_ref64(inp, disps, wts, eps_scale=8.0, seed=sd)[1] for sd in (1, 2, 3, 4, 5, 6)]
    for k, v in leaves.items():
        r = rl[k].grad
        scale = float(r.abs().max())
        if scale == 0:
            assert float(v.grad.abs().max()) == 0, k
            continue
        err = (v.grad.cpu().double() - r).abs() / scale
        # the yardstick at every element: how far the float64 gradient itself moves when the inputs move by +-8 fp32 ulps.  Away from every kink
        # that is ~1e-6 and the gate is a flat 2e-4 of the tensor's scale at EVERY element (measured worst: 1.0e-4, one pixel of 102 900); at an element fed by a kink pixel (the pose / intrinsics entries sum over whole
        # frames, a coarse disparity pixel over a block of frame pixels) the float64 result jumps and the kernel may land anywhere within 3 jumps
        spread = torch.stack([(mv[k].grad - r).abs() for mv in moved]).max(0).values / scale
        if spread.dim() == 4 and spread.shape[-1] > 4:  # a kink at one pixel reaches the gradient of its 5 x 5 neighbourhood through the SSIM windows
            spread = torch.nn.functional.max_pool2d(spread, 5, 1, 2)
        viol = err > torch.clamp(3 * spread, min=2e-4)
        assert not bool(viol.any()), f"{k}: {int(viol.sum())} of {viol.numel()} elements beyond the gate; worst {float(err[viol].max()):.3e} of scale where the float64 spread is {float(spread[viol][err[viol].argmax()]):.3e}"
        assert float(err.max()) <= 0.5, (k, float(err.max()))
        if k[0] in ("refined", "transform"):  # per-pixel tensors: the flat 1e-4 gate is the one that decides at most pixels
            assert float((3 * spread > 2e-4).float().mean()) < 0.2, (k, float((3 * spread > 2e-4).float().mean()))


def test_trainer_loss_is_reproducible_with_the_default_weights_and_refuses_bad_input(lib, cuda):
    inp = losses.synthetic_trainer_inputs(3, 32, 48, seed=5)
    disps = losses.synthetic_disps(3, [(28, 42), (14, 21), (7, 10), (4, 5)], seed=5)
    a, la = _run_hip(inp, disps, losses.TrainerLossWeights(), cuda, set(kat_leaves(inp, disps)))
    b, lb = _run_hip(inp, disps, losses.TrainerLossWeights(), cuda, set(kat_leaves(inp, disps)))
    assert all(torch.equal(a[k], b[k]) for k in a)
    assert all(torch.equal(la[k].grad, lb[k].grad) for k in la)  # no atomics without the depth-consistency terms
    # only the disparity maps require grad: the optional gradient buffers are not even allocated
    c, lc = _run_hip(inp, disps, losses.TrainerLossWeights(), cuda, {("disp", s) for s in range(4)})
    assert torch.equal(c["loss"], a["loss"])
    for s in range(4):  # (another instantiation of the geometry kernel: the same arithmetic, contracted differently by the compiler)
        assert float((lc[("disp", s)].grad - la[("disp", s)].grad).abs().max()) <= 1e-6 * float(la[("disp", s)].grad.abs().max())
    assert lc["K"].grad is None and lc[("refined", 0, 1)].grad is None
    bad = dict(inp)
    bad[("color", 0, 1)] = bad[("color", 0, 1)][..., :-1]
    with pytest.raises(ValueError, match="pyramid"):
        _run_hip(bad, disps, losses.TrainerLossWeights(), cuda, set())
