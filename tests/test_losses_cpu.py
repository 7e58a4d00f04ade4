"""endodav_amd/losses.py (the fine-tune step's photometric loss, PyTorch: SURVEY.md section 8f rank 4) against known answers captured from
the reference's own loss layers by tests/golden/make_golden.py losses (utils/layers.py SSIM, BackprojectDepth, Project3D,
get_smooth_loss, disp_to_depth, and their composition as in trainer_end_to_end_video.py:808-868, 899-951)."""
import numpy as np
import torch

from endodav_amd import losses
from tests import helpers as H
from tests.golden.make_golden import loss_inputs


def _inputs():
    n, Hh, W, frames, disps = loss_inputs()
    return n, Hh, W, torch.from_numpy(frames), {("disp", s): torch.from_numpy(v) for s, v in disps.items()}


def test_loss_layers_match_the_reference_known_answers():
    g = H.load_golden("loss_kat")
    n, Hh, W, frames, disps = _inputs()
    K, inv_K, Tp, Tn = losses.synthetic_camera(n, Hh, W, "cpu")
    assert np.abs(losses.ssim(frames, torch.roll(frames, 1, 0)).numpy() - g["ssim"]).max() <= 1e-6
    assert abs(float(losses.smooth_loss(disps[("disp", 0)], frames)) - float(g["smooth"])) <= 1e-7
    _, depth = losses.disp_to_depth(disps[("disp", 0)])
    assert np.abs(depth.numpy() - g["depth"]).max() <= 1e-6 * np.abs(g["depth"]).max()
    cam = losses.backproject(depth, inv_K, losses.pixel_grid(n, Hh, W, "cpu"))
    assert np.abs(cam.numpy() - g["cam_points"]).max() <= 1e-5 * np.abs(g["cam_points"]).max()
    pix = losses.project(cam, K, Tn, Hh, W)
    assert np.abs(pix.numpy() - g["pix_coords"]).max() <= 1e-5
    warped = torch.nn.functional.grid_sample(torch.roll(frames, -1, 0), pix, padding_mode="border", align_corners=True)
    assert np.abs(warped.numpy() - g["warped_next"]).max() <= 1e-5
    total = losses.photometric_loss(disps, frames, K, inv_K, Tp, Tn)
    assert abs(float(total) - float(g["total"])) <= 1e-6 * abs(float(g["total"])) + 1e-7


def test_loss_is_differentiable_in_every_scale():
    n, Hh, W, frames, disps = _inputs()
    K, inv_K, Tp, Tn = losses.synthetic_camera(n, Hh, W, "cpu")
    leaves = {k: v.clone().requires_grad_(True) for k, v in disps.items()}
    losses.photometric_loss(leaves, frames, K, inv_K, Tp, Tn).backward()
    for k, v in leaves.items():
        assert v.grad is not None and torch.isfinite(v.grad).all() and v.grad.abs().max() > 0, k


# ---- round 3: the trainer's whole loss (generate_images_pred + compute_losses, trainer_end_to_end_video.py:808-971) ----------------------------
def kat_leaves(inp, disps):
    """The tensors the reference's autograd reaches, keyed as tests/golden/trainer_loss_kat.npz names their gradients."""
    leaves = {("disp", s): disps[("disp", s)] for s in range(4)}
    leaves["K"], leaves["inv_K"] = inp["K"], inp["inv_K"]
    for fid in (-1, 1):
        leaves[("cam_T_cam", 0, fid)] = inp[("cam_T_cam", 0, fid)]
        for s in range(4):
            leaves[("refined", s, fid)] = inp[("refined", s, fid)]
            leaves[("transform", "high", s, fid)] = inp[("transform", "high", s, fid)]
    return leaves


def kat_name(k):
    return "grad:" + (k if isinstance(k, str) else ":".join(str(x) for x in k))


def test_trainer_losses_match_the_references_compute_losses():
    """Values of every entry of the trainer's `losses` dict and the gradient of every tensor its autograd reaches (four disparity maps, poses,
    K / inv_K with learn_intrinsics, refined, transform_high), captured from the reference's OWN methods by make_golden.py trainer_losses."""
    from tests.golden.make_golden import TRAINER_LOSS_CASE as C

    g = H.load_golden("trainer_loss_kat")
    inp = losses.synthetic_trainer_inputs(C["n"], C["H"], C["W"], seed=C["seed"])
    disps = losses.synthetic_disps(C["n"], C["disp_sizes"], seed=C["seed"])
    leaves = {k: v.clone().requires_grad_(True) for k, v in kat_leaves(inp, disps).items()}
    inp = {**inp, **{k: v for k, v in leaves.items() if not (isinstance(k, tuple) and k[0] == "disp")}}
    out = losses.trainer_losses({k: v for k, v in leaves.items() if isinstance(k, tuple) and k[0] == "disp"}, inp, losses.TrainerLossWeights(**C["weights"]))
    keys = [k[len("value:"):] for k in g if k.startswith("value:")]
    assert len(keys) == 4 * 7 + 1 and set(keys) == set(out)
    for k in keys:
        assert abs(float(out[k]) - float(g["value:" + k])) <= 1e-6 * abs(float(g["value:" + k])) + 1e-9, k
    for s in range(4):  # every term is live in this fixture
        for t in ("loss_reprojection", "loss_transform", "loss_cvt", "loss_smooth", "loss_depth_reproj", "loss_depth_flow"):
            assert float(g[f"value:loss/{t}/{s}"]) > 0, (t, s)
    out["loss"].backward()
    assert len([k for k in g if k.startswith("grad:")]) == len(leaves) == 24
    for k, v in leaves.items():
        ref = torch.from_numpy(g[kat_name(k)])
        assert ref.abs().max() > 0, k
        err = float((v.grad - ref).abs().max() / ref.abs().max())
        assert err <= 2e-5, (k, err)


def test_trainer_losses_default_weights_skip_the_depth_consistency_terms():
    inp = losses.synthetic_trainer_inputs(3, 24, 32, seed=1)
    disps = losses.synthetic_disps(3, [(24, 32), (12, 16), (6, 8), (3, 4)], seed=1)
    out = losses.trainer_losses(disps, inp)  # options.py defaults: depth_reproj = depth_flow = 0, tune_temporal off
    assert all(float(out[f"loss/loss_depth_reproj/{s}"]) == 0 and float(out[f"loss/loss_depth_flow/{s}"]) == 0 for s in range(4))
    assert float(out["loss"]) > 0 and torch.isfinite(out["loss"])
