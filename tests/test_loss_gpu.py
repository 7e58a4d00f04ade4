"""edv_photometric_loss (csrc/loss.hip: the fine-tune step's loss and its gradient, fused) against the PyTorch definition
endodav_amd/losses.py::photometric_loss, which tests/test_losses_cpu.py pins to known answers captured from the reference's own loss
layers.  Reference here: the definition in float64 on the CPU (value and autograd gradients)."""
import numpy as np
import pytest
import torch

from endodav_amd import losses, synth

pytestmark = pytest.mark.gpu


def _inputs(B, T, H, W, sizes, seed=0):
    n = B * T
    frames = synth.synth_clip(B, T, H, W, seed=seed, kind="tissue").reshape(n, 3, H, W).astype(np.float32)
    noise = synth.uniform(f"loss:noise:{seed}", (n, 3, H, W), -0.05, 0.05)
    frames = np.clip(frames + noise, 0.0, 1.0).astype(np.float32)
    disps = {("disp", s): torch.from_numpy(synth.uniform(f"loss:disp{s}:{seed}", (n, 1, h, w), 0.05, 0.9)) for s, (h, w) in enumerate(sizes)}
    return torch.from_numpy(frames), disps


def _reference(disps, frames, B, T, smoothness, dtype=torch.float64):
    n, _, H, W = frames.shape
    leaves = {k: v.to(dtype).clone().requires_grad_(True) for k, v in disps.items()}
    total = 0.0
    for b in range(B):
        sl = slice(b * T, (b + 1) * T)
        K, inv_K, Tp, Tn = losses.synthetic_camera(T, H, W, "cpu", dtype)
        total = total + losses.photometric_loss({k: v[sl] for k, v in leaves.items()}, frames[sl].to(dtype), K, inv_K, Tp, Tn, disparity_smoothness=smoothness)
    total = total / B
    total.backward()
    return float(total), {k: v.grad for k, v in leaves.items()}


@pytest.mark.parametrize("B,T,H,W,sizes,smooth", [
    (1, 3, 20, 28, [(20, 28), (10, 14), (5, 7), (2, 3)], 1e-4),             # the known-answer geometry: tiles hang over every edge
    (2, 4, 70, 98, [(70, 98), (35, 49), (17, 24), (8, 12)], 1e-3),          # two clips, VDA-head pyramid, several SSIM tiles
    (1, 5, 64, 96, [(80, 112), (40, 56), (20, 28), (10, 14)], 1e-1),        # conv-head maps larger than the frames: resized down; strong smoothness term
    (1, 2, 33, 3, [(33, 3), (16, 3), (8, 3), (4, 3)], 1e-2),                # the narrowest frame the SSIM window allows, two frames
], ids=["kat_20x28", "B2T4_70x98", "convhead_64x96", "narrow_33x3"])
def test_fused_loss_matches_the_pytorch_definition(lib, cuda, B, T, H, W, sizes, smooth):
    frames, disps = _inputs(B, T, H, W, sizes)
    ref, ref_g = _reference(disps, frames, B, T, smooth)
    _, t32_g = _reference(disps, frames, B, T, smooth, torch.float32)  # the same definition in fp32: how far rounding alone moves a gradient
    n = B * T
    cams = [losses.synthetic_camera(T, H, W, cuda) for _ in range(B)]
    K, inv_K, Tp, Tn = [torch.cat([c[i] for c in cams]) for i in range(4)]
    dev = {k: v.to(cuda).requires_grad_(True) for k, v in disps.items()}
    loss = losses.photometric_loss_hip(dev, frames.to(cuda), K, inv_K, Tp, Tn, clips=B, disparity_smoothness=smooth)
    (2.0 * loss).backward()  # the upstream gradient scales dL/d disp
    assert abs(float(loss) - ref) <= 2e-5 * abs(ref), (float(loss), ref)
    for k, g in ref_g.items():
        got = dev[k].grad.cpu().double() / 2.0
        scale = g.abs().max().item()
        err = (got - g).abs()
        # bilinear sampling has a kink at every integer source coordinate: a pixel that lands within fp32 rounding of one may take the
        # neighbouring cell's slope.  Hold the bulk tightly and bound what single pixels may do.
        frac_bad = float((err > 1e-4 * scale).double().mean())
        srt = err.flatten().sort().values
        bulk = srt[: int(srt.numel() * 0.995)]
        l2_all = (err.pow(2).sum().sqrt() / g.pow(2).sum().sqrt()).item()
        l2_bulk = (bulk.pow(2).sum().sqrt() / g.pow(2).sum().sqrt()).item()
        print(f"\n[{k}] max err {err.max().item() / scale:.2e} of scale, {frac_bad:.2%} of pixels over 1e-4, {float((err > 1e-3 * scale).double().mean()):.3%} over 1e-3; "
              f"relative L2 {l2_all:.2e} (without the worst 0.5 % of pixels {l2_bulk:.2e})")
        e32 = (t32_g[k].double() - g).abs()
        bad32 = float((e32 > 1e-4 * scale).double().mean())
        print(f"      the fp32 PyTorch definition against the same fp64 graph: max err {e32.max().item() / scale:.2e} of scale, {bad32:.2%} of pixels over 1e-4")
        # The loss is piecewise smooth: |.| at 0, the clamp of SSIM, the border clip of grid_sample and the cell boundaries of its bilinear
        # interpolation are kinks, and a pixel within fp32 rounding of one takes either side's slope -- in ANY fp32 evaluation (the line
        # printed above).  Hold the bulk at rounding level and bound how many pixels may sit on a kink and what they may do.
        assert l2_bulk <= 2e-4, (k, l2_bulk)
        assert frac_bad <= max(1e-2, 3.0 * bad32) and err.max().item() <= 0.2 * scale, (k, frac_bad, err.max().item() / scale)
    # and the torch definition on the GPU in fp32 gives the same value (the path bench.py --torch-loss times)
    K1, iK1, Tp1, Tn1 = cams[0]
    t32 = sum(losses.photometric_loss({k: v[b * T:(b + 1) * T].to(cuda) for k, v in disps.items()}, frames[b * T:(b + 1) * T].to(cuda), K1, iK1, Tp1, Tn1,
                                      disparity_smoothness=smooth) for b in range(B)) / B
    assert abs(float(t32) - float(loss)) <= 2e-5 * abs(ref)


def test_fused_loss_is_reproducible_and_refuses_bad_input(lib, cuda):
    frames, disps = _inputs(1, 4, 70, 98, [(70, 98), (35, 49), (17, 24), (8, 12)], seed=3)
    K, inv_K, Tp, Tn = losses.synthetic_camera(4, 70, 98, cuda)
    outs = []
    for _ in range(3):
        dev = {k: v.to(cuda).requires_grad_(True) for k, v in disps.items()}
        loss = losses.photometric_loss_hip(dev, frames.to(cuda), K, inv_K, Tp, Tn)
        loss.backward()
        outs.append((loss.detach().clone(), [dev[("disp", s)].grad.clone() for s in range(4)]))
    for l, gs in outs[1:]:
        assert torch.equal(l, outs[0][0]) and all(torch.equal(a, b) for a, b in zip(gs, outs[0][1]))
    with pytest.raises(RuntimeError):
        losses.photometric_loss_hip(disps, frames, K.cpu(), inv_K.cpu(), Tp.cpu(), Tn.cpu())  # CPU tensors: the PyTorch definition is the CPU path
    one = {k: v[:1].to(cuda) for k, v in disps.items()}
    with pytest.raises(ValueError, match="two frames"):
        losses.photometric_loss_hip(one, frames[:1].to(cuda), K[:1], inv_K[:1], Tp[:1], Tn[:1])
