"""Fine-tune step on MI355X: edv_forward under edv_set_train + edv_backward against torch
autograd through the CPU oracle on the same weights, inputs and upstream gradients.

Gate: every LoRA-factor gradient within 2e-4 of the largest entry of its reference tensor (micro configurations, fp32
oracle) / 1e-3 (full size, fp64 oracle).  The gradient passes through ~60 ReLU stages whose masks can flip at fp32-noise
zero crossings; see ``upstream`` for why the tests feed a gradient with a definite sign, and test_bwd_kernels_gpu.py for
the per-kernel gates (2e-6 .. 1e-5, random-sign gradients)."""
import numpy as np
import pytest
import torch

import endodav_amd
from endodav_amd import synth
from oracle import endodav_oracle as orc
from tests.helpers import build_model, case_input, oracle_config

pytestmark = pytest.mark.gpu

FACTORS = ("lora_A", "lora_B", "lora_U", "lora_V")


def upstream(shapes, seed=5, signed=False):
    """dL/d("disp", k).  Default 1 + 0.5 U(-1, 1): a gradient with a definite sign, as a loss has.  ``signed=True`` gives
    U(-1, 1): then the factor gradients are an incoherent sum over the output pixels and ONE ReLU mask that flips at an
    fp32-noise zero crossing moves all of them by ~1/sqrt(pixels) (1e-3 .. 1e-2 here) -- in the fp32 reference as well."""
    g = [torch.from_numpy(synth.uniform(f"gout{k}", tuple(s), -1.0, 1.0, seed=seed)) for k, s in enumerate(shapes)]
    return g if signed else [1.0 + 0.5 * t for t in g]


def oracle_grads(model, kwargs, x, names, gouts, dtype=torch.float32):
    sd = {k: (v.detach().cpu().clone().to(dtype) if v.is_floating_point() else v.detach().cpu().clone()) for k, v in model.state_dict().items()}
    for n in names:
        sd[n].requires_grad_(True)
    out = orc.forward(sd, x.to(dtype), oracle_config(kwargs))
    loss = sum((out[("disp", s)] * gouts[s].to(dtype)).sum() for s in range(4))
    return dict(zip(names, torch.autograd.grad(loss, [sd[n] for n in names]))), out


def hip_grads(model, x, names, gouts, cuda):
    model.zero_grad(set_to_none=True)
    out = model(x.to(cuda))
    loss = sum((out[("disp", s)] * gouts[s].to(cuda)).sum() for s in range(4))
    loss.backward()
    sd = model.state_dict(keep_vars=True)
    return {n: sd[n].grad for n in names}, out


def set_trainable(model, tags):
    names = []
    for n, p in model.named_parameters():
        p.requires_grad = (".mlp.fc" in n or ".ff.net.2." in n) and n.rsplit(".", 1)[-1] in tags
        if p.requires_grad:
            names.append(n)
    return names


def check(hip, ref, tol=2e-4):
    worst = 0.0
    for n, r in ref.items():
        g = hip[n]
        assert g is not None and g.shape == r.shape, n
        err = (g.cpu().double() - r.double()).abs().max().item() / max(r.abs().max().item(), 1e-30)
        worst = max(worst, err)
        assert err <= tol, f"{n}: scale-relative gradient error {err:.2e} > {tol:.0e}"
    return worst


@pytest.mark.parametrize("case", ["micro_vda_dvlora", "micro_vda_lora_b2", "micro_t1", "micro_vda_temporal_lora", "micro_rope", "micro_vda_nocls", "micro_vitl",
                                  "micro_resize_in", "micro_clstoken"])
def test_lora_gradients_match_oracle_autograd(lib, cuda, case):
    model, kwargs, shape, kind, _ = build_model(case)
    x = case_input(case)
    names = set_trainable(model, FACTORS)
    assert names
    model = model.to(cuda).train()
    BT = shape[0] * shape[1]
    gouts = upstream([(BT, 1, h, w) for (h, w) in model.output_shapes()])
    ref, out_ref = oracle_grads(model, kwargs, x, names, gouts)
    hip, out = hip_grads(model, x, names, gouts, cuda)
    for s in range(4):  # the training forward is the inference forward
        a, b = out[("disp", s)].detach().cpu(), out_ref[("disp", s)].detach()
        assert (a - b).abs().max().item() <= 5e-5 * b.abs().max().item()
    worst = check(hip, ref)
    print(f"\n[{case}] {len(names)} tensors, worst scale-relative gradient error {worst:.2e}")


@pytest.mark.parametrize("case,head_key", [("micro_conv_dvlora", "head.conv_depth_"), ("micro_conv_invsig_ssb", "head.conv_depth_"),
                                           ("micro_vda_dvlora", "head.scratch.output_conv"), ("micro_clstoken_nocls", "head.conv_depth_")],
                         ids=["conv_head", "conv_head_inv_sigmoid_ssb", "vda_train_output_conv", "conv_head_use_clstoken_no_cls_token"])
def test_output_head_conv_gradients(lib, cuda, case, head_key):
    """The reference's default options leave the four HeadDepth heads trainable next to the LoRA factors (endodav/layers.py:5-34:
    names containing conv_depth_), and --train_output_conv does the same for scratch.output_conv* of the VDA head: weight and bias
    gradients of those convolutions (conv3_wgrad / colsum_rows) and the LoRA gradients that now collect from all four paths."""
    model, kwargs, shape, kind, _ = build_model(case)
    x = case_input(case)
    names = []
    for n, p in model.named_parameters():
        p.requires_grad = ((".mlp.fc" in n and n.rsplit(".", 1)[-1] in ("lora_A", "lora_B")) or n.startswith(head_key))
        if p.requires_grad:
            names.append(n)
    n_head = sum(n.startswith(head_key) for n in names)
    assert n_head == (24 if "conv_depth" in head_key else 6), names
    model = model.to(cuda).train()
    BT = shape[0] * shape[1]
    gouts = upstream([(BT, 1, h, w) for (h, w) in model.output_shapes()])
    # fp64 graph: in the use_clstoken case the fp32 CPU oracle itself sits 3e-4 .. 6e-4 from it (one ReLU mask at an fp32-noise zero
    # crossing, scratch/dbg_clstoken_grad.py) while the HIP gradients are within 1e-5
    ref, out_ref = oracle_grads(model, kwargs, x, names, gouts, torch.float64)
    hip, out = hip_grads(model, x, names, gouts, cuda)
    for s in range(4):
        a, b = out[("disp", s)].detach().cpu().double(), out_ref[("disp", s)].detach()
        assert (a - b).abs().max().item() <= 5e-5 * b.abs().max().item()
    worst = check(hip, ref)
    print(f"\n[{case}] {len(names)} tensors ({n_head} head convolution tensors), worst scale-relative gradient error vs the fp64 graph {worst:.2e}")


@pytest.mark.parametrize("active", [False, True], ids=["warm_up", "active"])
def test_dash_gradients(lib, cuda, active):
    """DashLinear (mylora/layers.py:497-583): plain LoRA (lora_alpha = 2r) during its 100-call warm-up; from call 101 on the layer adds
    x (U_top diag(lora_index) Vt_top)^T and frees lora_index.  lora_A / lora_B keep LoRA's gradient, lora_index gets its own."""
    case = "micro_dash_active" if active else "micro_dash"
    model, kwargs, shape, kind, _ = build_model(case)
    x = case_input(case)
    model = model.to(cuda).train()
    if active:
        with torch.no_grad():
            model(x.to(cuda))  # call 101: direction selection, lora_index freed
        assert all(m.lora_index.requires_grad for m in model._dash_layers())
    names = []
    for n, p in model.named_parameters():
        p.requires_grad = ".mlp.fc" in n and n.rsplit(".", 1)[-1] in (("lora_A", "lora_B", "lora_index") if active else ("lora_A", "lora_B"))
        if p.requires_grad:
            names.append(n)
    assert len(names) == (72 if active else 48)
    BT = shape[0] * shape[1]
    gouts = upstream([(BT, 1, h, w) for (h, w) in model.output_shapes()])
    sd = {k: (v.detach().cpu().clone()) for k, v in model.state_dict().items()}
    for n in names:
        sd[n].requires_grad_(True)
    out_ref = orc.forward(sd, x, oracle_config(kwargs, dash_active=active))
    loss = sum((out_ref[("disp", s)] * gouts[s]).sum() for s in range(4))
    ref = dict(zip(names, torch.autograd.grad(loss, [sd[n] for n in names])))
    calls = model._dash_calls
    hip, out = hip_grads(model, x, names, gouts, cuda)
    assert model._dash_calls == calls + 1
    for s in range(4):
        a, b = out[("disp", s)].detach().cpu(), out_ref[("disp", s)].detach()
        assert (a - b).abs().max().item() <= 5e-5 * b.abs().max().item()
    worst = check(hip, ref)
    print(f"\n[dash {'active' if active else 'warm-up'}] {len(names)} tensors, worst scale-relative gradient error {worst:.2e}")


@pytest.mark.parametrize("lora_type", ["none", "dvlora"])
def test_out_sigmoid_gradients(lib, cuda, lora_type):
    """--out_sigmoid with the VDA head (dpt_pyramid.py:97-101): every scale passes through a sigmoid after the downsampling chain.
    lora_type none: only scratch.output_conv* trainable (--train_output_conv), the backward stops at the head; dvlora: LoRA factors too."""
    from tests.golden.cases import VITS_SMALL_HEAD

    kwargs = dict(VITS_SMALL_HEAD, image_shape=(42, 56), lora_type=lora_type, disable_conv_head=True, out_sigmoid=True)
    model = endodav_amd.endodav(**kwargs, pretrained_path=None)
    synth.fill_module_(model)
    names = []
    for n, p in model.named_parameters():
        p.requires_grad = n.startswith("head.scratch.output_conv") or (".mlp.fc" in n and n.rsplit(".", 1)[-1] in ("lora_A", "lora_B"))
        if p.requires_grad:
            names.append(n)
    assert len(names) == (6 if lora_type == "none" else 54)
    x = torch.from_numpy(synth.synth_clip(1, 3, 42, 56, seed=2, kind="tissue"))
    model = model.to(cuda).train()
    gouts = upstream([(3, 1, h, w) for (h, w) in model.output_shapes()])
    ref, out_ref = oracle_grads(model, kwargs, x, names, gouts)
    hip, out = hip_grads(model, x, names, gouts, cuda)
    for s in range(4):
        a, b = out[("disp", s)].detach().cpu(), out_ref[("disp", s)].detach()
        assert (a - b).abs().max().item() <= 5e-5 * b.abs().max().item()
    worst = check(hip, ref)
    print(f"\n[out_sigmoid, lora_type {lora_type}] {len(names)} tensors, worst scale-relative gradient error {worst:.2e}")


@pytest.mark.parametrize("temporal", [False, True], ids=["spatial", "spatial+temporal"])
def test_ssb_gradients(lib, cuda, temporal):
    """Linear_SSB (lora_A [in,1], lora_B [out,1]), the lora_type of the reference's scripts/train_video.sh, with and without
    --temporal_lora (ff.net.2 of the four motion modules)."""
    from tests.golden.cases import VITS_SMALL_HEAD

    kwargs = dict(VITS_SMALL_HEAD, image_shape=(42, 56), lora_type="ssb", disable_conv_head=True, temporal_lora=temporal)
    model = endodav_amd.endodav(**kwargs, pretrained_path=None)
    synth.fill_module_(model)
    names = set_trainable(model, ("lora_A", "lora_B"))
    assert len(names) == 48 + (8 if temporal else 0)
    x = torch.from_numpy(synth.synth_clip(1, 3, 42, 56, seed=2, kind="tissue"))
    model = model.to(cuda).train()
    gouts = upstream([(3, 1, h, w) for (h, w) in model.output_shapes()])
    ref, _ = oracle_grads(model, kwargs, x, names, gouts)
    hip, _ = hip_grads(model, x, names, gouts, cuda)
    worst = check(hip, ref)
    print(f"\n[ssb temporal={temporal}] {len(names)} tensors, worst scale-relative gradient error {worst:.2e}")


@pytest.mark.parametrize("H,W,T,resize_from", [(224, 280, 2, (256, 320)), (518, 518, 1, None)], ids=["224x280_T2", "518x518_T1"])
def test_vits_gradients_full_size(lib, cuda, H, W, T, resize_from):
    """ViT-S at the trainer's 256x320 -> (224, 280) geometry (BASELINE config 4) and at 518x518, against the oracle's
    autograd in fp64.

    The upstream gradient has a definite sign (see ``upstream``).  With a random-SIGN upstream gradient the factor
    gradients are an incoherent sum over ~10^5 output pixels and a single ReLU whose pre-activation sits within fp32
    noise of zero (here: one element of 4 M in output_conv2, +1e-5 on the GPU, exactly 0 in fp64) moves every one of
    them by ~1e-3 of its scale -- the fp32 CPU oracle shows the same sensitivity (1e-3 .. 7e-3 against its own fp64 run
    on other inputs).  A loss gradient with a definite sign, as a real loss has, is not hostage to one mask flip:
    the same GPU gradients are then within 1e-5 .. 3e-4 of the fp64 graph, run to run with the summation order of the
    kernels (gate 1e-3; the micro cases above hold 2e-4 with a random-sign gradient and the per-kernel tests 1e-5)."""
    kwargs = dict(encoder="vits", features=64, out_channels=[48, 96, 192, 384], image_shape=(H, W), lora_type="dvlora", disable_conv_head=True)
    model = endodav_amd.endodav(**kwargs, pretrained_path=None)
    synth.fill_module_(model)
    names = set_trainable(model, FACTORS)
    h_in, w_in = resize_from or (H, W)
    x = torch.from_numpy(synth.synth_clip(1, T, h_in, w_in, seed=3, kind="tissue"))
    model = model.to(cuda).train()
    gouts = upstream([(T, 1, h, w) for (h, w) in model.output_shapes()])
    ref64, _ = oracle_grads(model, kwargs, x, names, gouts, torch.float64)
    hip, _ = hip_grads(model, x, names, gouts, cuda)
    worst = check(hip, ref64, tol=1e-3)
    print(f"\n[vits {H}x{W} T={T}] {len(names)} tensors, worst scale-relative gradient error vs the fp64 graph {worst:.2e}")


def test_vits_conv_head_gradients_full_size(lib, cuda):
    """The reference's default head (four HeadDepth heads, conv_depth_* trainable next to the LoRA factors) at the trainer's
    256x320 -> (224, 280) geometry with ViT-S's real head width (features 64: 64 -> 32 -> 32 -> 1 per head), fp64 oracle graph."""
    kwargs = dict(encoder="vits", features=64, out_channels=[48, 96, 192, 384], image_shape=(224, 280), lora_type="dvlora")
    model = endodav_amd.endodav(**kwargs, pretrained_path=None)
    synth.fill_module_(model)
    names = []
    for n, p in model.named_parameters():
        p.requires_grad = (".mlp.fc" in n and n.rsplit(".", 1)[-1] in ("lora_A", "lora_B")) or n.startswith("head.conv_depth_")
        if p.requires_grad:
            names.append(n)
    x = torch.from_numpy(synth.synth_clip(1, 2, 256, 320, seed=3, kind="tissue"))
    model = model.to(cuda).train()
    gouts = upstream([(2, 1, h, w) for (h, w) in model.output_shapes()])
    ref64, _ = oracle_grads(model, kwargs, x, names, gouts, torch.float64)
    hip, _ = hip_grads(model, x, names, gouts, cuda)
    worst = check(hip, ref64, tol=1e-3)
    print(f"\n[vits conv head 224x280 T=2] {len(names)} tensors, worst scale-relative gradient error vs the fp64 graph {worst:.2e}")


def test_reference_default_trainable_set_gradients(lib, cuda):
    """What the reference trains with its default options (no --disable_* flags; trainer_end_to_end_video.py:337 ->
    endodav/layers.py:5-34): LoRA A/B of every MLP, every parameter of the ResBottleneckBlocks in encoder blocks 2/5/8/11
    (residual_*; they exist at image_shape (224, 280) only) and the four HeadDepth heads (conv_depth_*).  The trainable set is the
    one mark_only_part_as_trainable leaves, unchanged; fp64 oracle graph."""
    kwargs = dict(encoder="vits", features=32, out_channels=[32, 32, 64, 64], image_shape=(224, 280), lora_type="dvlora", residual_block_indexes=[2, 5, 8, 11])
    model = endodav_amd.endodav(**kwargs, pretrained_path=None)
    synth.fill_module_(model)
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    assert sum(".residual_." in n for n in names) == 4 * 9 and sum("conv_depth_" in n for n in names) == 24 and sum("lora_" in n for n in names) == 48
    x = torch.from_numpy(synth.synth_clip(1, 2, 224, 280, seed=3, kind="tissue"))
    model = model.to(cuda).train()
    gouts = upstream([(2, 1, h, w) for (h, w) in model.output_shapes()])
    ref64, _ = oracle_grads(model, kwargs, x, names, gouts, torch.float64)
    hip, _ = hip_grads(model, x, names, gouts, cuda)
    worst = check(hip, ref64, tol=1e-3)
    print(f"\n[reference default trainable set, 224x280 T=2] {len(names)} tensors, worst scale-relative gradient error vs the fp64 graph {worst:.2e}")


def test_gradients_on_an_odd_geometry(lib, cuda):
    """Two clips of three frames on a non-square 9 x 13 patch grid, conv head, ssb + temporal_lora, everything the reference would train
    in that configuration: a geometry and an option mix no other gradient test has; fp64 oracle graph."""
    kwargs = dict(encoder="vits", features=64, out_channels=[48, 96, 192, 384], image_shape=(126, 182), lora_type="ssb", temporal_lora=True)
    model = endodav_amd.endodav(**kwargs, pretrained_path=None)
    synth.fill_module_(model)
    names = []
    for n, p in model.named_parameters():
        p.requires_grad = ((".mlp.fc" in n or ".ff.net.2." in n) and n.rsplit(".", 1)[-1] in ("lora_A", "lora_B")) or n.startswith("head.conv_depth_")
        if p.requires_grad:
            names.append(n)
    x = torch.from_numpy(synth.synth_clip(2, 3, 150, 200, seed=6, kind="tissue"))
    model = model.to(cuda).train()
    gouts = upstream([(6, 1, h, w) for (h, w) in model.output_shapes()])
    ref64, _ = oracle_grads(model, kwargs, x, names, gouts, torch.float64)
    hip, _ = hip_grads(model, x, names, gouts, cuda)
    worst = check(hip, ref64, tol=5e-4)
    print(f"\n[126x182 B=2 T=3 conv head ssb + temporal_lora] {len(names)} tensors, worst scale-relative gradient error vs the fp64 graph {worst:.2e}")


def test_training_forward_equals_inference(lib, cuda):
    """Same kernels, same values; the inference path only orders the head differently (the fusion blocks' skip branches
    run on a second stream and are added where x is produced), which moves the result by fp32 rounding."""
    model, kwargs, shape, kind, _ = build_model("micro_vda_dvlora")
    x = case_input("micro_vda_dvlora").to(cuda)
    set_trainable(model, FACTORS)
    model = model.to(cuda)
    model.products = "f32"  # the training forward always computes fp32 products: compare like with like
    with torch.no_grad():
        ref = model(x)
    out = model(x)
    assert out[("disp", 0)].requires_grad
    for s in range(4):
        a, b = out[("disp", s)].detach(), ref[("disp", s)]
        assert (a - b).abs().max().item() <= 2e-6 * b.abs().max().item()
    # the inference default (bf16 x 6 products in the encoder) is a different fp32-accurate arithmetic: rounding-level apart, not bit-equal
    model.products = "bf16x6"
    with torch.no_grad():
        ref6 = model(x)
    for s in range(4):
        a, b = out[("disp", s)].detach(), ref6[("disp", s)]
        assert (a - b).abs().max().item() <= 1e-5 * b.abs().max().item()


def test_freeze_schedule_selects_the_gradient_set(lib, cuda):
    """mark_only_part_as_trainable (endodav/layers.py:5-34): A/B during warm-up, U/V afterwards."""
    model, kwargs, shape, kind, _ = build_model("micro_vda_dvlora")
    x = case_input("micro_vda_dvlora").to(cuda)
    model = model.to(cuda)
    for warm, on, off in ((True, ("lora_A", "lora_B"), ("lora_U", "lora_V")), (False, ("lora_U", "lora_V"), ("lora_A", "lora_B"))):
        endodav_amd.mark_only_part_as_trainable(model, warm_up=warm)
        model.zero_grad(set_to_none=True)
        out = model(x)
        sum(o.sum() for o in out.values()).backward()
        for n, p in model.named_parameters():
            tag = n.rsplit(".", 1)[-1]
            if (".mlp.fc" in n or ".ff.net.2." in n) and tag in on:
                assert p.grad is not None and torch.isfinite(p.grad).all() and p.grad.abs().max() > 0, n
            else:
                assert p.grad is None, n


def test_gradient_is_linear_in_the_upstream_gradient_at_full_size(lib, cuda):
    """Size-independent property at BASELINE's full size (ViT-S 518x518 T=8, no oracle run): the backward is a linear map of
    dL/d disp.  Doubling the upstream gradient doubles every factor gradient bit for bit (a power of two commutes with
    rounding); the gradient of a sum is the sum of the gradients up to summation order."""
    kwargs = dict(encoder="vits", features=64, out_channels=[48, 96, 192, 384], image_shape=(518, 518), lora_type="dvlora", disable_conv_head=True)
    model = endodav_amd.endodav(**kwargs, pretrained_path=None)
    synth.fill_module_(model)
    names = set_trainable(model, FACTORS)
    x = torch.from_numpy(synth.synth_clip(1, 8, 518, 518, seed=6, kind="tissue"))
    model = model.to(cuda).train()
    shapes = [(8, 1, h, w) for (h, w) in model.output_shapes()]
    ga, gb = upstream(shapes, seed=5), upstream(shapes, seed=9, signed=True)
    clone = lambda d: {n: g.clone() for n, g in d.items()}
    g1 = clone(hip_grads(model, x, names, ga, cuda)[0])
    g2 = clone(hip_grads(model, x, names, [2.0 * g for g in ga], cuda)[0])
    g3 = clone(hip_grads(model, x, names, gb, cuda)[0])
    g4 = clone(hip_grads(model, x, names, [a + b for a, b in zip(ga, gb)], cuda)[0])
    for n in names:
        assert torch.isfinite(g1[n]).all() and g1[n].abs().max() > 0, n
        assert torch.equal(g2[n], 2.0 * g1[n]), n
        ref = g1[n] + g3[n]
        assert (g4[n] - ref).abs().max().item() <= 2e-5 * max(ref.abs().max().item(), g1[n].abs().max().item()), n


def test_temporal_only_phase_stops_at_the_head(lib, cuda):
    """The trainer's temporal tuning phase (trainer_end_to_end_video.py:327-339): only ff.net.2 factors are trainable, the
    backward does not enter the encoder, and the gradients equal those of a full backward."""
    model, kwargs, shape, kind, _ = build_model("micro_vda_temporal_lora")
    x = case_input("micro_vda_temporal_lora")
    model = model.to(cuda).train()
    gouts = upstream([(shape[0] * shape[1], 1, h, w) for (h, w) in model.output_shapes()])
    all_names = set_trainable(model, FACTORS)
    full, _ = hip_grads(model, x, all_names, gouts, cuda)
    full = {n: g.clone() for n, g in full.items()}
    n_full = model.launch_count()
    t_names = [n for n in all_names if ".ff.net.2." in n]
    for n, p in model.named_parameters():
        p.requires_grad = n in t_names
    part, _ = hip_grads(model, x, t_names, gouts, cuda)
    assert model.launch_count() < n_full  # the encoder backward did not run
    for n in t_names:
        assert torch.equal(part[n], full[n]), n


def test_optimizer_step_is_seen_by_the_next_forward(lib, cuda):
    model, kwargs, shape, kind, _ = build_model("micro_vda_lora_b2")
    x = case_input("micro_vda_lora_b2").to(cuda)
    model = model.to(cuda)
    endodav_amd.mark_only_part_as_trainable(model, warm_up=True)
    opt = torch.optim.SGD([p for p in model.parameters() if p.requires_grad], lr=1e-2)
    target = torch.zeros(4, 1, 42, 42, device=cuda)
    losses = []
    for _ in range(4):
        opt.zero_grad(set_to_none=True)
        loss = ((model(x)[("disp", 0)] - target) ** 2).mean()
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert losses[-1] < losses[0], losses  # plain gradient descent on a quadratic of the output


def test_refresh_after_optimizer_step_equals_a_full_prepare(lib, cuda):
    """After optimizer.step() the host sees that only trainable tensors changed version and calls edv_refresh_lora (LoRA folds, the
    trainable convolutions' packings and their backward copies) instead of edv_prepare.  The refreshed context must behave exactly
    like a freshly prepared one: same outputs, same gradients, bit for bit."""
    kwargs = dict(encoder="vits", features=32, out_channels=[32, 32, 64, 64], image_shape=(224, 280), lora_type="dvlora", residual_block_indexes=[2, 5, 8, 11])
    model = endodav_amd.endodav(**kwargs, pretrained_path=None)
    synth.fill_module_(model)
    model = model.to(cuda).train()
    x = torch.from_numpy(synth.synth_clip(1, 2, 224, 280, seed=4, kind="tissue")).to(cuda)
    params = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.SGD(params, lr=1e-3)

    def step_grads():
        model.zero_grad(set_to_none=True)
        out = model(x)
        sum((o * o).mean() for o in out.values()).backward()
        return [o.detach().clone() for o in out.values()], [p.grad.detach().clone() for p in params]

    for _ in range(3):
        step_grads()
        opt.step()
    out_a, g_a = step_grads()          # context kept across the steps: refresh path
    model._native.clear()              # drop it: the next forward binds and prepares from scratch
    out_b, g_b = step_grads()
    assert all(torch.equal(a, b) for a, b in zip(out_a, out_b))
    assert all(torch.equal(a, b) for a, b in zip(g_a, g_b))


def test_unsupported_trainable_parameters_are_refused(lib, cuda):
    model, kwargs, shape, kind, _ = build_model("micro_vda_dvlora")
    x = case_input("micro_vda_dvlora").to(cuda)
    model = model.to(cuda)
    for p in model.parameters():
        p.requires_grad = False
    model.get_parameter("head.scratch.layer1_rn.weight").requires_grad = True
    with pytest.raises(NotImplementedError, match="no gradient for"):
        model(x)
    with torch.no_grad():
        model(x)  # inference is unaffected


# ---- round 2: one set of kept activations per context; gradients land in one flat buffer -----------------------------------------
def test_backward_of_an_overwritten_forward_is_refused(lib, cuda):
    """A context keeps ONE set of activations.  fwd(x1), fwd(x2), loss1.backward() used to differentiate the second clip silently (plausible
    but wrong LoRA gradients); now the backward names its forward (edv_generation) and the engine refuses a mismatch.  The latest
    forward's backward still works, and so does the usual one-forward-per-step loop afterwards."""
    model, kwargs, shape, kind, _ = build_model("micro_vda_dvlora")
    x1 = case_input("micro_vda_dvlora").to(cuda)
    x2 = torch.flip(x1, dims=[1]).contiguous() * 0.5
    names = set_trainable(model, FACTORS)
    model = model.to(cuda).train()
    out1 = model(x1)
    out2 = model(x2)  # overwrites the activations of the first forward
    with pytest.raises(RuntimeError, match="overwrote"):
        sum(o.sum() for o in out1.values()).backward()
    sum(o.sum() for o in out2.values()).backward()  # the activations in the context are this forward's
    sd = model.state_dict(keep_vars=True)
    g2 = {n: sd[n].grad.clone() for n in names}
    model.zero_grad(set_to_none=True)
    ref, _ = hip_grads(model, x2.cpu(), names, [torch.ones(s, device="cpu") for s in [tuple(o.shape) for o in out2.values()]], cuda)
    for n in names:
        assert torch.equal(ref[n], g2[n]), n
    # a detached, grad-enabled forward used only as a target also overwrites: the stale graph is refused, not mis-differentiated
    out1 = model(x1)
    with torch.no_grad():
        model(x2)  # inference forward in between: the kept activations are gone
    with pytest.raises(RuntimeError, match="activations"):
        sum(o.sum() for o in out1.values()).backward()


def test_gradients_live_in_one_flat_buffer(lib, cuda):
    """edv_grad_bind_flat: after loss.backward() every p.grad IS its slice of one contiguous buffer the engine wrote (no per-tensor
    copy), the slices follow state-dict order on 16-byte boundaries, and the values equal the per-tensor gradients of the context."""
    model, kwargs, shape, kind, _ = build_model("micro_conv_dvlora")
    x = case_input("micro_conv_dvlora").to(cuda)
    model = model.to(cuda).train()
    endodav_amd.mark_only_part_as_trainable(model, warm_up=True)  # LoRA A/B + the four HeadDepth heads (weights and 1-float biases)
    params = [p for p in model.parameters() if p.requires_grad]
    out = model(x)
    sum((o * o).mean() for o in out.values()).backward()
    flat = model.flat_gradients(params)
    assert flat is not None and flat.dim() == 1
    fg = model._last.flat
    sd = model.state_dict(keep_vars=True)
    assert list(fg.names) == [n for n, p in sd.items() if p.requires_grad]
    end = 0
    for i, n in enumerate(fg.names):
        p = sd[n]
        assert fg.offsets[i] % 4 == 0 and fg.offsets[i] >= end
        end = fg.offsets[i] + p.numel()
        assert p.grad.data_ptr() == flat.data_ptr() + 4 * fg.offsets[i], n   # a view, adopted by autograd without a copy
        assert torch.isfinite(p.grad).all() and p.grad.abs().max() > 0, n
    assert end <= flat.numel() < end + 4
    # the padding between slices is zero, so reducing the whole buffer reduces exactly the gradients
    mask = torch.ones_like(flat, dtype=torch.bool)
    for i, n in enumerate(fg.names):
        mask[fg.offsets[i]:fg.offsets[i] + sd[n].numel()] = False
    assert (flat[mask] == 0).all()
    from endodav_amd import parallel

    before = flat.clone()
    assert parallel.allreduce_gradients(params, model=model) == sum(p.numel() for p in params)  # world 1: in place, nothing to do
    assert torch.equal(flat, before)
    # the C API's per-tensor accessors answer from the flat slices for bound names (they used to look only at the context-owned
    # buffers and returned "no gradient", or a stale one of an earlier unbound backward)
    import ctypes as C

    h = C.c_void_p(model._last.handle)
    for i, n in enumerate(fg.names):
        ptr, numel = C.c_void_p(), C.c_int64()
        assert lib.edv_grad(h, n.encode(), C.byref(ptr), C.byref(numel)) == 0, lib.edv_last_error()
        assert ptr.value == flat.data_ptr() + 4 * fg.offsets[i] and numel.value == sd[n].numel(), n
        got = torch.empty(sd[n].numel(), device=cuda)
        assert lib.edv_grad_copy(h, n.encode(), got.data_ptr(), got.numel(), None) == 0, lib.edv_last_error()
        torch.cuda.synchronize()
        assert torch.equal(got, sd[n].grad.reshape(-1)), n
    # a parameter set that is not the buffer's: the packed fallback is taken (None here)
    assert model.flat_gradients(params[:-1]) is None


def test_gradient_accumulation_with_the_flat_buffer(lib, cuda):
    """Two backward passes without zero_grad: p.grad already IS the flat slice the engine is about to overwrite.  The sums must still be
    g1 + g2 (autograd's accumulation semantics), with set_to_none=False zeroing as well."""
    model, kwargs, shape, kind, _ = build_model("micro_vda_dvlora")
    x1 = case_input("micro_vda_dvlora").to(cuda)
    x2 = (1.0 - x1).contiguous()
    names = set_trainable(model, FACTORS)
    model = model.to(cuda).train()
    sd = model.state_dict(keep_vars=True)

    def run(x):
        sum((o * o).mean() for o in model(x).values()).backward()

    model.zero_grad(set_to_none=True)
    run(x1)
    g1 = {n: sd[n].grad.clone() for n in names}
    model.zero_grad(set_to_none=True)
    run(x2)
    g2 = {n: sd[n].grad.clone() for n in names}
    model.zero_grad(set_to_none=True)
    run(x1)
    run(x2)  # accumulates
    for n in names:
        assert torch.equal(sd[n].grad, g1[n] + g2[n]), n
    model.zero_grad(set_to_none=False)  # zeros written through the views
    run(x2)
    for n in names:
        assert torch.equal(sd[n].grad, g2[n]), n
    # one tensor's .grad replaced by a foreign tensor: autograd adds the view to it; the others keep accumulating in place
    sd[names[0]].grad = torch.ones_like(sd[names[0]])
    run(x1)
    assert torch.equal(sd[names[0]].grad, 1.0 + g1[names[0]])
    for n in names[1:]:
        assert torch.equal(sd[n].grad, g2[n] + g1[n]), n
