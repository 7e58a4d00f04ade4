#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the imported reference.

CONTAINER-ONLY: needs /root/reference (read-only, never shipped).  Run as

    python tests/golden/make_golden.py [case ...]

What it does, per case of ``cases.CASES``:
  1. imports the reference's ``models.endodav`` (with in-memory stand-ins for the four
     third-party packages absent from this image, see ``_install_standins``);
  2. constructs the reference model, overwrites every parameter from
     ``endodav_amd.synth`` (name-keyed), runs the reference forward on a synthetic clip;
  3. runs ``oracle.endodav_oracle.forward`` on the very same state dict and input and
     REFUSES to write a fixture unless the oracle matches the reference tightly;
  4. stores the reference outputs (whole, or strided samples + per-frame statistics) as
     ``<case>.npz`` (plain float arrays, no pickles).

The stand-ins supply only what the reference imports at module scope and the one
arithmetic op on the forward path, torchvision's ``Normalize`` = (x-mean)/std
(reference models/endodav/endodav.py:88,155).
"""
from __future__ import annotations

import os
import sys
import types
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REFERENCE = "/root/reference"


def _install_standins() -> None:
    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")

    class Compose:
        def __init__(self, ts):
            self.ts = ts

        def __call__(self, x):
            for t in self.ts:
                x = t(x)
            return x

    class Normalize:
        def __init__(self, mean, std):
            self.mean, self.std = mean, std

        def __call__(self, x):
            m = torch.tensor(self.mean, dtype=x.dtype, device=x.device)[:, None, None]
            s = torch.tensor(self.std, dtype=x.dtype, device=x.device)[:, None, None]
            return (x - m) / s

    tvt.Compose, tvt.Normalize = Compose, Normalize
    tv.transforms = tvt
    sys.modules["torchvision"], sys.modules["torchvision.transforms"] = tv, tvt

    cv2 = types.ModuleType("cv2")
    cv2.INTER_CUBIC, cv2.INTER_AREA, cv2.INTER_NEAREST = 2, 3, 0

    def resize(img, size, interpolation=None):  # only the identity case is exercised (video golden)
        assert (img.shape[1], img.shape[0]) == tuple(size), "stand-in cv2.resize: identity only"
        return img

    cv2.resize = resize
    sys.modules["cv2"] = cv2

    ed = types.ModuleType("easydict")

    class EasyDict(dict):
        def __getattr__(self, k):
            return self[k]

    ed.EasyDict = EasyDict
    sys.modules["easydict"] = ed

    fv, fvn, fvw = types.ModuleType("fvcore"), types.ModuleType("fvcore.nn"), types.ModuleType("fvcore.nn.weight_init")

    def c2_msra_fill(m):  # only reached when residual_block_indexes != []; weights are overwritten anyway
        torch.nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    fvw.c2_msra_fill = c2_msra_fill
    fvn.weight_init, fv.nn = fvw, fvn
    sys.modules["fvcore"], sys.modules["fvcore.nn"], sys.modules["fvcore.nn.weight_init"] = fv, fvn, fvw


def load_reference():
    _install_standins()
    sys.path.insert(0, REFERENCE)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        import models.endodav as ref  # noqa
    return ref


def stats(a: np.ndarray) -> np.ndarray:
    """per-frame [mean, min, max, l2] in float64."""
    f = a.reshape(a.shape[0], -1).astype(np.float64)
    return np.stack([f.mean(1), f.min(1), f.max(1), np.sqrt((f * f).sum(1))], axis=1)


def strided(a: np.ndarray, step: int = 7) -> np.ndarray:
    return np.ascontiguousarray(a[..., ::step, ::step])


VIDEO_CASE = dict(n_frames=40, h=28, w=42)


def fake_window_disp(call: int, h: int, w: int):
    """Synthetic per-window disparity [32,1,h,w]: a shared smooth field, window-specific gain/offset and noise,
    so that the least-squares alignment between windows has real work to do."""
    from endodav_amd import synth

    base = synth.uniform("video:base", (32, 1, h, w), 0.5, 2.0)
    noise = synth.uniform(f"video:noise:{call}", (32, 1, h, w), -0.05, 0.05)
    return ((1.0 + 0.3 * call) * base + 0.2 * call + noise).astype(np.float32)


def make_video_golden(ref):
    """Pins windowing, key-frame reuse and stitching of infer_video_depth (reference endodav.py:162-254):
    the reference method runs on CPU with its forward replaced by a recorder that returns synthetic maps."""
    from endodav_amd import synth

    n, h, w = VIDEO_CASE["n_frames"], VIDEO_CASE["h"], VIDEO_CASE["w"]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model = ref.endodav(encoder="vits", features=32, out_channels=[32, 32, 64, 64], image_shape=(h, w), lora_type="none",
                            disable_conv_head=True, pretrained_path=None).eval()
    frames = (synth.uniform("video:frames", (n, h, w, 3), 0.0, 1.0) * 255).astype(np.uint8)
    seen = []

    def recorder(x):
        seen.append(x[0].mean(dim=(1, 2, 3)).numpy().astype(np.float64))  # per-frame mean of the window input
        return {("disp", 0): torch.from_numpy(fake_window_disp(len(seen) - 1, h, w))}

    model.forward = recorder
    out = model.infer_video_depth(frames, device="cpu")
    assert out.shape == (n, h, w)
    path = os.path.join(HERE, "video_stitch.npz")
    np.savez_compressed(path, out=out.astype(np.float32), window_input_means=np.stack(seen))
    print(f"video_stitch: {len(seen)} windows, out mean {out.mean():.4f}; wrote {os.path.getsize(path)} B")


def metrics_inputs():
    """Deterministic inputs of the metric known-answer test (regenerated identically by tests/test_evaluate_cpu.py)."""
    from endodav_amd import synth

    h, w = 40, 56
    gt = synth.uniform("kat:gt", (3, h, w), 5.0, 120.0).astype(np.float32)
    gt[0, :4] = 0.0          # invalid rows (below MIN_DEPTH)
    gt[1, -3:] = 200.0       # above MAX_DEPTH
    pred = (gt * synth.uniform("kat:noise", (3, h, w), 0.8, 1.3) + 1.0).astype(np.float32)
    disp = synth.uniform("kat:disp", (3, h, w), 0.0, 1.0).astype(np.float32)
    K = np.eye(4)
    K[0, 0] = K[1, 1] = 50.0
    K[0, 2], K[1, 2] = w / 2.0, h / 2.0
    pose_a, pose_b = np.eye(4), np.eye(4)
    pose_b[0, 3], pose_b[2, 3] = 0.8, -0.5
    smooth = (30.0 + 10.0 * np.sin(np.arange(w)[None, :] / 9.0) + 5.0 * np.cos(np.arange(h)[:, None] / 7.0)).astype(np.float32)
    return dict(gt=gt, pred=pred, disp=disp, K=K, pose_a=pose_a, pose_b=pose_b, depth_a=smooth, depth_b=(smooth * 1.03 + 0.4).astype(np.float32))


def make_metrics_kat():
    """Known answers of the reference's metric helpers (utils/utils.py:112-133, utils/layers.py:11-20,
    utils/eval_utils.py:63-143,265-282) on the deterministic inputs above."""
    im = types.ModuleType("imageio")
    imv2 = types.ModuleType("imageio.v2")
    im.v2 = imv2
    sys.modules.setdefault("imageio", im)
    sys.modules.setdefault("imageio.v2", imv2)  # import-time only (video writers), never called here
    from utils.utils import compute_errors
    from utils.layers import disp_to_depth
    from utils import eval_utils as eu

    x = metrics_inputs()
    out = {}
    valid = np.logical_and(x["gt"] > 1e-3, x["gt"] < 150)
    out["compute_errors"] = np.array(compute_errors(x["gt"], x["pred"], valid), dtype=np.float64)
    sd, d = disp_to_depth(x["disp"], 0.1, 150.0)
    out["scaled_disp"], out["depth"] = sd, d
    ms, ratio = eu.median_scaling(x["gt"].copy(), x["pred"].copy())
    out["median_scaled"], out["median_ratio"] = ms, np.float64(ratio)
    al, t_gt, s_gt, t_pred, s_pred = eu.align_shift_and_scale(x["gt"].copy(), x["pred"].copy())
    out["aligned"], out["align_params"] = al, np.array([t_gt, s_gt, t_pred, s_pred], dtype=np.float64)
    mask = np.ones_like(x["depth_a"], dtype=bool)
    mask[:3] = False
    i2w_a, i2w_b = np.linalg.inv(x["K"] @ x["pose_a"]), np.linalg.inv(x["K"] @ x["pose_b"])
    out["tae"] = np.float64(eu.tae(x["depth_a"], mask, i2w_a, x["depth_b"], mask, i2w_b))
    out["tas"] = np.float64(eu.tas(x["depth_a"], mask, i2w_a, x["depth_b"], mask, i2w_b))
    path = os.path.join(HERE, "metrics_kat.npz")
    np.savez_compressed(path, **out)
    print(f"metrics_kat: errors {out['compute_errors'][:3]}, tae {out['tae']:.5f}, tas {out['tas']:.4f}; wrote {os.path.getsize(path)} B")


def loss_inputs():
    """Deterministic inputs of the loss known-answer test (regenerated identically by tests/test_losses_cpu.py)."""
    from endodav_amd import synth

    n, H, W = 3, 20, 28
    frames = synth.uniform("losskat:frames", (n, 3, H, W), 0.0, 1.0).astype(np.float32)
    # neighbouring frames of a clip are correlated: frame i = a shifted blend of frame 0 and its own noise
    for i in range(1, n):
        frames[i] = 0.8 * np.roll(frames[0], i, axis=2) + 0.2 * frames[i]
    disps = {s: synth.uniform(f"losskat:disp{s}", (n, 1, H >> s, W >> s), 0.05, 0.9).astype(np.float32) for s in range(4)}
    return n, H, W, frames, disps


def make_loss_kat():
    """Known answers of the reference's loss layers (utils/layers.py:11-20 disp_to_depth, :134-189 BackprojectDepth / Project3D,
    :222-236 get_smooth_loss, :276-306 SSIM) and of their composition in the trainer (trainer_end_to_end_video.py:808-868
    generate_images_pred, :899-911 compute_reprojection_loss, :927-951 the per-scale sum) on the deterministic inputs above, with the
    camera and the relative poses of endodav_amd.losses.synthetic_camera standing in for the pose network's outputs."""
    from utils import layers as L
    from endodav_amd import losses as mine

    n, H, W, frames_np, disps_np = loss_inputs()
    frames = torch.from_numpy(frames_np)
    disps = {s: torch.from_numpy(v) for s, v in disps_np.items()}
    K, inv_K, Tp, Tn = mine.synthetic_camera(n, H, W, "cpu")
    out = {}
    ssim = L.SSIM()
    out["ssim"] = ssim(frames, torch.roll(frames, 1, 0)).numpy()
    out["smooth"] = np.float64(L.get_smooth_loss(disps[0], frames))
    sd, depth = L.disp_to_depth(disps[0], 0.1, 150.0)
    out["depth"] = depth.numpy()
    bp, pj = L.BackprojectDepth(n, H, W), L.Project3D(n, H, W)
    cam = bp(depth, inv_K)
    pix, _ = pj(cam, K, Tn)
    out["cam_points"], out["pix_coords"] = cam.detach().numpy(), pix.detach().numpy()

    def reprojection(pred, target):  # trainer_end_to_end_video.py:899-911 with no_ssim False
        l1 = torch.abs(target - pred).mean(1, True)
        return 0.85 * ssim(pred, target).mean(1, True) + 0.15 * l1

    total = 0.0
    for s in range(4):  # trainer_end_to_end_video.py:808-868 (depth, warp) and :927-951 (loss terms kept by the build: reprojection, smoothness)
        d = torch.nn.functional.interpolate(disps[s], [H, W], mode="bilinear", align_corners=True)
        _, dep = L.disp_to_depth(d, 0.1, 150.0)
        cam = bp(dep, inv_K)
        rep = 0.0
        for T, shift, keep in ((Tp, 1, slice(1, None)), (Tn, -1, slice(None, -1))):
            grid, _ = pj(cam, K, T)
            warped = torch.nn.functional.grid_sample(torch.roll(frames, shift, 0), grid, padding_mode="border", align_corners=True)
            rep = rep + reprojection(warped[keep], frames[keep]).mean()
            if s == 0 and shift == -1:
                out["warped_next"] = warped.numpy()
        mean_disp = d.mean(2, True).mean(3, True)
        total = total + rep / 2.0 + 1e-4 * L.get_smooth_loss(d / (mean_disp + 1e-7), frames) / (2 ** s)
    out["total"] = np.float64(total / 4)
    path = os.path.join(HERE, "loss_kat.npz")
    np.savez_compressed(path, **out)
    print(f"loss_kat: ssim mean {out['ssim'].mean():.5f}, smooth {out['smooth']:.6f}, total {out['total']:.6f}; wrote {os.path.getsize(path)} B")


TRAINER_LOSS_CASE = dict(n=4, H=32, W=48, disp_sizes=[(28, 42), (14, 21), (7, 10), (4, 5)], seed=3,
                         weights=dict(disparity_smoothness=1e-3, transform_constraint=0.01, transform_smoothness=0.01, depth_reproj=1e-2, depth_flow=1e-3,
                                      tune_temporal=True))


def reference_trainer_methods():
    """The reference's OWN generate_images_pred / compute_reprojection_loss / compute_losses (trainer_end_to_end_video.py:808-971), compiled from the
    source where it lies.  The module itself cannot be imported here -- its top pulls in tensorboardX, torchvision.models and the dataset readers
    (trainer :1-20), none of which the loss needs -- so the three methods are taken out of the file's syntax tree and executed in a namespace that
    holds what they use: torch, F and the reference's own utils.layers (imported for real).  Nothing of the file is written anywhere."""
    import ast

    import utils.layers as Lr

    path = os.path.join(REFERENCE, "trainer_end_to_end_video.py")
    with open(path) as f:
        tree = ast.parse(f.read(), filename=path)
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "Trainer")
    want = ("generate_images_pred", "compute_reprojection_loss", "compute_losses")
    mod = ast.Module(body=[n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name in want], type_ignores=[])
    assert len(mod.body) == 3
    ns = {k: getattr(Lr, k) for k in dir(Lr) if not k.startswith("__")}
    ns.update(torch=torch, F=torch.nn.functional)
    exec(compile(mod, path, "exec"), ns)
    return {k: ns[k] for k in want}, Lr


def make_trainer_loss_kat():
    """Known answers of the trainer's whole loss: the reference's own methods on a synthetic inputs / outputs dict; the
    restatement endodav_amd/losses.py::trainer_losses must agree before anything is written (values 1e-6, gradients 2e-5 of scale)."""
    import types as _types

    from endodav_amd import losses as mine

    C = TRAINER_LOSS_CASE
    n, H, W = C["n"], C["H"], C["W"]
    fns, Lr = reference_trainer_methods()
    inp = mine.synthetic_trainer_inputs(n, H, W, seed=C["seed"])
    disps = mine.synthetic_disps(n, C["disp_sizes"], seed=C["seed"])
    wts = mine.TrainerLossWeights(**C["weights"])
    leaves = {}

    def leaf(key, t):
        leaves[key] = t.clone().requires_grad_(True)
        return leaves[key]

    # the reference's dicts: `inputs` from the dataset, `outputs` from the depth model (disp) and the side networks
    inputs = {("color", 0, s): inp[("color", 0, s)] for s in range(4)}
    inputs.update({("color", fid, 0): inp[("color", fid, 0)] for fid in (-1, 1)})
    outputs = {("disp", s): leaf(("disp", s), disps[("disp", s)]) for s in range(4)}
    outputs[("K", 0)], outputs[("inv_K", 0)] = leaf("K", inp["K"]), leaf("inv_K", inp["inv_K"])  # learn_intrinsics (options.py:94-97)
    for fid in (-1, 1):
        outputs[("cam_T_cam", 0, fid)] = leaf(("cam_T_cam", 0, fid), inp[("cam_T_cam", 0, fid)])
        outputs[("occu_mask_backward", 0, fid)] = inp[("occu_mask_backward", 0, fid)]
        for s in range(4):
            outputs[("refined", s, fid)] = leaf(("refined", s, fid), inp[("refined", s, fid)])
            outputs[("transform", "high", s, fid)] = leaf(("transform", "high", s, fid), inp[("transform", "high", s, fid)])
            outputs[("registration", s, fid)] = inp[("registration", s, fid)]
            outputs[("position", "high", s, fid)] = inp[("position", "high", s, fid)]
    opt = _types.SimpleNamespace(scales=[0, 1, 2, 3], v1_multiscale=False, height=H, width=W, min_depth=0.1, max_depth=150.0, learn_intrinsics=True,
                                 frame_ids=[0, -1, 1], pose_model_type="separate_resnet", no_ssim=False, **{k: v for k, v in C["weights"].items() if k != "tune_temporal"})
    me = _types.SimpleNamespace(opt=opt, num_scales=4, tune_temporal=C["weights"]["tune_temporal"], ssim=Lr.SSIM(),
                                backproject_depth={0: Lr.BackprojectDepth(n, H, W)}, project_3d={0: Lr.Project3D(n, H, W)},
                                position_depth={0: Lr.optical_flow((H, W), n, H, W)}, spatial_transform=Lr.SpatialTransformer((H, W)))
    me.compute_reprojection_loss = _types.MethodType(fns["compute_reprojection_loss"], me)
    fns["generate_images_pred"](me, inputs, outputs)
    ref = fns["compute_losses"](me, inputs, outputs)
    ref["loss"].backward()
    # the restatement on the same numbers
    mine_leaves = {k: v.detach().clone().requires_grad_(True) for k, v in leaves.items()}
    inp2 = dict(inp)
    inp2["K"], inp2["inv_K"] = mine_leaves["K"], mine_leaves["inv_K"]
    for k in mine_leaves:
        if isinstance(k, tuple) and k[0] != "disp":
            inp2[k] = mine_leaves[k]
    got = mine.trainer_losses({("disp", s): mine_leaves[("disp", s)] for s in range(4)}, inp2, wts)
    got["loss"].backward()
    out = {}
    worst_v = worst_g = 0.0
    for k, v in ref.items():
        rv = float(v)
        worst_v = max(worst_v, abs(float(got[k]) - rv) / max(abs(rv), 1e-12))
        out["value:" + k] = np.float64(rv)
    for k, t in leaves.items():
        g, gm = t.grad, mine_leaves[k].grad
        assert g is not None and gm is not None, k
        worst_g = max(worst_g, float((g - gm).abs().max() / g.abs().max().clamp_min(1e-30)))
        name = "grad:" + (k if isinstance(k, str) else ":".join(str(x) for x in k))
        out[name] = g.numpy()
    print(f"trainer_loss_kat: restatement vs the reference's compute_losses: values {worst_v:.2e}, gradients {worst_g:.2e} of scale")
    if worst_v > 1e-6 or worst_g > 2e-5:
        raise SystemExit("trainer_losses disagrees with the reference: fixture not written")
    path = os.path.join(HERE, "trainer_loss_kat.npz")
    np.savez_compressed(path, **out)
    print(f"trainer_loss_kat: loss {float(ref['loss']):.6f}, {len(out)} arrays; wrote {os.path.getsize(path)} B")


def dump_state_keys(ref):
    """state_dict key -> shape listings of the reference for the drop-in check."""
    import json

    combos = {
        "vits_dvlora_vda": dict(encoder="vits", features=64, out_channels=[48, 96, 192, 384], lora_type="dvlora", disable_conv_head=True),
        "vits_lora_conv": dict(encoder="vits", features=64, out_channels=[48, 96, 192, 384], lora_type="lora"),
        "vits_ssb_vda_tlora": dict(encoder="vits", features=64, out_channels=[48, 96, 192, 384], lora_type="ssb", disable_conv_head=True, temporal_lora=True),
        "vits_dash_conv": dict(encoder="vits", features=64, out_channels=[48, 96, 192, 384], lora_type="dash"),
        "vits_none_vda": dict(encoder="vits", features=64, out_channels=[48, 96, 192, 384], lora_type="none", disable_conv_head=True),
        "vitl_dvlora_vda": dict(encoder="vitl", features=256, out_channels=[256, 512, 1024, 1024], lora_type="dvlora", disable_conv_head=True),
        "vits_bn_rope": dict(encoder="vits", features=64, out_channels=[48, 96, 192, 384], lora_type="dvlora", disable_conv_head=True, use_bn=True, pe="rope"),
        "vits_clstoken_resblocks": dict(encoder="vits", features=64, out_channels=[48, 96, 192, 384], lora_type="dvlora", use_clstoken=True,
                                        residual_block_indexes=[2, 5, 8, 11]),
    }
    out = {}
    for name, kw in combos.items():
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            m = ref.endodav(**kw, pretrained_path=None)
        out[name] = {"kwargs": kw, "keys": [[k, list(v.shape)] for k, v in m.state_dict().items()],
                     "trainable": sorted(n for n, p in m.named_parameters() if p.requires_grad)}
    path = os.path.join(HERE, "state_keys.json")
    with open(path, "w") as f:
        json.dump(out, f)
    print(f"state_keys: {len(out)} configurations; wrote {os.path.getsize(path)} B")


def build_reference(ref, kwargs):
    """The reference model for one case.  encoder="vitb" raises KeyError in the reference's constructor although vit_base exists
    (SURVEY.md section 0.5): that case is built from the reference's own parts -- its vit_base (vision_transformer.py:368-382)
    handed to the constructor through the 'vits' slot (same tapped blocks [2, 5, 8, 11], endodav.py:76-85); the head takes its
    width from pretrained.embed_dim (endodav.py:94-97), so the module tree and the state-dict keys are what a 'vitb' entry gives."""
    import models.backbones as backbones

    kw = dict(kwargs)
    small = backbones.vits.vit_small
    if kw["encoder"] == "vitb":
        kw["encoder"] = "vits"
        backbones.vits.vit_small = backbones.vits.vit_base
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            model = ref.endodav(**kw, pretrained_path=None).eval()
    finally:
        backbones.vits.vit_small = small
    if kwargs["encoder"] == "vitb":
        assert model.pretrained.embed_dim == 768 and len(model.pretrained.blocks) == 12
    return model


def main(argv):
    from endodav_amd import synth
    from oracle import endodav_oracle as orc
    from tests.golden.cases import CASES, STAGE_KEYS

    ref = load_reference()
    if not argv or "video" in argv:
        make_video_golden(ref)
    if not argv or "keys" in argv:
        dump_state_keys(ref)
    if not argv or "metrics" in argv:
        make_metrics_kat()
    if not argv or "losses" in argv:
        make_loss_kat()
    if not argv or "trainer_losses" in argv:
        make_trainer_loss_kat()
    names = [a for a in argv if a not in ("video", "keys", "metrics", "losses", "trainer_losses")] if argv else list(CASES)
    torch.set_num_threads(8)
    for name in names:
        kwargs, (B, T, H, W), kind, store = CASES[name]
        model = build_reference(ref, kwargs)
        synth.fill_module_(model)
        x = torch.from_numpy(synth.synth_clip(B, T, H, W, seed=1, kind=kind))
        dash_active = name.endswith("_dash_active")
        with torch.no_grad():
            if dash_active:  # DashLinear switches its SVD term on at call 101 (mylora/layers.py:558-583)
                from tests.golden.cases import DASH_WARMUP_CALLS
                for _ in range(DASH_WARMUP_CALLS):
                    model(x[:, :1])
            out_ref = model(x)
        sd = {k: v.detach() for k, v in model.state_dict().items()}
        cfg = orc.OracleConfig(
            encoder=kwargs["encoder"], image_shape=tuple(kwargs["image_shape"]), lora_type=kwargs.get("lora_type", "lora"),
            r=kwargs.get("r", 4), include_cls_token=kwargs.get("include_cls_token", True),
            disable_conv_head=kwargs.get("disable_conv_head", False), inv_sigmoid=kwargs.get("inv_sigmoid", False),
            out_sigmoid=kwargs.get("out_sigmoid", False), use_clstoken=kwargs.get("use_clstoken", False),
            residual_block_indexes=tuple(kwargs.get("residual_block_indexes", ())), dash_active=dash_active,
            use_bn=kwargs.get("use_bn", False), pe=kwargs.get("pe", "ape"))
        stages = {}
        with torch.no_grad():
            out_orc = orc.forward(sd, x, cfg, stages)
        worst = 0.0
        for k in out_ref:
            a, b = out_ref[k], out_orc[k]
            assert a.shape == b.shape, (name, k, a.shape, b.shape)
            err = float((a - b).abs().max() / a.abs().max().clamp_min(1e-12))
            worst = max(worst, err)
        assert worst < 2e-5, f"{name}: oracle deviates from the reference by {worst:.3e}"
        nz = float((out_ref[('disp', 0)] > 0).float().mean())
        payload = {"oracle_vs_reference_maxrel": np.float64(worst)}
        for k, v in out_ref.items():
            a = v.numpy()
            payload[f"disp{k[1]}_stats"] = stats(a)
            payload[f"disp{k[1]}"] = a if store == "full" else strided(a)
        for sk in STAGE_KEYS:  # per-stage pins come from the (reference-validated) oracle
            if sk in stages:
                a = stages[sk].numpy()
                payload[f"stage_{sk}_stats"] = stats(a)
                if store == "strided":
                    flat = a.reshape(a.shape[0], -1)
                    payload[f"stage_{sk}_sample"] = np.ascontiguousarray(flat[:, ::997][:, :256])
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **payload)
        print(f"{name}: oracle-vs-reference max rel {worst:.2e}; disp0 nonzero frac {nz:.3f}; "
              f"disp0 mean {float(out_ref[('disp', 0)].mean()):.4f}; wrote {os.path.getsize(path)} B")


if __name__ == "__main__":
    main(sys.argv[1:])
