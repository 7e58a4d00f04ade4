"""The self-supervised photometric loss of the fine-tune step, on PyTorch-ROCm (BASELINE.json north_star keeps the loss on
the host framework).  A functional restatement of the reference's loss layers, pinned by known-answer
fixtures captured from the imported reference (``tests/golden/loss_kat.npz``, generator ``tests/golden/make_golden.py losses``):

  ssim                       utils/layers.py:276-306   (SSIM: 3x3 mean pools on a reflection-padded pair, clamp((1 - n/d)/2, 0, 1))
  disp_to_depth              utils/layers.py:11-20
  backproject / project      utils/layers.py:134-189   (BackprojectDepth, Project3D)
  reprojection_loss          trainer_end_to_end_video.py:899-911   (0.85 * SSIM + 0.15 * L1, per pixel)
  smooth_loss                utils/layers.py:222-236   (edge-aware first-order smoothness of the mean-normalised disparity)
  photometric_loss           trainer_end_to_end_video.py:808-868, 913-971: per scale, the disparity is resized to the frame size,
                             turned into depth, back-projected, projected into the neighbouring frames (frame_ids -1, +1), the
                             neighbours are sampled there and compared with the frame; plus disparity_smoothness * smooth / 2**scale.

What the reference's trainer adds on top -- pose / optical-flow / appearance networks and their own loss terms
(trainer_end_to_end_video.py:84-126, 741-806, 870-898) -- is outside the hot path: here the relative
poses and intrinsics are inputs.  ``bench.py --train`` uses this loss so that the timed fine-tune step carries the real loss's
shape (grid_sample + SSIM + smoothness on four scales at the frame size) and not a stand-in.
"""
from __future__ import annotations

from typing import Dict, Sequence, Tuple

import torch
import torch.nn.functional as F

MIN_DEPTH, MAX_DEPTH = 0.1, 150.0  # options.py: --min_depth / --max_depth defaults
DISPARITY_SMOOTHNESS = 1e-4        # options.py: --disparity_smoothness default


def disp_to_depth(disp: torch.Tensor, min_depth: float = MIN_DEPTH, max_depth: float = MAX_DEPTH) -> Tuple[torch.Tensor, torch.Tensor]:
    min_disp, max_disp = 1 / max_depth, 1 / min_depth
    scaled = min_disp + (max_disp - min_disp) * disp
    return scaled, 1 / scaled


def ssim(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    x = F.pad(x, (1, 1, 1, 1), mode="reflect")
    y = F.pad(y, (1, 1, 1, 1), mode="reflect")
    mu_x, mu_y = F.avg_pool2d(x, 3, 1), F.avg_pool2d(y, 3, 1)
    sigma_x = F.avg_pool2d(x ** 2, 3, 1) - mu_x ** 2
    sigma_y = F.avg_pool2d(y ** 2, 3, 1) - mu_y ** 2
    sigma_xy = F.avg_pool2d(x * y, 3, 1) - mu_x * mu_y
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    n = (2 * mu_x * mu_y + c1) * (2 * sigma_xy + c2)
    d = (mu_x ** 2 + mu_y ** 2 + c1) * (sigma_x + sigma_y + c2)
    return torch.clamp((1 - n / d) / 2, 0, 1)


def reprojection_loss(pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    l1 = torch.abs(target - pred).mean(1, True)
    return 0.85 * ssim(pred, target).mean(1, True) + 0.15 * l1


def smooth_loss(disp: torch.Tensor, img: torch.Tensor) -> torch.Tensor:
    gdx = torch.abs(disp[:, :, :, :-1] - disp[:, :, :, 1:])
    gdy = torch.abs(disp[:, :, :-1, :] - disp[:, :, 1:, :])
    gix = torch.mean(torch.abs(img[:, :, :, :-1] - img[:, :, :, 1:]), 1, keepdim=True)
    giy = torch.mean(torch.abs(img[:, :, :-1, :] - img[:, :, 1:, :]), 1, keepdim=True)
    return (gdx * torch.exp(-gix)).mean() + (gdy * torch.exp(-giy)).mean()


def pixel_grid(n: int, h: int, w: int, device, dtype=torch.float32) -> torch.Tensor:
    """Homogeneous pixel coordinates [n, 3, h*w] (x, y, 1), x fastest: BackprojectDepth.pix_coords."""
    ys, xs = torch.meshgrid(torch.arange(h, device=device, dtype=dtype), torch.arange(w, device=device, dtype=dtype), indexing="ij")
    pix = torch.stack([xs.reshape(-1), ys.reshape(-1), torch.ones(h * w, device=device, dtype=dtype)], 0)
    return pix.unsqueeze(0).repeat(n, 1, 1)


def backproject(depth: torch.Tensor, inv_K: torch.Tensor, pix: torch.Tensor) -> torch.Tensor:
    n = depth.shape[0]
    cam = torch.matmul(inv_K[:, :3, :3], pix)
    cam = depth.view(n, 1, -1) * cam
    return torch.cat([cam, torch.ones_like(cam[:, :1])], 1)  # [n, 4, h*w]


def project(points: torch.Tensor, K: torch.Tensor, T: torch.Tensor, h: int, w: int, eps: float = 1e-7) -> torch.Tensor:
    n = points.shape[0]
    P = torch.matmul(K, T)[:, :3, :]
    cam = torch.matmul(P, points)
    pix = cam[:, :2, :] / (cam[:, 2, :].unsqueeze(1) + eps)
    pix = pix.view(n, 2, h, w).permute(0, 2, 3, 1)
    scale = torch.tensor([w - 1, h - 1], device=pix.device, dtype=pix.dtype)
    return (pix / scale - 0.5) * 2  # [n, h, w, 2] in grid_sample's [-1, 1]


def photometric_loss(disps: Dict[Tuple[str, int], torch.Tensor], frames: torch.Tensor, K: torch.Tensor, inv_K: torch.Tensor,
                     T_prev: torch.Tensor, T_next: torch.Tensor, scales: Sequence[int] = (0, 1, 2, 3),
                     disparity_smoothness: float = DISPARITY_SMOOTHNESS) -> torch.Tensor:
    """``disps``: {("disp", s): [n, 1, h_s, w_s]} for the n = T frames of one clip; ``frames``: [n, 3, H, W] in [0, 1];
    ``K`` / ``inv_K``: [n, 4, 4]; ``T_prev`` / ``T_next``: [n, 4, 4] poses from frame i to frames i-1 / i+1 (the clip's first /
    last frame has no such neighbour and is left out of that term, as the trainer's frame triplets leave it out)."""
    n, _, H, W = frames.shape
    pix = pixel_grid(n, H, W, frames.device, frames.dtype)
    total = frames.new_zeros(())
    for s in scales:
        disp = disps[("disp", s)]
        if disp.shape[-2:] != (H, W):
            disp = F.interpolate(disp, [H, W], mode="bilinear", align_corners=True)
        _, depth = disp_to_depth(disp)
        cam = backproject(depth, inv_K, pix)
        rep = frames.new_zeros(())
        for T, src, keep in ((T_prev, torch.roll(frames, 1, 0), slice(1, None)), (T_next, torch.roll(frames, -1, 0), slice(None, -1))):
            grid = project(cam, K, T, H, W)
            warped = F.grid_sample(src, grid, padding_mode="border", align_corners=True)
            rep = rep + reprojection_loss(warped[keep], frames[keep]).mean()
        norm = disp / (disp.mean(2, True).mean(3, True) + 1e-7)
        total = total + rep / 2.0 + disparity_smoothness * smooth_loss(norm, frames) / (2 ** s)
    return total / len(scales)


# ---------------------------------------------------------------------------------------------------------------------------------
# The trainer's loss as the trainer computes it: generate_images_pred + compute_losses of
# trainer_end_to_end_video.py:808-971 with the side networks' outputs as INPUTS (pose, optical-flow and appearance networks are outside the hot
# path: SURVEY.md section 2.1 rows 12-15).  `photometric_loss` above is the round-2 subset (neighbours taken from the clip itself, no mask, raw
# frame as target); this is the full per-scale sum:
#
#   loss_reprojection   sum(m * (0.85 SSIM + 0.15 L1)(color_warp(fid), refined(s, fid))) / sum(m)     m = occu_mask_backward(0, fid), detached   (:940-941)
#   loss_transform      sum(m * mean_c |refined(s, fid) - registration(0, fid)|) / sum(m)                                            (:942-943)
#   loss_cvt            get_smooth_bright(transform_high(s, fid), color(0, 0), registration(s, fid), m)       utils/layers.py:239-264  (:944-945)
#   loss_depth_reproj   mean over sampled_depth > 1e-3 of |z of frame i's points in frame i+-1 - depth(i+-1) sampled there|          (:863-877)
#   loss_depth_flow     mean over warp_depth > 1e-3 of |depth(i) carried along the optical flow - depth(i+-1)|                      (:879-890)
#   loss_smooth         get_smooth_loss(disp_s / (mean + 1e-7), color(0, s)) at the resolution of color(0, s)                       (:929-933, :949-951)
#   per scale: rep / 2 + tc * tr / 2 + ts * cvt / 2 + ds * smooth / 2^s + tw * (dr * reproj / 2 + df * flow / 2);  total = mean over the scales (:953-968)
#
# Everything the reference's autograd would reach is differentiable here too: the four disparity maps, refined, transform_high, the poses and --
# with learn_intrinsics (options.py:94-97, default on) -- K and inv_K.  registration and the occlusion mask are detached in the reference.
from dataclasses import dataclass, field


@dataclass
class TrainerLossWeights:
    """options.py defaults; ``tune_temporal`` is the trainer's phase flag (temporal_weight, trainer_end_to_end_video.py:951)."""
    disparity_smoothness: float = 1e-3
    transform_constraint: float = 0.01
    transform_smoothness: float = 0.01
    depth_reproj: float = 0.0
    depth_flow: float = 0.0
    tune_temporal: bool = False
    min_depth: float = MIN_DEPTH
    max_depth: float = MAX_DEPTH


FIDS = (-1, 1)  # opt.frame_ids[1:]


def smooth_bright(transform: torch.Tensor, target: torch.Tensor, pred: torch.Tensor, occu_mask: torch.Tensor) -> torch.Tensor:
    """get_smooth_bright (utils/layers.py:239-264): first-order smoothness of the appearance-flow map, damped where the residue has edges, masked."""
    gtx = torch.mean(torch.abs(transform[:, :, :, :-1] - transform[:, :, :, 1:]), 1, keepdim=True)
    gty = torch.mean(torch.abs(transform[:, :, :-1, :] - transform[:, :, 1:, :]), 1, keepdim=True)
    res = target - pred
    grx = torch.mean(torch.abs(res[:, :, :, :-1] - res[:, :, :, 1:]), 1, keepdim=True)
    gry = torch.mean(torch.abs(res[:, :, :-1, :] - res[:, :, 1:, :]), 1, keepdim=True)
    mx, my = occu_mask[:, :, :, :-1], occu_mask[:, :, :-1, :]
    return (gtx * torch.exp(-grx) * mx).sum() / mx.sum() + (gty * torch.exp(-gry) * my).sum() / my.sum()


def flow_sample(src: torch.Tensor, flow: torch.Tensor, padding: str) -> torch.Tensor:
    """SpatialTransformer.forward (utils/layers.py:387-426): sample ``src`` at (pixel + flow); flow[:, 0] is the row (y) displacement."""
    n, _, H, W = flow.shape
    ys, xs = torch.meshgrid(torch.arange(H, device=flow.device, dtype=flow.dtype), torch.arange(W, device=flow.device, dtype=flow.dtype), indexing="ij")
    ny = 2 * ((ys + flow[:, 0]) / (H - 1) - 0.5)
    nx = 2 * ((xs + flow[:, 1]) / (W - 1) - 0.5)
    return F.grid_sample(src, torch.stack([nx, ny], -1), mode="bilinear", padding_mode=padding, align_corners=True)


def trainer_losses(disps: Dict[Tuple[str, int], torch.Tensor], inp: Dict, weights: TrainerLossWeights = TrainerLossWeights(),
                   scales: Sequence[int] = (0, 1, 2, 3)) -> Dict[str, torch.Tensor]:
    """``inp`` keys (N = B*T flattened frames, as the trainer flattens them, trainer_end_to_end_video.py:406-409):
        ("color", 0, s)            [N, 3, H >> s, W >> s]      frames at the loss scales (s = 0: the frame itself)
        ("color", -1, 0), ("color", 1, 0)                     [N, 3, H, W] the previous / next frame of every frame
        "K", "inv_K"               [N, 4, 4];   ("cam_T_cam", 0, fid) [N, 4, 4]
        ("refined", s, fid), ("registration", s, fid), ("transform", "high", s, fid)   [N, 3, H, W]
        ("occu_mask_backward", 0, fid)   [N, 1, H, W];   ("position", "high", s, fid)   [N, 2, H, W]   (only for depth_flow)
    Returns the trainer's ``losses`` dict (tensors): "loss", "loss/{s}", "loss/loss_reprojection/{s}", ... (:960-966)."""
    w = weights
    frame = inp[("color", 0, 0)]
    n, _, H, W = frame.shape
    pix = pixel_grid(n, H, W, frame.device, frame.dtype)
    tw = 1.0 if w.tune_temporal else 0.0
    out: Dict[str, torch.Tensor] = {}
    total = frame.new_zeros(())
    for s in scales:
        disp_s = disps[("disp", s)]
        d_full = disp_s if disp_s.shape[-2:] == (H, W) else F.interpolate(disp_s, [H, W], mode="bilinear", align_corners=True)
        _, depth = disp_to_depth(d_full, w.min_depth, w.max_depth)
        cam = backproject(depth, inp["inv_K"], pix)
        rep = tr = cvt = drp = dfl = frame.new_zeros(())
        for fid in FIDS:
            T = inp[("cam_T_cam", 0, fid)]
            P = torch.matmul(inp["K"], T)[:, :3, :]
            camp = torch.matmul(P, cam)
            grid = project(cam, inp["K"], T, H, W)
            warped = F.grid_sample(inp[("color", fid, 0)], grid, padding_mode="border", align_corners=True)
            m = inp[("occu_mask_backward", 0, fid)].detach()
            refined = inp[("refined", s, fid)]
            rep = rep + (reprojection_loss(warped, refined) * m).sum() / m.sum()
            tr = tr + (torch.abs(refined - inp[("registration", 0, fid)].detach()).mean(1, True) * m).sum() / m.sum()
            cvt = cvt + smooth_bright(inp[("transform", "high", s, fid)], frame, inp[("registration", s, fid)].detach(), m)
            if tw * w.depth_reproj != 0.0:
                z = camp[:, 2:3, :].reshape(n, 1, H, W)
                if fid == 1:
                    tgt, coords, src = depth[1:], grid[:-1], z[:-1]
                else:
                    tgt, coords, src = depth[:-1], grid[1:], z[1:]
                sampled = F.grid_sample(tgt, coords, padding_mode="zeros", align_corners=True)
                drp = drp + torch.abs(src - sampled)[sampled > 1e-3].mean()
            if tw * w.depth_flow != 0.0:
                flow = inp[("position", "high", s, fid)]
                if fid == 1:
                    origin, fl, fwd = depth[:-1], flow[:-1], depth[1:]
                else:
                    origin, fl, fwd = depth[1:], flow[1:], depth[:-1]
                warp = flow_sample(origin, fl, "zeros")
                dfl = dfl + torch.abs(warp - fwd)[warp > 1e-3].mean()
        color = inp[("color", 0, s)]
        d_c = disp_s if disp_s.shape[-2:] == color.shape[-2:] else F.interpolate(disp_s, list(color.shape[-2:]), mode="bilinear", align_corners=True)
        sm = smooth_loss(d_c / (d_c.mean(2, True).mean(3, True) + 1e-7), color)
        terms = {"loss_reprojection": rep / 2.0, "loss_transform": w.transform_constraint * tr / 2.0, "loss_cvt": w.transform_smoothness * cvt / 2.0,
                 "loss_smooth": w.disparity_smoothness * sm / (2 ** s), "loss_depth_reproj": tw * w.depth_reproj * drp / 2.0,
                 "loss_depth_flow": tw * w.depth_flow * dfl / 2.0}
        loss_s = sum(terms.values())
        total = total + loss_s
        out[f"loss/{s}"] = loss_s
        for k, v in terms.items():
            out[f"loss/{k}/{s}"] = v
    out["loss"] = total / len(scales)
    return out


def synthetic_trainer_inputs(n: int, H: int, W: int, device="cpu", seed: int = 0, dtype=torch.float32) -> Dict:
    """Everything ``trainer_losses`` takes besides the disparity maps, from the portable counter-based generator (so the golden fixtures, the tests
    and bench.py --train see the same numbers on any machine): a tissue-like clip of n + 2 frames (frame i's neighbours are frames i - 1 / i + 1 of
    it, the trainer's dataset layout, datasets/scared_video_dataset.py:289-296), pinhole intrinsics, small relative poses that differ per frame, and
    stand-ins for the side networks' outputs with the statistics the reference's own expressions give them: registration = the neighbour plus a
    small error, transform_high = a low-amplitude field, refined = clamp(transform_high * mask + frame, 0, 1) (trainer :789-790), a ~90 % occlusion
    mask, optical flow of a pixel or two."""
    from . import synth

    clip = torch.from_numpy(synth.synth_clip(1, n + 2, H, W, seed=seed, kind="tissue")[0]).to(dtype)  # [n + 2, 3, H, W]
    u = lambda key, shape, lo, hi: torch.from_numpy(synth.uniform(f"trainer_loss:{seed}:{key}", shape, lo, hi)).to(dtype)
    inp: Dict = {}
    frame = clip[1:-1].contiguous()
    for s in range(4):
        inp[("color", 0, s)] = frame if s == 0 else F.interpolate(frame, [H >> s, W >> s], mode="bilinear", align_corners=False)
    inp[("color", -1, 0)], inp[("color", 1, 0)] = clip[:-2].contiguous(), clip[2:].contiguous()
    K = torch.eye(4, dtype=dtype).repeat(n, 1, 1)
    K[:, 0, 0] = 0.82 * W * (1 + u("fx", (n,), -0.02, 0.02))
    K[:, 1, 1] = 1.02 * H * (1 + u("fy", (n,), -0.02, 0.02))
    K[:, 0, 2], K[:, 1, 2] = 0.5 * W, 0.5 * H
    inp["K"], inp["inv_K"] = K, torch.linalg.inv(K.double()).to(dtype)
    for fid in FIDS:
        T = torch.eye(4, dtype=dtype).repeat(n, 1, 1)
        ang = u(f"ang{fid}", (n,), -0.02, 0.02)
        T[:, 0, 0], T[:, 0, 2], T[:, 2, 0], T[:, 2, 2] = torch.cos(ang), torch.sin(ang), -torch.sin(ang), torch.cos(ang)
        T[:, :3, 3] = u(f"t{fid}", (n, 3), -0.04, 0.04) + torch.tensor([0.05 * fid, 0.0, -0.02 * fid], dtype=dtype)
        inp[("cam_T_cam", 0, fid)] = T
        m = (u(f"mask{fid}", (n, 1, H, W), 0.0, 1.0) > 0.1).to(dtype)
        inp[("occu_mask_backward", 0, fid)] = m
        for s in range(4):
            inp[("registration", s, fid)] = (inp[("color", fid, 0)] + u(f"reg{s}{fid}", (n, 3, H, W), -0.03, 0.03)).clamp(0, 1)
            th = u(f"tr{s}{fid}", (n, 3, H, W), -0.04, 0.04)
            inp[("transform", "high", s, fid)] = th
            inp[("refined", s, fid)] = torch.clamp(th * m + frame, min=0.0, max=1.0)
            inp[("position", "high", s, fid)] = u(f"pos{s}{fid}", (n, 2, H, W), -1.5, 1.5)
    return {k: v.to(device) for k, v in inp.items()}


def synthetic_disps(n: int, sizes: Sequence[Tuple[int, int]], device="cpu", seed: int = 0, dtype=torch.float32) -> Dict[Tuple[str, int], torch.Tensor]:
    """Four smooth positive disparity maps [n, 1, h_s, w_s] in (0.05, 0.9) (sigmoid-like range of the heads)."""
    from . import synth

    out = {}
    for s, (h, w) in enumerate(sizes):
        low = torch.from_numpy(synth.uniform(f"trainer_loss:{seed}:disp{s}", (n, 1, max(h // 6, 2), max(w // 6, 2)), 0.05, 0.9)).to(dtype)
        out[("disp", s)] = F.interpolate(low, [h, w], mode="bilinear", align_corners=True).contiguous().to(device)
    return out


def synthetic_camera(n: int, H: int, W: int, device, dtype=torch.float32):
    """Intrinsics of a pinhole camera normalised like the reference's datasets (fx = 0.82 W, fy = 1.02 H, centre = half size:
    datasets/scared_dataset.py K) and small forward / backward relative poses, for synthetic fine-tune steps."""
    K = torch.eye(4, device=device, dtype=dtype)
    K[0, 0], K[1, 1], K[0, 2], K[1, 2] = 0.82 * W, 1.02 * H, 0.5 * W, 0.5 * H
    Tn = torch.eye(4, device=device, dtype=dtype)
    Tn[0, 3], Tn[2, 3] = 0.05, -0.02
    Tp = torch.linalg.inv(Tn)
    rep = lambda m: m.unsqueeze(0).repeat(n, 1, 1).contiguous()
    return rep(K), rep(torch.linalg.inv(K)), rep(Tp), rep(Tn)


# ---------------------------------------------------------------------------------------------------------------------------------
# The same loss through libendodav_hip (csrc/loss.hip): value and dL/d disp in one fused call -- seven launches per scale instead of
# ~125 eager kernels.  `photometric_loss` above stays the definition the kernels are tested against (tests/test_loss_gpu.py) and the
# path for CPU tensors.
_WS: Dict[Tuple, torch.Tensor] = {}


class _PhotometricLossHip(torch.autograd.Function):
    @staticmethod
    def forward(ctx, frames, K, inv_K, T_prev, T_next, clips, smoothness, d0, d1, d2, d3):
        import ctypes as C

        from . import _lib

        lib = _lib.load()
        disps = [d.detach().contiguous().float() for d in (d0, d1, d2, d3)]
        N, _, H, W = frames.shape
        if N % clips:
            raise ValueError(f"{N} frames do not split into {clips} clips")
        T = N // clips
        dev = frames.device
        mats = [m.detach().contiguous().float() for m in (K, inv_K, T_prev, T_next)]
        for m in mats:
            if tuple(m.shape) != (N, 4, 4):
                raise ValueError(f"camera matrices must be [{N}, 4, 4], got {tuple(m.shape)}")
        fr = frames.detach().contiguous().float()
        with torch.cuda.device(dev):
            nbytes = lib.edv_photometric_loss_workspace(clips, T, H, W)
            if nbytes == 0:
                raise ValueError("the photometric loss needs clips of at least two frames of at least 3x3 pixels")
            key = (str(dev), clips, T, H, W)
            ws = _WS.get(key)
            if ws is None or ws.numel() * 4 < nbytes:
                ws = _WS[key] = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=dev)
            loss = torch.empty((), dtype=torch.float32, device=dev)
            grads = [torch.empty_like(d) for d in disps]
            dptr = (C.c_void_p * 4)(*[d.data_ptr() for d in disps])
            gptr = (C.c_void_p * 4)(*[g.data_ptr() for g in grads])
            dh = (C.c_int32 * 4)(*[d.shape[-2] for d in disps])
            dw = (C.c_int32 * 4)(*[d.shape[-1] for d in disps])
            _lib.check(lib.edv_photometric_loss(fr.data_ptr(), dptr, dh, dw, clips, T, H, W, mats[0].data_ptr(), mats[1].data_ptr(), mats[2].data_ptr(),
                                                mats[3].data_ptr(), MIN_DEPTH, MAX_DEPTH, float(smoothness), loss.data_ptr(), gptr, ws.data_ptr(), ws.numel() * 4,
                                                C.c_void_p(_lib.stream_ptr(dev))), "edv_photometric_loss")
        ctx.grads = grads
        return loss

    @staticmethod
    def backward(ctx, g):
        return (None,) * 7 + tuple(g * gr for gr in ctx.grads)


def photometric_loss_hip(disps: Dict[Tuple[str, int], torch.Tensor], frames: torch.Tensor, K: torch.Tensor, inv_K: torch.Tensor, T_prev: torch.Tensor,
                         T_next: torch.Tensor, clips: int = 1, disparity_smoothness: float = DISPARITY_SMOOTHNESS) -> torch.Tensor:
    """``photometric_loss`` for ``clips`` clips at once (``frames`` [clips*T, 3, H, W], matrices [clips*T, 4, 4]; the mean over the clips)
    on MI355X through ``edv_photometric_loss``.  Differentiable with respect to the four disparity maps."""
    if not frames.is_cuda:
        raise RuntimeError("photometric_loss_hip runs on MI355X only; use photometric_loss for CPU tensors")
    return _PhotometricLossHip.apply(frames, K, inv_K, T_prev, T_next, int(clips), float(disparity_smoothness), *[disps[("disp", s)] for s in range(4)])


# ---------------------------------------------------------------------------------------------------------------------------------
# trainer_losses through libendodav_hip (csrc/loss_trainer.hip): every value of the trainer's `losses` dict and every gradient its autograd reaches
# in one call.  `trainer_losses` above stays the definition (pinned by tests/golden/trainer_loss_kat.npz, captured from the reference's own methods).
_TL_LEAVES = ([("disp", s) for s in range(4)] + ["K", "inv_K"] + [("cam_T_cam", 0, fid) for fid in FIDS] +
              [("refined", s, fid) for s in range(4) for fid in FIDS] + [("transform", "high", s, fid) for s in range(4) for fid in FIDS])


class _TrainerLossHip(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inp, weights, *leaves):
        import ctypes as C

        from . import _lib

        lib = _lib.load()
        t = {k: v.detach().contiguous().float() for k, v in zip(_TL_LEAVES, leaves)}
        frame = inp[("color", 0, 0)]
        N, _, H, W = frame.shape
        dev = frame.device
        keep = []  # contiguous fp32 copies must outlive the launch

        def ptr(x):
            x = x.detach().contiguous().float()
            keep.append(x)
            return x.data_ptr()

        a, g, w = _lib.TrainerLossInputs(), _lib.TrainerLossGrads(), _lib.TrainerLossWeights()
        need = [v.requires_grad for v in leaves]
        grads = {}

        def gbuf(key):
            if not need[_TL_LEAVES.index(key)]:
                return None
            grads[key] = torch.empty_like(t[key])
            return grads[key].data_ptr()

        want_flow = bool(weights.tune_temporal) and weights.depth_flow != 0.0
        for s in range(4):
            a.color[s] = ptr(inp[("color", 0, s)])
            if tuple(inp[("color", 0, s)].shape[-2:]) != (H >> s, W >> s):
                raise ValueError(f'("color", 0, {s}) must be {H >> s} x {W >> s} (height // 2**s, the dataset\'s pyramid)')
            a.disp[s], a.disp_h[s], a.disp_w[s] = t[("disp", s)].data_ptr(), t[("disp", s)].shape[-2], t[("disp", s)].shape[-1]
            grads[("disp", s)] = torch.empty_like(t[("disp", s)])
            g.disp[s] = grads[("disp", s)].data_ptr()
            for n, fid in enumerate(FIDS):
                a.refined[s][n] = t[("refined", s, fid)].data_ptr()
                a.transform[s][n] = t[("transform", "high", s, fid)].data_ptr()
                a.registration[s][n] = ptr(inp[("registration", s, fid)])
                a.position[s][n] = ptr(inp[("position", "high", s, fid)]) if want_flow else None
                g.refined[s][n] = gbuf(("refined", s, fid))
                g.transform[s][n] = gbuf(("transform", "high", s, fid))
        for n, fid in enumerate(FIDS):
            a.color_nb[n] = ptr(inp[("color", fid, 0)])
            a.T[n] = t[("cam_T_cam", 0, fid)].data_ptr()
            a.mask[n] = ptr(inp[("occu_mask_backward", 0, fid)])
            g.T[n] = gbuf(("cam_T_cam", 0, fid))
        a.K, a.invK = t["K"].data_ptr(), t["inv_K"].data_ptr()
        g.K, g.invK = gbuf("K"), gbuf("inv_K")
        w.disparity_smoothness, w.transform_constraint, w.transform_smoothness = weights.disparity_smoothness, weights.transform_constraint, weights.transform_smoothness
        w.depth_reproj, w.depth_flow, w.tune_temporal = weights.depth_reproj, weights.depth_flow, int(bool(weights.tune_temporal))
        w.min_depth, w.max_depth = weights.min_depth, weights.max_depth
        with torch.cuda.device(dev):
            nbytes = lib.edv_trainer_loss_workspace(N, H, W)
            if nbytes == 0:
                raise ValueError("the trainer loss needs frames of at least 16 x 16 pixels")
            key = ("trainer", str(dev), N, H, W)
            ws = _WS.get(key)
            if ws is None or ws.numel() * 4 < nbytes:
                ws = _WS[key] = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=dev)
            values = torch.empty(29, dtype=torch.float32, device=dev)
            _lib.check(lib.edv_trainer_loss(C.byref(a), N, H, W, C.byref(w), values.data_ptr(), C.byref(g), ws.data_ptr(), ws.numel() * 4,
                                            C.c_void_p(_lib.stream_ptr(dev))), "edv_trainer_loss")
        ctx.grads = [grads.get(k) for k in _TL_LEAVES]
        ctx.mark_non_differentiable(values)
        return values[28].clone(), values

    @staticmethod
    def backward(ctx, g, _):
        return (None, None) + tuple(None if gr is None else g * gr for gr in ctx.grads)


def trainer_losses_hip(disps: Dict[Tuple[str, int], torch.Tensor], inp: Dict, weights: TrainerLossWeights = TrainerLossWeights()) -> Dict[str, torch.Tensor]:
    """``trainer_losses`` on MI355X through ``edv_trainer_loss``: the same dict; "loss" is differentiable with respect to the four disparity maps and to
    whichever of K, inv_K, cam_T_cam, refined and transform_high require grad (the reference's autograd graph), the per-scale entries are values
    (the trainer logs them with .item(), trainer_end_to_end_video.py:960-966)."""
    if not inp[("color", 0, 0)].is_cuda:
        raise RuntimeError("trainer_losses_hip runs on MI355X only; use trainer_losses for CPU tensors")
    leaves = [disps[k] if isinstance(k, tuple) and k[0] == "disp" else inp[k] for k in _TL_LEAVES]
    loss, values = _TrainerLossHip.apply(inp, weights, *leaves)
    out = {"loss": loss}
    names = ("", "loss_reprojection/", "loss_transform/", "loss_cvt/", "loss_smooth/", "loss_depth_reproj/", "loss_depth_flow/")
    for s in range(4):
        for k, nm in enumerate(names):
            out[f"loss/{nm}{s}"] = values[s * 7 + k]
    return out
