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
