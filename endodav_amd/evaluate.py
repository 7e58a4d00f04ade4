"""Video-depth evaluation around ``infer_video_depth``: the counterpart of the reference's
``evaluate_depth_video.py:50-253``.

Host-side numpy only — as in the reference — and shaped so that ``evaluate_video(model, dataset, ...)`` can be fed by
the reference's own ``SCAREDVideos`` loader (items are dicts with ``colors, depths, poses, Ks, filename``) or by
``SyntheticVideos`` below (the SCARED/Hamlyn frames are not in the build container).

Metric definitions restated from ``utils/utils.py:112-133`` (compute_errors), ``utils/layers.py:11-20`` (disp_to_depth),
``utils/eval_utils.py:63-143`` (reprojection, TAE, TAS) and ``:265-282`` (median / shift-scale alignment); pinned by
known-answer values captured from those functions (``tests/golden/metrics_kat.npz``).
"""
from __future__ import annotations

import time
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np

METRIC_NAMES = ("abs_rel", "sq_rel", "rmse", "rmse_log", "a1", "a2", "a3")
TEMPORAL_NAMES = ("tae", "tas")


def disp_to_depth(disp, min_depth: float, max_depth: float):
    """Sigmoid-style disparity in [0,1] -> (scaled disparity, depth) with depth in [min_depth, max_depth]."""
    lo, hi = 1.0 / max_depth, 1.0 / min_depth
    scaled = lo + (hi - lo) * disp
    return scaled, 1.0 / scaled


def compute_errors(gt: np.ndarray, pred: np.ndarray, mask: Optional[np.ndarray] = None) -> Tuple[float, ...]:
    """(abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3) over the masked pixels."""
    if mask is not None:
        gt, pred = gt[mask], pred[mask]
    ratio = np.maximum(gt / pred, pred / gt)
    acc = [(ratio < 1.25 ** k).mean() for k in (1, 2, 3)]
    diff = gt - pred
    rmse = np.sqrt((diff ** 2).mean())
    rmse_log = np.sqrt(((np.log(gt) - np.log(pred)) ** 2).mean())
    abs_rel = np.mean(np.abs(diff) / gt)
    sq_rel = np.mean(diff ** 2 / gt)
    return abs_rel, sq_rel, rmse, rmse_log, acc[0], acc[1], acc[2]


def median_scaling(gt: np.ndarray, pred: np.ndarray, min_depth: float = 1e-3, max_depth: float = 150.0):
    """Scale the whole prediction so the medians over the valid GT pixels agree.  Returns (pred * ratio, ratio)."""
    valid = (gt > min_depth) & (gt < max_depth)
    ratio = np.median(gt[valid]) / np.median(pred[valid])
    pred = pred * ratio
    return pred, ratio


def align_shift_and_scale(gt: np.ndarray, pred: np.ndarray, min_depth: float = 1e-3, max_depth: float = 150.0):
    """Robust affine alignment: match median (shift) and mean absolute deviation (scale) over the valid pixels."""
    valid = (gt > min_depth) & (gt < max_depth)
    g, p = gt[valid], pred[valid]
    t_gt, t_pred = np.median(g), np.median(p)
    s_gt, s_pred = np.mean(np.abs(g - t_gt)), np.mean(np.abs(p - t_pred))
    return (pred - t_pred) * (s_gt / s_pred) + t_gt, t_gt, s_gt, t_pred, s_pred


# ---- temporal consistency (TAE / TAS) ------------------------------------------------------------------------------
def _lift(depth: np.ndarray, mask: np.ndarray, img2world: np.ndarray) -> np.ndarray:
    """Masked pixels -> 3-D points: pixel centres (x+.5, y+.5) scaled by depth, then the 4x4 image->world map."""
    h, w = depth.shape
    ys, xs = np.meshgrid(np.linspace(0.5, h - 0.5, h), np.linspace(0.5, w - 0.5, w), indexing="ij")
    pts = np.stack([xs, ys, depth, np.ones_like(xs)], axis=-1)[mask]
    pts[..., :2] *= pts[..., 2:3]
    return (pts @ img2world.T)[..., :3]


def _splat(points: np.ndarray, mask: np.ndarray, img2world: np.ndarray) -> np.ndarray:
    """3-D points -> depth image of the other frame (nearest-pixel splat, later points overwrite earlier ones)."""
    pts = np.concatenate([points, np.ones_like(points[..., :1])], axis=-1) @ np.linalg.inv(img2world).T
    z = pts[..., 2]
    eps = 1e-6
    ok = z > eps
    uv = np.round(pts[..., :2] / np.clip(pts[..., 2:3], a_min=eps, a_max=None)).astype(np.int32)
    h, w = mask.shape
    ok &= (uv[..., 0] >= 0) & (uv[..., 0] < w) & (uv[..., 1] >= 0) & (uv[..., 1] < h)
    out = np.zeros((h, w), dtype=np.float32)
    out[uv[ok][..., 1], uv[ok][..., 0]] = z[ok]
    return out * mask


def _pairwise(metric, depth_a, mask_a, i2w_a, depth_b, mask_b, i2w_b) -> float:
    a2b = _splat(_lift(depth_a, mask_a, i2w_a), mask_b, i2w_b)
    m = (a2b > 1e-6) & mask_b
    e1 = metric(depth_b[m], a2b[m])
    b2a = _splat(_lift(depth_b, mask_b, i2w_b), mask_a, i2w_a)
    m = (b2a > 1e-6) & mask_a
    e2 = metric(depth_a[m], b2a[m])
    return 0.5 * (e1 + e2)


def tae(depth_a, mask_a, i2w_a, depth_b, mask_b, i2w_b) -> float:
    """Temporal alignment error: symmetric abs_rel between a frame's depth and its neighbour's reprojected depth."""
    return _pairwise(lambda gt, pred: (np.abs(gt - pred) / gt).mean(), depth_a, mask_a, i2w_a, depth_b, mask_b, i2w_b)


def tas(depth_a, mask_a, i2w_a, depth_b, mask_b, i2w_b) -> float:
    """Temporal alignment score: the same with the delta < 1.25 accuracy."""
    return _pairwise(lambda gt, pred: (np.maximum(gt / pred, pred / gt) < 1.25).mean(), depth_a, mask_a, i2w_a, depth_b, mask_b, i2w_b)


# ---- harness ---------------------------------------------------------------------------------------------------------
def _evaluate_clip(depther, item: dict, min_depth, max_depth, depth_align, pred_depth_scale_factor, eval_max_depth, device) -> Dict[str, object]:
    """One clip of evaluate_depth_video.py:163-215: ``infer_video_depth`` -> depth -> alignment -> per-frame errors and
    frame-to-frame TAE (x100) / TAS."""
    MIN_DEPTH = 1e-3
    colors, gts, poses, Ks = item["colors"], item["depths"], item["poses"], item["Ks"]
    rec: Dict[str, object] = {"errors": [], "temporal": [], "ratio": None, "align": None}
    t0 = time.time()
    disp = depther.infer_video_depth(colors, device=device) if device is not None else depther.infer_video_depth(colors)
    rec["time"] = time.time() - t0
    _, pred = disp_to_depth(disp, min_depth, max_depth)
    if depth_align == "scale":
        pred, ratio = median_scaling(gts, pred)
        if not np.isnan(ratio).all():
            rec["ratio"] = float(ratio)
    elif depth_align == "scale_shift":
        pred, *abcd = align_shift_and_scale(gts, pred)
        rec["align"] = tuple(float(v) for v in abcd)
    prev = None
    for p, g, pose, K in zip(pred, gts, poses, Ks):
        valid = (g > MIN_DEPTH) & (g < eval_max_depth)
        p = np.clip(p * pred_depth_scale_factor, MIN_DEPTH, eval_max_depth)
        err = compute_errors(g, p, valid)
        if not np.isnan(err).all():
            rec["errors"].append(tuple(float(e) for e in err))
        i2w = np.linalg.inv(K @ pose)
        if prev is not None:
            rec["temporal"].append([tae(prev[0], prev[1], prev[2], p, valid, i2w) * 100.0, tas(prev[0], prev[1], prev[2], p, valid, i2w)])
        prev = (p, valid, i2w)
    return rec


def evaluate_video(depther, dataset: Iterable[dict], *, min_depth: float = 0.1, max_depth: float = 150.0, depth_align: str = "scale",
                   pred_depth_scale_factor: float = 1.0, eval_max_depth: float = 150.0, device: str = "cuda",
                   rank: Optional[int] = None, world: Optional[int] = None) -> Optional[Dict[str, object]]:
    """The loop of evaluate_depth_video.py:163-215.  ``depther`` only needs ``infer_video_depth(colors)``.

    Clips are independent units: with ``world`` > 1 (default: the ``torch.distributed`` process group, if any) rank r
    evaluates clips r, r + world, ... on its own GPU -- indexed directly when the dataset supports ``len`` / ``[]`` (the reference's
    SCAREDVideos does), skipped over otherwise -- and rank 0 gathers the per-clip records and concatenates them in clip order, so
    the result equals the one-rank run's bit for bit.  The other ranks return None.  No collective on the data path."""
    from . import parallel

    if rank is None or world is None:
        rank, world = parallel.rank_world()
    args = (min_depth, max_depth, depth_align, pred_depth_scale_factor, eval_max_depth, device)
    indexable = hasattr(dataset, "__getitem__") and hasattr(dataset, "__len__")
    if indexable:
        n_clips = len(dataset)
        mine = [_evaluate_clip(depther, dataset[i], *args) for i in parallel.clip_shard(n_clips, rank, world)]
    else:
        mine, n_clips = [], 0
        for i, item in enumerate(dataset):
            n_clips += 1
            if i % world == rank:
                mine.append(_evaluate_clip(depther, item, *args))
    if world == 1:
        recs = mine
    else:
        shards = parallel.gather_to_rank0(mine)
        if shards is None:
            return None
        recs = parallel.merge_shards(shards, n_clips)
    errors = [e for r in recs for e in r["errors"]]
    temporal = [t for r in recs for t in r["temporal"]]
    return {"errors": np.array(errors), "temporal": np.array(temporal), "inference_times": np.array([r["time"] for r in recs]),
            "ratios": np.array([r["ratio"] for r in recs if r["ratio"] is not None]),
            "aligns": np.array([r["align"] for r in recs if r["align"] is not None])}


def _mean_ci(a: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """Column means and their two-sided 95 % Student-t confidence bounds (evaluate_depth_video.py:230-244)."""
    import scipy.stats as st

    mean = a.mean(axis=0)
    ci = []
    for j in range(a.shape[1]):
        lo, hi = st.t.interval(0.95, df=len(a) - 1, loc=mean[j], scale=st.sem(a[:, j]))
        ci += [lo, hi]
    return mean, np.array(ci)


def format_results(res: Dict[str, object]) -> str:
    """The results.txt text of evaluate_depth_video.py:245-250 (same columns, same format strings)."""
    mean, ci = _mean_ci(res["errors"])
    tmean, tci = _mean_ci(res["temporal"])
    txt = ("{:>11}      | " * 9).format(*METRIC_NAMES, *TEMPORAL_NAMES)
    txt += "\nmean:" + ("&{: 12.3f}      " * 9).format(*mean.tolist(), *tmean.tolist()) + "\\\\"
    txt += "\ncls: " + ("& [{: 6.3f}, {: 6.3f}] " * 9).format(*ci.tolist(), *tci.tolist()) + "\\\\"
    txt += "\naverage inference time: {:0.1f} ms".format(float(np.mean(res["inference_times"])) * 1000)
    return txt


class SyntheticVideos:
    """Stand-in for datasets.SCAREDVideos (datasets/scared_video_dataset.py:77): ``n_clips`` videos of a smooth,
    slowly drifting depth field seen by a camera translating along x, with matching intrinsics and poses."""

    def __init__(self, n_clips: int = 2, n_frames: int = 24, height: int = 70, width: int = 98, seed: int = 0):
        self.n_clips, self.n_frames, self.h, self.w, self.seed = n_clips, n_frames, height, width, seed

    def __len__(self):
        return self.n_clips

    def __iter__(self):
        for c in range(self.n_clips):
            yield self[c]

    def __getitem__(self, c: int) -> dict:
        from . import synth

        if not (0 <= c < self.n_clips):
            raise IndexError(c)
        clip = synth.synth_clip(1, self.n_frames, self.h, self.w, seed=self.seed + c, kind="tissue")[0]  # [N,3,H,W]
        colors = (clip.transpose(0, 2, 3, 1) * 255).astype(np.uint8)
        depths = (20.0 + 60.0 * clip.mean(axis=1)).astype(np.float32)  # 20 .. 80 units, correlated with the image
        K = np.eye(4, dtype=np.float64)
        K[0, 0] = K[1, 1] = 0.9 * self.w
        K[0, 2], K[1, 2] = self.w / 2.0, self.h / 2.0
        poses = np.stack([np.eye(4) for _ in range(self.n_frames)])
        poses[:, 0, 3] = 0.05 * np.arange(self.n_frames)
        return {"colors": colors, "depths": depths, "poses": poses, "Ks": np.stack([K] * self.n_frames), "filename": f"synthetic/clip{c}/0"}
