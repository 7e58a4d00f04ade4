"""Shared helpers of the parity tests."""
from __future__ import annotations

import os
from typing import Dict, Tuple

import numpy as np
import torch

from endodav_amd import synth
from oracle import endodav_oracle as orc
from tests.golden.cases import CASES

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def oracle_config(kwargs, dash_active: bool = False) -> orc.OracleConfig:
    return orc.OracleConfig(
        use_clstoken=kwargs.get("use_clstoken", False), residual_block_indexes=tuple(kwargs.get("residual_block_indexes", ())),
        dash_active=dash_active, use_bn=kwargs.get("use_bn", False), pe=kwargs.get("pe", "ape"),
        encoder=kwargs["encoder"], image_shape=tuple(kwargs["image_shape"]), lora_type=kwargs.get("lora_type", "lora"),
        r=kwargs.get("r", 4), include_cls_token=kwargs.get("include_cls_token", True),
        disable_conv_head=kwargs.get("disable_conv_head", False), inv_sigmoid=kwargs.get("inv_sigmoid", False),
        out_sigmoid=kwargs.get("out_sigmoid", False))


def build_model(name: str):
    """The build's model for golden case ``name`` with the synthetic weights, on CPU."""
    import endodav_amd

    kwargs, shape, kind, store = CASES[name]
    model = endodav_amd.endodav(**kwargs, pretrained_path=None).eval()
    synth.fill_module_(model)
    if name.endswith("_dash_active"):
        from tests.golden.cases import DASH_WARMUP_CALLS

        model._dash_calls = DASH_WARMUP_CALLS  # the next forward is call 101: SVD selection + active term
    return model, kwargs, shape, kind, store


def case_input(name: str) -> torch.Tensor:
    _, (B, T, H, W), kind, _ = CASES[name]
    return torch.from_numpy(synth.synth_clip(B, T, H, W, seed=1, kind=kind))


def load_golden(name: str) -> Dict[str, np.ndarray]:
    with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def frame_stats(a: np.ndarray) -> np.ndarray:
    f = a.reshape(a.shape[0], -1).astype(np.float64)
    return np.stack([f.mean(1), f.min(1), f.max(1), np.sqrt((f * f).sum(1))], axis=1)


def rel_err(a, b) -> float:
    """max |a-b| / max |b| (scale-relative: disparities contain exact zeros after the ReLU)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def depth_rel_err(disp_a, disp_b) -> float:
    """max relative error of depth = 1/(min_disp + (max_disp-min_disp) disp), the quantity BASELINE gates at 1e-3."""
    _, da = orc.disp_to_depth(np.asarray(disp_a, dtype=np.float64))
    _, db = orc.disp_to_depth(np.asarray(disp_b, dtype=np.float64))
    return float((np.abs(da - db) / db).max())


def depth_gate(disp_a, disp_ref, floor_frac: float = 1e-2):
    """(max depth rel. error over pixels with disp_ref >= floor_frac * max(disp_ref), excluded fraction).

    depth = 1/(1/150 + 9.993 disp) has condition number 9.993*depth (~1500 at disp -> 0) with respect to an
    ABSOLUTE disparity error.  Pixels at the zero crossing of the VDA head's trailing ReLU carry an absolute
    fp32 rounding error of ~1e-6*max|disp| whatever their value, so below ~1% of the frame maximum a 1e-3
    relative depth criterion is under the fp32 reproducibility of the reference ITSELF: the pure-torch oracle
    vs the reference golden already shows 2.1e-3 / 5.6e-3 there.  Those pixels are still held by
    the all-pixel gates (disp error <= 5e-5 of scale, abs_rel <= 1e-4)."""
    a = np.asarray(disp_a, dtype=np.float64)
    b = np.asarray(disp_ref, dtype=np.float64)
    _, da = orc.disp_to_depth(a)
    _, db = orc.disp_to_depth(b)
    m = b >= floor_frac * b.max()
    return float((np.abs(da - db) / db)[m].max()), float(1.0 - m.mean())


def abs_rel(disp_a, disp_b) -> float:
    """utils/utils.py:129 abs_rel of build depth against reference depth."""
    _, da = orc.disp_to_depth(np.asarray(disp_a, dtype=np.float64))
    _, db = orc.disp_to_depth(np.asarray(disp_b, dtype=np.float64))
    return float(np.mean(np.abs(db - da) / db))


def nhwc_to_nchw(flat: torch.Tensor, F: int, h: int, w: int, C: int) -> torch.Tensor:
    return flat.reshape(F, h, w, C).permute(0, 3, 1, 2).contiguous()
