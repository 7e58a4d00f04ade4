"""Portable counter-based generator for synthetic weights and clips.

There are no checkpoints in the reference tree (``ckpts/models.zip`` is a git-LFS
pointer), and the reference's default init is degenerate for
parity work (zeroed ``proj_out``, LayerScale 1e-5, LoRA ``B`` = 0,
dead trailing ReLU).  Every parity test, the golden-vector generator and
``bench.py`` therefore overwrite *every* parameter from this generator, keyed by the
state-dict name, so that the reference (in the survey container), the oracle and
the HIP path see bit-identical weights without committing 100 MB+ of tensors.

The generator is splitmix64 over (FNV-1a(name) + index): pure integer arithmetic,
so numpy on any host reproduces it exactly.
"""
from __future__ import annotations

import math
import re
from typing import Dict, Iterable, Mapping, Tuple

import numpy as np

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def fnv1a64(text: str) -> int:
    h = 0xCBF29CE484222325
    for b in text.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _MASK
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
        return z ^ (z >> np.uint64(31))


def uniform01(key: str, n: int, seed: int = 0) -> np.ndarray:
    """n float32 values in [0, 1) for stream ``key`` (24-bit mantissa, exact)."""
    base = np.uint64((fnv1a64(key) ^ (seed * 0xD1342543DE82EF95)) & 0xFFFFFFFFFFFFFFFF)
    idx = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        state = (base + idx * np.uint64(0x9E3779B97F4A7C15)) & _MASK
    z = _splitmix64(state)
    return ((z >> np.uint64(40)).astype(np.float32)) * np.float32(1.0 / (1 << 24))


def uniform(key: str, shape: Iterable[int], lo: float, hi: float, seed: int = 0) -> np.ndarray:
    shape = tuple(int(s) for s in shape)
    n = int(np.prod(shape)) if shape else 1
    u = uniform01(key, n, seed)
    return (np.float32(lo) + u * np.float32(hi - lo)).astype(np.float32).reshape(shape)


# ---------------------------------------------------------------------------
# Per-tensor-class distributions (SURVEY.md §8c "golden-vector recipe").
# ---------------------------------------------------------------------------
_NORM_W = re.compile(r"(^|\.)(norm\d?|norms\.\d+|ff_norm)\.weight$")
_NORM_B = re.compile(r"(^|\.)(norm\d?|norms\.\d+|ff_norm)\.bias$")
_CONVT = re.compile(r"head\.resize_layers\.[01]\.weight$")


def param_range(name: str, shape: Tuple[int, ...]) -> Tuple[float, float] | None:
    """(lo, hi) of the uniform law for state-dict entry ``name``; None = keep as built."""
    if name.endswith("pos_encoder.pe"):
        return None  # deterministic sinusoid buffer (motion_module.py:180-194)
    if name.endswith(".running_var") or re.search(r"\.bn\d\.weight$", name):  # BatchNorm (use_bn=True): positive scales
        return (0.5, 1.5)
    if name.endswith(".gamma"):  # LayerScale: far from the 1e-5 identity trap
        return (0.2, 1.0)
    if _NORM_W.search(name):
        return (0.5, 1.5)
    if _NORM_B.search(name):
        return (-0.1, 0.1)
    if name.endswith("cls_token") or name.endswith("pos_embed") or name.endswith("mask_token"):
        return (-0.1, 0.1)
    if name.endswith("lora_U") or name.endswith("lora_V"):
        return (0.5, 1.5)
    if name.endswith("lora_A") or name.endswith("lora_B"):
        if len(shape) == 2 and shape[1] == 1:  # Linear_SSB row/column scalers
            return (0.5, 1.5)
        fan = shape[1]
        a = math.sqrt(3.0 / fan) * (0.5 if name.endswith("lora_B") else 1.0)
        return (-a, a)
    if name.endswith("lora_index"):
        return (-0.1, 0.1)
    if name.endswith("output_conv2.2.bias"):  # keep the trailing ReLU alive
        return (1.0, 1.5)
    if name.endswith(".bias"):
        return (-0.1, 0.1)
    if len(shape) >= 2:
        if _CONVT.search(name):  # ConvTranspose2d weight is [in, out, k, k]; k == stride
            fan = shape[0]
        else:
            fan = int(np.prod(shape[1:]))
        a = math.sqrt(3.0 / fan)
        return (-a, a)
    return (-0.1, 0.1)


def synth_state(shapes: Mapping[str, Tuple[int, ...]], seed: int = 0) -> Dict[str, np.ndarray]:
    """name -> float32 array for every entry of ``shapes`` that has a law."""
    out: Dict[str, np.ndarray] = {}
    for name, shape in shapes.items():
        rng = param_range(name, tuple(shape))
        if rng is None:
            continue
        out[name] = uniform("w:" + name, shape, rng[0], rng[1], seed)
    return out


def fill_module_(module, seed: int = 0) -> None:
    """Overwrite every parameter/buffer of a torch module in place (reference or build)."""
    import torch

    sd = module.state_dict()
    vals = synth_state({k: tuple(v.shape) for k, v in sd.items() if v.is_floating_point()}, seed)
    with torch.no_grad():
        for k, arr in vals.items():
            sd[k].copy_(torch.from_numpy(arr).to(sd[k].dtype))


def synth_clip(B: int, T: int, H: int, W: int, seed: int = 0, kind: str = "uniform") -> np.ndarray:
    """Synthetic clip ``[B, T, 3, H, W]`` float32 in [0, 1).

    ``uniform``: i.i.d. U[0,1).  ``tissue``: a few low-frequency 2-D
    cosines with a per-frame phase drift, so temporal attention sees correlated frames.
    """
    if kind == "uniform":
        return uniform(f"clip:{B}x{T}x{H}x{W}", (B, T, 3, H, W), 0.0, 1.0, seed)
    if kind != "tissue":
        raise ValueError(kind)
    par = uniform(f"tissue:{B}", (B, 3, 6, 5), 0.0, 1.0, seed)  # amp, fx, fy, phase, drift
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float32) / H, np.arange(W, dtype=np.float32) / W, indexing="ij")
    out = np.zeros((B, T, 3, H, W), np.float32)
    for b in range(B):
        for t in range(T):
            for c in range(3):
                acc = np.zeros((H, W), np.float32)
                for j in range(6):
                    amp, fx, fy, ph, dr = par[b, c, j]
                    acc += amp * np.cos(2 * np.pi * ((1 + 4 * fx) * xx + (1 + 4 * fy) * yy + ph + 0.02 * dr * t))
                out[b, t, c] = acc
    out -= out.min()
    out /= max(float(out.max()), 1e-6) * 1.0001
    return out.astype(np.float32)
