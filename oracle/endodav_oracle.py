"""ORACLE — test infrastructure only.  Never imported by the product path.

A CPU (PyTorch fp32) restatement of the reference's per-clip forward
``models.endodav.endodav.forward`` (reference ``models/endodav/endodav.py:150-160``),
written functionally over a plain ``state_dict`` so that it shares no module code with
either the reference or the HIP build.  Only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import this file.

Pinning: the reference ships no tests, fixtures or golden vectors for this path
(SURVEY.md §0.3, §4).  The oracle is pinned instead against the reference itself,
imported in the build container by ``tests/golden/make_golden.py`` (which also stores
small golden fixtures under ``tests/golden/``); ``tests/test_oracle_golden.py`` replays
those fixtures on every run.

Every function cites the reference lines it restates (paths relative to the
reference root).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Mapping, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
State = Mapping[str, Tensor]

# encoder -> (embed_dim, depth, heads, tapped blocks).  vits/vitl: endodav.py:76-85 +
# vision_transformer.py:352-398; vitb is the build's extension using
# vision_transformer.py:368-382 and the sibling endodac tap list.
ENCODERS = {
    "vits": (384, 12, 6, (2, 5, 8, 11)),
    "vitb": (768, 12, 12, (2, 5, 8, 11)),
    "vitl": (1024, 24, 16, (4, 11, 17, 23)),
}
IMAGENET_MEAN = (0.485, 0.456, 0.406)  # endodav.py:88
IMAGENET_STD = (0.229, 0.224, 0.225)
PATCH = 14


@dataclass
class OracleConfig:
    encoder: str = "vits"
    image_shape: Tuple[int, int] = (224, 280)
    lora_type: str = "lora"
    r: int = 4
    include_cls_token: bool = True
    use_clstoken: bool = False
    disable_conv_head: bool = False
    inv_sigmoid: bool = False
    out_sigmoid: bool = False
    residual_block_indexes: Sequence[int] = field(default_factory=tuple)
    temporal_heads: int = 8  # dpt_temporal.py:35
    dash_active: bool = False  # DashLinear past its 100-call warm-up (mylora/layers.py:572-583)
    use_bn: bool = False  # eval-mode BatchNorm2d after both convs of every ResidualConvUnit (util/blocks.py:60-62,80-86)
    pe: str = "ape"  # "ape": sinusoid added to the attention input; "rope": rotary q/k (motion_module.py:214-225,252-255)
    # test-only override so tiny encoders can be exercised: (embed_dim, depth, heads, taps)
    encoder_dims: Optional[Tuple[int, int, int, Tuple[int, ...]]] = None

    def dims(self):
        return self.encoder_dims if self.encoder_dims is not None else ENCODERS[self.encoder]


# ---------------------------------------------------------------------------
# LoRA linears — models/backbones/mylora/layers.py
# ---------------------------------------------------------------------------
def lora_linear(sd: State, prefix: str, x: Tensor, cfg: OracleConfig) -> Tensor:
    """``mlp.fc1/fc2`` (and ``ff.net.2`` under temporal_lora) after endodav.py:102-137.

    lora   : y = xWᵀ+b + 2·(x Aᵀ Bᵀ)                    layers.py:148-157, alpha=2r  (endodav.py:111)
    dvlora : y = xWᵀ+b + 1·(x (A⊙U)ᵀ (B⊙V)ᵀ)            layers.py:384-393, alpha=r   (endodav.py:108)
    ssb    : y = x (a ⊙ W ⊙ b)ᵀ + bias                  layers.py:423-430
    dash   : lora with alpha=2r, plus x (U_top diag(idx) Vt_top)ᵀ once warmed up   layers.py:553-585
    """
    W = sd[prefix + ".weight"]
    b = sd.get(prefix + ".bias")
    if prefix + ".lora_A" not in sd or cfg.lora_type == "none":
        return F.linear(x, W, b)
    A, B = sd[prefix + ".lora_A"], sd[prefix + ".lora_B"]
    if cfg.lora_type == "ssb":
        return F.linear(x, A.view(1, -1) * W * B, b)
    y = F.linear(x, W, b)
    if cfg.lora_type == "dvlora":
        U, V = sd[prefix + ".lora_U"], sd[prefix + ".lora_V"]
        return y + (x @ (A * U).T @ (B * V).T) * 1.0
    if cfg.lora_type == "lora":
        return y + (x @ A.T @ B.T) * 2.0
    if cfg.lora_type == "dash":
        y = y + (x @ A.T @ B.T) * 2.0
        if cfg.dash_active:
            top = sd[prefix + ".weight_u_top"] @ torch.diag(sd[prefix + ".lora_index"]) @ sd[prefix + ".weight_vt_top"]
            y = y + x @ top.T
        return y
    raise ValueError(cfg.lora_type)


# ---------------------------------------------------------------------------
# Encoder — models/backbones/vision_transformer.py + layers/
# ---------------------------------------------------------------------------
def pos_embed_for(sd: State, n_tokens: int, H: int, W: int, include_cls: bool) -> Tensor:
    """vision_transformer.py:186-217 (bicubic resample of the patch grid, +0.1 offset).

    The reference names its height ``w`` and width ``h`` (``B, nc, w, h = x.shape``,
    :220); the first scale factor therefore applies to image rows.
    """
    pe = sd["pretrained.pos_embed"]
    npatch = n_tokens - 1
    N = pe.shape[1] - 1
    if npatch == N and H == W:
        return pe
    dim = pe.shape[-1]
    s = int(math.sqrt(N))
    h0, w0 = H // PATCH + 0.1, W // PATCH + 0.1
    grid = pe[:, 1:].float().reshape(1, s, s, dim).permute(0, 3, 1, 2)
    grid = F.interpolate(grid, scale_factor=(float(h0) / math.sqrt(N), float(w0) / math.sqrt(N)), mode="bicubic", antialias=False)
    assert grid.shape[-2] == int(h0) and grid.shape[-1] == int(w0)
    grid = grid.permute(0, 2, 3, 1).reshape(1, -1, dim)
    if include_cls:
        return torch.cat((pe[:, :1], grid), dim=1)
    return grid


def patch_tokens(sd: State, x: Tensor, cfg: OracleConfig) -> Tensor:
    """patch_embed.py:68-81 + vision_transformer.py:219-239."""
    BT, _, H, W = x.shape
    assert H % PATCH == 0 and W % PATCH == 0
    t = F.conv2d(x, sd["pretrained.patch_embed.proj.weight"], sd["pretrained.patch_embed.proj.bias"], stride=PATCH)
    t = t.flatten(2).transpose(1, 2)
    if cfg.include_cls_token:
        t = torch.cat((sd["pretrained.cls_token"].expand(BT, -1, -1), t), dim=1)
    return t + pos_embed_for(sd, t.shape[1], H, W, cfg.include_cls_token)


def vit_attention(sd: State, p: str, x: Tensor, heads: int) -> Tensor:
    """layers/attention.py:56-69 — q is scaled *before* QKᵀ."""
    B, N, C = x.shape
    d = C // heads
    qkv = F.linear(x, sd[p + ".qkv.weight"], sd[p + ".qkv.bias"]).reshape(B, N, 3, heads, d).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * d ** -0.5, qkv[1], qkv[2]
    a = (q @ k.transpose(-2, -1)).softmax(dim=-1)
    o = (a @ v).transpose(1, 2).reshape(B, N, C)
    return F.linear(o, sd[p + ".proj.weight"], sd[p + ".proj.bias"])


def channels_first_ln(x: Tensor, w: Tensor, b: Tensor, eps: float = 1e-6) -> Tensor:
    """layers/utils.py:155-179 (channels_first branch)."""
    u = x.mean(1, keepdim=True)
    s = (x - u).pow(2).mean(1, keepdim=True)
    return w[:, None, None] * ((x - u) / torch.sqrt(s + eps)) + b[:, None, None]


def res_bottleneck(sd: State, p: str, x: Tensor) -> Tensor:
    """layers/utils.py:90-153: 1×1 → LN → GELU → 3×3 → LN → GELU → 1×1 → LN (no conv bias)."""
    y = F.conv2d(x, sd[p + ".conv1.weight"])
    y = F.gelu(channels_first_ln(y, sd[p + ".norm1.weight"], sd[p + ".norm1.bias"]))
    y = F.conv2d(y, sd[p + ".conv2.weight"], padding=1)
    y = F.gelu(channels_first_ln(y, sd[p + ".norm2.weight"], sd[p + ".norm2.bias"]))
    y = F.conv2d(y, sd[p + ".conv3.weight"])
    return channels_first_ln(y, sd[p + ".norm3.weight"], sd[p + ".norm3.bias"])


def vit_block(sd: State, i: int, x: Tensor, cfg: OracleConfig, ph: int, pw: int) -> Tensor:
    """layers/block.py:110-151, eval branch (:144-150)."""
    D, _, heads, _ = cfg.dims()
    p = f"pretrained.blocks.{i}"
    h = F.layer_norm(x, (D,), sd[p + ".norm1.weight"], sd[p + ".norm1.bias"], 1e-6)
    x = x + sd[p + ".ls1.gamma"] * vit_attention(sd, p + ".attn", h, heads)
    h = F.layer_norm(x, (D,), sd[p + ".norm2.weight"], sd[p + ".norm2.bias"], 1e-6)
    h = lora_linear(sd, p + ".mlp.fc1", h, cfg)
    h = F.gelu(h)  # nn.GELU() default = exact erf (block.py:58)
    h = lora_linear(sd, p + ".mlp.fc2", h, cfg)
    x = x + sd[p + ".ls2.gamma"] * h
    if i in cfg.residual_block_indexes:
        c = 1 if cfg.include_cls_token else 0
        B, N, C = x.shape
        grid = x[:, c:, :].reshape(B, ph, pw, C).permute(0, 3, 1, 2)
        upd = res_bottleneck(sd, p + ".residual_", grid).permute(0, 2, 3, 1).reshape(B, N - c, C)
        x = torch.cat((x[:, :c], x[:, c:] + upd), dim=1)
    return x


def encoder_taps(sd: State, x: Tensor, cfg: OracleConfig, stages: Optional[dict] = None) -> List[Tuple[Tensor, Tensor]]:
    """vision_transformer.py:279-289,305-333 with norm=True, return_class_token=True."""
    D, depth, _, taps = cfg.dims()
    ph, pw = x.shape[-2] // PATCH, x.shape[-1] // PATCH
    t = patch_tokens(sd, x, cfg)
    if stages is not None:
        stages["tokens"] = t
    outs = []
    for i in range(depth):
        t = vit_block(sd, i, t, cfg, ph, pw)
        if stages is not None and (i == 0 or i == depth - 1):
            stages[f"block{i}"] = t
        if i in taps:
            n = F.layer_norm(t, (D,), sd["pretrained.norm.weight"], sd["pretrained.norm.bias"], 1e-6)
            if cfg.include_cls_token:
                outs.append((n[:, 1:], n[:, 0]))
            else:  # "not real cls tokens" (vision_transformer.py:322-324)
                outs.append((n, n[:, 0]))
    if stages is not None:
        for j, (o, _) in enumerate(outs):
            stages[f"tap{j}"] = o
    return outs


# ---------------------------------------------------------------------------
# Motion module — models/endodav/motion_module/{motion_module,attention}.py
# ---------------------------------------------------------------------------
def rope_table(C: int, T: int, theta: float = 10000.0) -> Tuple[Tensor, Tensor]:
    """cos, sin [T, C/2] of attention.py:402-408 (precompute_freqs_cis over the full channel width, before the head split)."""
    freqs = 1.0 / (theta ** (torch.arange(0, C, 2)[: C // 2].float() / C))
    ang = torch.outer(torch.arange(T, dtype=torch.float32), freqs)
    return torch.cos(ang), torch.sin(ang)


def rope_rotate(t: Tensor, cos: Tensor, sin: Tensor) -> Tensor:
    """attention.py:419-429: channel pairs (2i, 2i+1) of row f rotated by f·freq_i.  ``t``: [N, T, C]."""
    a, b = t[..., 0::2], t[..., 1::2]
    return torch.stack((a * cos - b * sin, a * sin + b * cos), dim=-1).flatten(-2)


def temporal_attention(sd: State, p: str, xn: Tensor, T: int, heads: int, pe: str = "ape") -> Tensor:
    """motion_module.py:230-297 + attention.py:182-211.

    ``xn``: [(b·T), P, C] already layer-normed.  Returns the attention block output
    (to_out applied), same shape; the caller adds the residual.
    """
    BT, P, C = xn.shape
    Bc = BT // T
    h = xn.reshape(Bc, T, P, C).permute(0, 2, 1, 3).reshape(Bc * P, T, C)  # "(b f) d c -> (b d) f c"
    if pe == "ape":
        h = h + sd[p + ".pos_encoder.pe"][:, :T]  # PE enters q, k *and* v (motion_module.py:234-250)
    q = F.linear(h, sd[p + ".to_q.weight"])
    k = F.linear(h, sd[p + ".to_k.weight"])
    v = F.linear(h, sd[p + ".to_v.weight"])
    if pe == "rope":  # motion_module.py:252-255
        cos, sin = rope_table(C, T)
        q, k = rope_rotate(q, cos, sin), rope_rotate(k, cos, sin)
    d = C // heads

    def split(t):
        return t.reshape(Bc * P, T, heads, d).permute(0, 2, 1, 3)

    q, k, v = split(q), split(k), split(v)
    a = ((q @ k.transpose(-1, -2)) * d ** -0.5).softmax(dim=-1)  # baddbmm(alpha=scale)
    o = (a @ v).permute(0, 2, 1, 3).reshape(Bc * P, T, C)
    o = F.linear(o, sd[p + ".to_out.0.weight"], sd[p + ".to_out.0.bias"])
    return o.reshape(Bc, P, T, C).permute(0, 2, 1, 3).reshape(BT, P, C)


def motion_module(sd: State, m: int, x: Tensor, T: int, cfg: OracleConfig) -> Tensor:
    """motion_module.py:102-126 and :164-177.  ``x``: [(b·T), C, h, w] → same shape."""
    p = f"head.motion_modules.{m}.temporal_transformer"
    BT, C, hh, ww = x.shape
    # the reference hands GroupNorm a contiguous NCHW copy (rearrange at motion_module.py:105);
    # torch's channels-last GroupNorm kernel is measurably less accurate on thin groups
    x = x.contiguous()
    g = F.group_norm(x, 32, sd[p + ".norm.weight"], sd[p + ".norm.bias"], 1e-6)
    t = g.permute(0, 2, 3, 1).reshape(BT, hh * ww, C)
    t = F.linear(t, sd[p + ".proj_in.weight"], sd[p + ".proj_in.bias"])
    b = p + ".transformer_blocks.0"
    for a in range(2):  # num_attention_blocks = 2 (dpt_temporal.py:37)
        n = F.layer_norm(t, (C,), sd[f"{b}.norms.{a}.weight"], sd[f"{b}.norms.{a}.bias"], 1e-5)
        t = temporal_attention(sd, f"{b}.attention_blocks.{a}", n, T, cfg.temporal_heads, cfg.pe) + t
    n = F.layer_norm(t, (C,), sd[b + ".ff_norm.weight"], sd[b + ".ff_norm.bias"], 1e-5)
    val, gate = F.linear(n, sd[b + ".ff.net.0.proj.weight"], sd[b + ".ff.net.0.proj.bias"]).chunk(2, dim=-1)
    ff = lora_linear(sd, b + ".ff.net.2", val * F.gelu(gate), cfg)  # GEGLU, attention.py:363-384
    t = ff + t
    t = F.linear(t, sd[p + ".proj_out.weight"], sd[p + ".proj_out.bias"])
    return (t.reshape(BT, hh, ww, C).permute(0, 3, 1, 2) + x).contiguous()


# ---------------------------------------------------------------------------
# DPT head — models/endodav/{dpt,dpt_temporal,dpt_pyramid}.py, util/blocks.py, layers.py
# ---------------------------------------------------------------------------
def _up(x: Tensor, size=None, scale=None) -> Tensor:
    return F.interpolate(x, size=size, scale_factor=scale, mode="bilinear", align_corners=True)


def residual_conv_unit(sd: State, p: str, x: Tensor, bn: bool = False) -> Tensor:
    """util/blocks.py:68-91: x + [bn2](conv2(relu([bn1](conv1(relu(x)))))); BatchNorm in eval mode (running statistics)."""
    def norm(y, q):
        if not bn:
            return y
        return F.batch_norm(y, sd[q + ".running_mean"], sd[q + ".running_var"], sd[q + ".weight"], sd[q + ".bias"], False, 0.0, 1e-5)

    y = norm(F.conv2d(F.relu(x), sd[p + ".conv1.weight"], sd[p + ".conv1.bias"], padding=1), p + ".bn1")
    y = norm(F.conv2d(F.relu(y), sd[p + ".conv2.weight"], sd[p + ".conv2.bias"], padding=1), p + ".bn2")
    return y + x


def fusion_block(sd: State, p: str, x: Tensor, skip: Optional[Tensor], size, bn: bool = False) -> Tensor:
    """util/blocks.py:135-162."""
    if skip is not None:
        x = x + residual_conv_unit(sd, p + ".resConfUnit1", skip, bn)
    x = residual_conv_unit(sd, p + ".resConfUnit2", x, bn)
    x = _up(x, size=size) if size is not None else _up(x, scale=2)
    return F.conv2d(x, sd[p + ".out_conv.weight"], sd[p + ".out_conv.bias"])


def head_depth(sd: State, p: str, x: Tensor) -> Tensor:
    """layers.py:206-221 (HeadDepth): 3×3 → ×2 bilinear → 3×3 → ReLU → 1×1."""
    y = F.conv2d(x, sd[p + ".head.0.weight"], sd[p + ".head.0.bias"], padding=1)
    y = _up(y, scale=2)
    y = F.relu(F.conv2d(y, sd[p + ".head.2.weight"], sd[p + ".head.2.bias"], padding=1))
    return F.conv2d(y, sd[p + ".head.4.weight"], sd[p + ".head.4.bias"])


def dpt_head(sd: State, feats, ph: int, pw: int, T: int, cfg: OracleConfig, stages: Optional[dict] = None) -> Dict[Tuple[str, int], Tensor]:
    """dpt_pyramid.py:51-113."""
    levels = []
    for i, (tok, cls) in enumerate(feats):
        if cfg.use_clstoken:  # dpt_pyramid.py:54-57
            ro = cls.unsqueeze(1).expand_as(tok)
            tok = F.gelu(F.linear(torch.cat((tok, ro), -1), sd[f"head.readout_projects.{i}.0.weight"], sd[f"head.readout_projects.{i}.0.bias"]))
        g = tok.permute(0, 2, 1).reshape(tok.shape[0], tok.shape[-1], ph, pw)
        g = F.conv2d(g, sd[f"head.projects.{i}.weight"], sd[f"head.projects.{i}.bias"])
        if i == 0:
            g = F.conv_transpose2d(g, sd["head.resize_layers.0.weight"], sd["head.resize_layers.0.bias"], stride=4)
        elif i == 1:
            g = F.conv_transpose2d(g, sd["head.resize_layers.1.weight"], sd["head.resize_layers.1.bias"], stride=2)
        elif i == 3:
            g = F.conv2d(g, sd["head.resize_layers.3.weight"], sd["head.resize_layers.3.bias"], stride=2, padding=1)
        levels.append(g)
    l1, l2, l3, l4 = levels
    l3 = motion_module(sd, 0, l3, T, cfg)
    l4 = motion_module(sd, 1, l4, T, cfg)
    if stages is not None:
        stages["mm0"], stages["mm1"] = l3, l4
    r1 = F.conv2d(l1, sd["head.scratch.layer1_rn.weight"], padding=1)
    r2 = F.conv2d(l2, sd["head.scratch.layer2_rn.weight"], padding=1)
    r3 = F.conv2d(l3, sd["head.scratch.layer3_rn.weight"], padding=1)
    r4 = F.conv2d(l4, sd["head.scratch.layer4_rn.weight"], padding=1)
    s = "head.scratch."
    p4 = fusion_block(sd, s + "refinenet4", r4, None, r3.shape[2:], cfg.use_bn)
    p4 = motion_module(sd, 2, p4, T, cfg)
    p3 = fusion_block(sd, s + "refinenet3", p4, r3, r2.shape[2:], cfg.use_bn)
    p3 = motion_module(sd, 3, p3, T, cfg)
    p2 = fusion_block(sd, s + "refinenet2", p3, r2, r1.shape[2:], cfg.use_bn)
    p1 = fusion_block(sd, s + "refinenet1", p2, r1, None, cfg.use_bn)
    if stages is not None:
        stages.update(path4=p4, path3=p3, path2=p2, path1=p1)
    out: Dict[Tuple[str, int], Tensor] = {}
    if cfg.disable_conv_head:  # VDA-style head, dpt.py:117-124 + dpt_pyramid.py:88-102
        o = F.conv2d(p1, sd[s + "output_conv1.weight"], sd[s + "output_conv1.bias"], padding=1)
        o = _up(o, size=(ph * PATCH, pw * PATCH))
        o = F.relu(F.conv2d(o, sd[s + "output_conv2.0.weight"], sd[s + "output_conv2.0.bias"], padding=1))
        o = F.relu(F.conv2d(o, sd[s + "output_conv2.2.weight"], sd[s + "output_conv2.2.bias"]))
        out[("disp", 0)] = o
        for k in (1, 2, 3):
            out[("disp", k)] = _up(out[("disp", k - 1)], scale=0.5)
        if cfg.out_sigmoid:
            out = {k: torch.sigmoid(v) for k, v in out.items()}
    else:  # four HeadDepth heads, dpt_pyramid.py:103-109
        sign = -1.0 if cfg.inv_sigmoid else 1.0
        for k, path in ((3, p4), (2, p3), (1, p2), (0, p1)):
            out[("disp", k)] = torch.sigmoid(sign * head_depth(sd, f"head.conv_depth_{k + 1}", path))
    return out


# ---------------------------------------------------------------------------
# Top level — models/endodav/endodav.py:150-160
# ---------------------------------------------------------------------------
def forward(sd: State, x: Tensor, cfg: OracleConfig, stages: Optional[dict] = None) -> Dict[Tuple[str, int], Tensor]:
    """``x``: [B, T, 3, H, W] in [0, 1] → {("disp", s): [B·T, 1, h_s, w_s]}."""
    B, T = x.shape[:2]
    xr = F.interpolate(x.flatten(0, 1), size=tuple(cfg.image_shape), mode="bilinear", align_corners=True)
    mean = torch.tensor(IMAGENET_MEAN, dtype=xr.dtype)[None, :, None, None]
    std = torch.tensor(IMAGENET_STD, dtype=xr.dtype)[None, :, None, None]
    xn = (xr - mean) / std
    ph, pw = xn.shape[-2] // PATCH, xn.shape[-1] // PATCH
    feats = encoder_taps(sd, xn, cfg, stages)
    return dpt_head(sd, feats, ph, pw, T, cfg, stages)


def disp_to_depth(disp, min_depth: float = 0.1, max_depth: float = 150.0):
    """utils/layers.py:11-20."""
    min_disp, max_disp = 1.0 / max_depth, 1.0 / min_depth
    scaled = min_disp + (max_disp - min_disp) * disp
    return scaled, 1.0 / scaled
