"""Host-side mirror of the reference model class ``models.endodav.endodav``
(reference ``models/endodav/endodav.py:53-254``), running the forward on MI355X through
``libendodav_hip.so``.

What this module owns (PyTorch = plumbing): the parameter tree with the reference's exact
``state_dict`` key names and shapes, device placement, streams, output
allocation.  What it does NOT own: arithmetic.  ``forward`` hands device pointers to
``edv_forward``; there is no PyTorch/CPU fallback — on a CPU tensor, or with the shared library
missing, it raises.

Constructor arguments, attribute names reached by the reference's callers
(``.pretrained.blocks[i].mlp.fc1``, ``.head.motion_modules``, ``.image_shape``) and error
behaviour (``KeyError`` for an unknown encoder, ``AssertionError`` for ``num_frames <= 0``,
``RuntimeError`` for T > num_frames, ``FileNotFoundError`` for a missing ``pretrained_path``)
follow SURVEY.md §8b.  ``vitb`` is an extension.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .trainable import mark_only_part_as_trainable

# infer settings of the reference (endodav.py:47-50)
INFER_LEN = 32
OVERLAP = 10
KEYFRAMES = [6, 12, 24, 25, 26, 27, 28, 29, 30, 31]
INTERP_LEN = 8

# encoder -> (embed_dim, depth, heads, img_size of the stored pos_embed, tapped blocks)
# vision_transformer.py:352-398 (vit_large keeps DinoVisionTransformer's img_size=224 default, hence
# a 257-row pos_embed: SURVEY.md §7 "drop-in quirks"); taps endodav.py:76-79; vitb from
# models/endodac/endodac.py:184-199.
PRODUCTS = {"f32": 0, "bf16x6": 1}  # EDV_PRODUCTS_* of include/endodav_hip.h
PRODUCTS_DEFAULT = "bf16x6"  # round 3: inference default (training forwards always run "f32")

ENCODERS = {
    "vits": (384, 12, 6, 518, (2, 5, 8, 11)),
    "vitb": (768, 12, 12, 518, (2, 5, 8, 11)),
    "vitl": (1024, 24, 16, 224, (4, 11, 17, 23)),
}
PATCH = 14


# ---------------------------------------------------------------------------------------------
# Parameter holders.  Same attribute names / shapes / default init as the reference modules.
# None of them is ever *called* by the product path.
# ---------------------------------------------------------------------------------------------
class _Holder(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover - guard against accidental eager use
        raise RuntimeError(f"{type(self).__name__} only holds parameters; the forward runs in libendodav_hip")


class LoRALinear(nn.Linear):
    """mylora/layers.py:90-157 (r > 0, merge_weights False): weight/bias + lora_A [r,in], lora_B [out,r]."""

    def __init__(self, in_features, out_features, r=0, lora_alpha=1):
        super().__init__(in_features, out_features)
        self.r, self.lora_alpha = r, lora_alpha
        self.scaling = lora_alpha / r
        self.lora_A = nn.Parameter(self.weight.new_zeros((r, in_features)))
        self.lora_B = nn.Parameter(self.weight.new_zeros((out_features, r)))
        self.weight.requires_grad = False
        nn.init.kaiming_uniform_(self.lora_A, a=math.sqrt(5))


class DVLinear(LoRALinear):
    """mylora/layers.py:328-393: adds lora_U [r,1] and lora_V [out,1]."""

    def __init__(self, in_features, out_features, r=0, lora_alpha=1):
        super().__init__(in_features, out_features, r, lora_alpha)
        self.lora_U = nn.Parameter(self.weight.new_zeros(r, 1))
        self.lora_V = nn.Parameter(self.weight.new_zeros(out_features, 1))
        nn.init.kaiming_uniform_(self.lora_U, a=math.sqrt(5))
        nn.init.kaiming_uniform_(self.lora_V, a=math.sqrt(5))


class SSBLinear(nn.Linear):
    """mylora/layers.py:396-430 (Linear_SSB): lora_A [in,1] and lora_B [out,1], initialised to one."""

    def __init__(self, in_features, out_features, r=0):
        super().__init__(in_features, out_features)
        self.r = r
        self.lora_A = nn.Parameter(torch.ones(in_features, 1))
        self.lora_B = nn.Parameter(torch.ones(out_features, 1))
        self.weight.requires_grad = False


class DashLinear(LoRALinear):
    """mylora/layers.py:487-585: LoRA + an 8-direction SVD term switched on after 100 calls."""

    WARMUP = 100

    def __init__(self, in_features, out_features, r=0, lora_alpha=1):
        super().__init__(in_features, out_features, r, lora_alpha)
        self.index = 8
        self.lora_index = nn.Parameter(self.weight.new_zeros(self.index))
        self.weight_u_top = nn.Parameter(self.weight.new_zeros(out_features, self.index))
        self.weight_vt_top = nn.Parameter(self.weight.new_zeros(self.index, in_features))
        self.warmup = 100
        self.FLAG = 0


def _make_lora(kind: str, fin: int, fout: int, r: int) -> nn.Module:
    if kind == "dvlora":
        return DVLinear(fin, fout, r=r, lora_alpha=r)  # endodav.py:108-109
    if kind == "lora":
        return LoRALinear(fin, fout, r=r, lora_alpha=2 * r)  # endodav.py:111-112
    if kind == "ssb":
        return SSBLinear(fin, fout, r=r)  # endodav.py:114-115
    if kind == "dash":
        return DashLinear(fin, fout, r=r, lora_alpha=2 * r)  # endodav.py:117-118
    raise ValueError(f"unknown lora_type {kind!r}")


class _Gamma(_Holder):  # LayerScale, layer_scale.py:16-27
    def __init__(self, dim, init_values=1e-5):
        super().__init__()
        self.gamma = nn.Parameter(init_values * torch.ones(dim))


class _Attn(_Holder):  # layers/attention.py:36-54
    def __init__(self, dim, heads):
        super().__init__()
        self.num_heads = heads
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.proj = nn.Linear(dim, dim, bias=True)


class _Mlp(_Holder):  # layers/mlp.py:16-31
    def __init__(self, dim):
        super().__init__()
        self.fc1 = nn.Linear(dim, 4 * dim)
        self.fc2 = nn.Linear(4 * dim, dim)


class _CFLayerNorm(_Holder):  # channels-first LayerNorm of layers/utils.py:155-179 (weight, bias only)
    def __init__(self, n):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(n))
        self.bias = nn.Parameter(torch.zeros(n))
        self.eps = 1e-6


class _ResBottleneck(_Holder):  # layers/utils.py:90-153: 1x1 -> LN -> GELU -> 3x3 -> LN -> GELU -> 1x1 -> LN
    def __init__(self, dim):
        super().__init__()
        b = dim // 8
        self.conv1 = nn.Conv2d(dim, b, 1, bias=False)
        self.norm1 = _CFLayerNorm(b)
        self.conv2 = nn.Conv2d(b, b, 3, padding=1, bias=False)
        self.norm2 = _CFLayerNorm(b)
        self.conv3 = nn.Conv2d(b, dim, 1, bias=False)
        self.norm3 = _CFLayerNorm(dim)
        for conv in (self.conv1, self.conv2, self.conv3):  # fvcore c2_msra_fill
            nn.init.kaiming_normal_(conv.weight, mode="fan_out", nonlinearity="relu")
        nn.init.zeros_(self.norm3.weight)  # "zero init last norm layer" (:149-151)


class _Block(_Holder):  # layers/block.py:42-109
    def __init__(self, dim, heads, use_residual_block=False):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = _Attn(dim, heads)
        self.ls1 = _Gamma(dim)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = _Mlp(dim)
        self.ls2 = _Gamma(dim)
        self.use_residual_block = use_residual_block
        if use_residual_block:
            self.residual_ = _ResBottleneck(dim)


class _PatchEmbed(_Holder):  # layers/patch_embed.py:38-66
    def __init__(self, dim):
        super().__init__()
        self.proj = nn.Conv2d(3, dim, kernel_size=PATCH, stride=PATCH)


class _Backbone(_Holder):
    """Parameter tree of DinoVisionTransformer (vision_transformer.py:43-184)."""

    def __init__(self, dim, depth, heads, img_size, residual_block_indexes=()):
        super().__init__()
        self.embed_dim = self.num_features = dim
        self.n_blocks, self.num_heads, self.patch_size = depth, heads, PATCH
        n_patches = (img_size // PATCH) ** 2
        self.patch_embed = _PatchEmbed(dim)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, n_patches + 1, dim))
        self.blocks = nn.ModuleList([_Block(dim, heads, i in residual_block_indexes) for i in range(depth)])
        self.norm = nn.LayerNorm(dim, eps=1e-6)
        self.mask_token = nn.Parameter(torch.zeros(1, dim))
        nn.init.trunc_normal_(self.pos_embed, std=0.02)  # vision_transformer.py:180-184
        nn.init.normal_(self.cls_token, std=1e-6)
        for m in self.modules():
            if isinstance(m, nn.Linear):  # init_weights_vit_timm, :342-348
                nn.init.trunc_normal_(m.weight, std=0.02)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)


class _PosEnc(_Holder):  # motion_module.py:180-198
    def __init__(self, d_model, max_len):
        super().__init__()
        position = torch.arange(max_len).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, d_model, 2) * (-math.log(10000.0) / d_model))
        pe = torch.zeros(1, max_len, d_model)
        pe[0, :, 0::2] = torch.sin(position * div_term)
        pe[0, :, 1::2] = torch.cos(position * div_term)
        self.register_buffer("pe", pe)


def _rope_table(dim, max_len, theta=10000.0):
    """(cos, sin) of attention.py:402-408 (precompute_freqs_cis) as a real tensor [max_len, dim/2, 2]."""
    freqs = 1.0 / (theta ** (torch.arange(0, dim, 2)[: (dim // 2)].float() / dim))
    ang = torch.outer(torch.arange(max_len, dtype=torch.float32), freqs)
    return torch.view_as_real(torch.polar(torch.ones_like(ang), ang)).contiguous()


class _TemporalAttention(_Holder):  # motion_module.py:200-228 + attention.py:44-91
    def __init__(self, dim, max_len, pe="ape"):
        super().__init__()
        self.to_q = nn.Linear(dim, dim, bias=False)
        self.to_k = nn.Linear(dim, dim, bias=False)
        self.to_v = nn.Linear(dim, dim, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(dim, dim), nn.Dropout(0.0)])
        if pe == "ape":  # motion_module.py:214-219
            self.pos_encoder = _PosEnc(dim, max_len)
        elif pe == "rope":  # :221-225 — a plain attribute in the reference: not part of the state_dict
            self.pos_encoder = None
            self.register_buffer("freqs_cis", _rope_table(dim, max_len), persistent=False)
        else:
            raise NotImplementedError  # motion_module.py:227-228


class _GEGLU(_Holder):  # attention.py:363-384
    def __init__(self, dim, inner):
        super().__init__()
        self.proj = nn.Linear(dim, inner * 2)


class _FeedForward(_Holder):  # attention.py:296-338
    def __init__(self, dim):
        super().__init__()
        self.net = nn.ModuleList([_GEGLU(dim, dim * 4), nn.Dropout(0.0), nn.Linear(dim * 4, dim)])


class _TemporalBlock(_Holder):  # motion_module.py:129-177
    def __init__(self, dim, max_len, pe="ape"):
        super().__init__()
        self.attention_blocks = nn.ModuleList([_TemporalAttention(dim, max_len, pe) for _ in range(2)])
        self.norms = nn.ModuleList([nn.LayerNorm(dim) for _ in range(2)])
        self.ff = _FeedForward(dim)
        self.ff_norm = nn.LayerNorm(dim)


class _TemporalTransformer(_Holder):  # motion_module.py:68-126
    def __init__(self, channels, max_len, pe="ape"):
        super().__init__()
        self.norm = nn.GroupNorm(num_groups=32, num_channels=channels, eps=1e-6, affine=True)
        self.proj_in = nn.Linear(channels, channels)
        self.transformer_blocks = nn.ModuleList([_TemporalBlock(channels, max_len, pe)])
        self.proj_out = nn.Linear(channels, channels)


class TemporalModule(_Holder):  # motion_module.py:32-65, zero_initialize=True
    def __init__(self, in_channels, max_len, pe="ape"):
        super().__init__()
        self.temporal_transformer = _TemporalTransformer(in_channels, max_len, pe)
        for p in self.temporal_transformer.proj_out.parameters():
            p.detach().zero_()


class _RCU(_Holder):  # util/blocks.py:37-66
    def __init__(self, f, bn=False):
        super().__init__()
        self.conv1 = nn.Conv2d(f, f, 3, 1, 1, bias=True)
        self.conv2 = nn.Conv2d(f, f, 3, 1, 1, bias=True)
        if bn:  # :60-62
            self.bn1 = nn.BatchNorm2d(f)
            self.bn2 = nn.BatchNorm2d(f)


class _Fusion(_Holder):  # util/blocks.py:94-133
    def __init__(self, f, bn=False):
        super().__init__()
        self.out_conv = nn.Conv2d(f, f, 1, 1, 0, bias=True)
        self.resConfUnit1 = _RCU(f, bn)
        self.resConfUnit2 = _RCU(f, bn)


class _Interp(nn.Module):  # endodav/layers.py:194-204; no parameters, keeps Sequential indices 0,2,4
    pass


class HeadDepth(_Holder):  # endodav/layers.py:206-221
    def __init__(self, f):
        super().__init__()
        self.head = nn.Sequential(nn.Conv2d(f, f // 2, 3, 1, 1), _Interp(), nn.Conv2d(f // 2, 32, 3, 1, 1), nn.ReLU(), nn.Conv2d(32, 1, 1, 1, 0))


class _Head(_Holder):
    """Parameter tree of DPTHeadPyramid (dpt.py:47-124, dpt_temporal.py:22-51, dpt_pyramid.py:22-49)."""

    def __init__(self, in_channels, features, out_channels, num_frames, disable_conv_head, use_clstoken=False, use_bn=False, pe="ape"):
        super().__init__()
        oc = list(out_channels)
        self.use_clstoken = use_clstoken
        self.projects = nn.ModuleList([nn.Conv2d(in_channels, c, 1, 1, 0) for c in oc])
        self.resize_layers = nn.ModuleList([
            nn.ConvTranspose2d(oc[0], oc[0], 4, 4, 0),
            nn.ConvTranspose2d(oc[1], oc[1], 2, 2, 0),
            nn.Identity(),
            nn.Conv2d(oc[3], oc[3], 3, 2, 1),
        ])
        if use_clstoken:  # dpt.py:92-98
            self.readout_projects = nn.ModuleList([nn.Sequential(nn.Linear(2 * in_channels, in_channels), nn.GELU()) for _ in oc])
        s = nn.Module()
        for j in range(4):
            setattr(s, f"layer{j + 1}_rn", nn.Conv2d(oc[j], features, 3, 1, 1, bias=False))
        s.stem_transpose = None
        for j in (1, 2, 3, 4):
            setattr(s, f"refinenet{j}", _Fusion(features, use_bn))
        if disable_conv_head:
            s.output_conv1 = nn.Conv2d(features, features // 2, 3, 1, 1)
            s.output_conv2 = nn.Sequential(nn.Conv2d(features // 2, 32, 3, 1, 1), nn.ReLU(True), nn.Conv2d(32, 1, 1, 1, 0), nn.ReLU(True), nn.Identity())
        self.scratch = s
        assert num_frames > 0  # dpt_temporal.py:34
        self.motion_modules = nn.ModuleList([
            TemporalModule(oc[2], num_frames, pe), TemporalModule(oc[3], num_frames, pe),
            TemporalModule(features, num_frames, pe), TemporalModule(features, num_frames, pe),
        ])
        self.disable_conv_head = disable_conv_head
        if not disable_conv_head:
            for k in (1, 2, 3, 4):
                setattr(self, f"conv_depth_{k}", HeadDepth(features))


def _is_lora_factor(name: str) -> bool:
    """Factors of the linears edv_refresh_lora re-folds: mlp.fc1/fc2 of the encoder blocks, ff.net.2 of the motion modules."""
    return (".mlp.fc" in name or ".ff.net.2." in name) and name.rsplit(".", 1)[-1] in ("lora_A", "lora_B", "lora_U", "lora_V", "lora_index")


def _is_head_conv(name: str) -> bool:
    """Weights / biases of the output-head convolutions the HIP backward differentiates: the HeadDepth heads (trainable by default,
    endodav/layers.py:5-34) and, for the VDA head, scratch.output_conv* (--train_output_conv)."""
    return (name.startswith("head.conv_depth_") or name.startswith("head.scratch.output_conv")) and name.rsplit(".", 1)[-1] in ("weight", "bias")


class _NativeCtx:
    """Owns one ``edv_ctx`` (one device).  Destroyed with the last module/replica that references it."""

    def __init__(self, handle: int, device: torch.device):
        self.handle, self.device, self.sig = handle, device, None
        self.flat: Optional["_FlatGrads"] = None

    def __del__(self):
        try:
            _lib.load().edv_destroy(C.c_void_p(self.handle))
        except Exception:
            pass


class _FlatGrads:
    """ONE contiguous device buffer that receives every trainable gradient of a context (``edv_grad_bind_flat``): the engine
    writes each gradient into its slice, ``p.grad`` is a view of that slice, and the data-parallel all-reduce runs on the whole
    buffer in place (``parallel.allreduce_gradients``) -- no per-tensor copy anywhere in the step.  Rebuilt when the trainable
    set changes (the freeze schedule switches A/B -> U/V after the warm-up, trainer_end_to_end_video.py:324-339)."""

    def __init__(self, nat: _NativeCtx, names: Tuple[str, ...], shapes: Sequence[Tuple[int, ...]]):
        lib = _lib.load()
        n = len(names)
        self.names, self.shapes = names, [tuple(s) for s in shapes]
        self._c_names = (C.c_char_p * n)(*[k.encode() for k in names])
        self._c_numels = (C.c_int64 * n)(*[int(np.prod(s)) if len(s) else 1 for s in shapes])
        offs = (C.c_int64 * (n + 1))()
        _lib.check(lib.edv_grad_bind_flat(C.c_void_p(nat.handle), n, self._c_names, self._c_numels, None, 0, offs), "edv_grad_bind_flat (layout)")
        self.offsets = list(offs)[:n]
        self.flat = torch.zeros(int(offs[n]), device=nat.device, dtype=torch.float32)  # the padding between slices stays zero
        _lib.check(lib.edv_grad_bind_flat(C.c_void_p(nat.handle), n, self._c_names, self._c_numels, self.flat.data_ptr(), self.flat.numel(), offs),
                   "edv_grad_bind_flat")

    def views(self, flat: Optional[torch.Tensor] = None) -> List[torch.Tensor]:
        """Fresh view tensors of the slices of ``flat`` (default: the bound buffer).  Fresh, so that autograd's AccumulateGrad may
        adopt them as ``.grad`` without a copy: it does so only for a gradient tensor nobody else references."""
        flat = self.flat if flat is None else flat
        out = []
        for off, shp in zip(self.offsets, self.shapes):
            st, acc = [], 1
            for d in reversed(shp):
                st.append(acc)
                acc *= d
            out.append(flat.as_strided(shp, tuple(reversed(st)), off))
        return out

    def aliases(self, i: int, t: Optional[torch.Tensor]) -> bool:
        return t is not None and t.data_ptr() == self.flat.data_ptr() + 4 * self.offsets[i] and t.is_contiguous() and tuple(t.shape) == self.shapes[i]


# ---------------------------------------------------------------------------------------------
class _EdvFunction(torch.autograd.Function):
    """One training step through libendodav_hip: ``edv_forward`` under ``edv_set_train`` keeps the activations,
    ``edv_backward`` turns dL/d("disp", 0..3) into the gradients of the trainable tensors (trainer_end_to_end_video.py:731,
    :427-431).  The tensors are passed as inputs only so that autograd routes their gradients.

    A context keeps ONE set of activations: the forward records the context's generation and the backward hands it back, so a
    backward whose activations a later grad-enabled forward has overwritten raises instead of differentiating the wrong clip."""

    @staticmethod
    def forward(ctx, model, x, names, *params):
        outs = model._run_native(x, train=True)
        ctx.model, ctx.names, ctx.device = model, names, x.device
        ctx.nat = model._last
        gen = C.c_uint64()
        _lib.check(_lib.load().edv_generation(C.c_void_p(ctx.nat.handle), C.byref(gen)), "edv_generation")
        ctx.generation = gen.value
        ctx.shapes = [tuple(o.shape) for o in outs]
        ctx.params = params
        ctx.save_for_backward(outs[0])
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        (disp0,) = ctx.saved_tensors
        lib = _lib.load()
        nat = ctx.nat
        with torch.cuda.device(ctx.device):
            gs = [(g.detach().contiguous().float() if g is not None else torch.zeros(shp, device=ctx.device)) for g, shp in zip(gouts, ctx.shapes)]
            ptrs = (C.c_void_p * 4)(*[g.data_ptr() for g in gs])
            stream = C.c_void_p(_lib.stream_ptr(ctx.device))
            enc = any(".mlp.fc" in n for n in ctx.names)
            tmp = any(".ff.net.2." in n for n in ctx.names)
            hd = any(_is_head_conv(n) for n in ctx.names)
            rb = any(".residual_." in n for n in ctx.names)
            shapes = [tuple(p.shape) for p in ctx.params]
            fg = nat.flat
            if fg is None or fg.names != ctx.names or fg.shapes != shapes:
                fg = nat.flat = _FlatGrads(nat, ctx.names, shapes)
            # Gradient accumulation (a second backward without zero_grad): a .grad that IS a slice of the flat buffer would be
            # overwritten by the engine before autograd could add to it.  Keep the old sums and add them back in place.
            held = [i for i, p in enumerate(ctx.params) if p.is_leaf and fg.aliases(i, p.grad)]
            prev = fg.flat.clone() if held else None
            _lib.check(lib.edv_set_grad_scope(C.c_void_p(nat.handle), int(enc), int(tmp), int(hd), int(rb)), "edv_set_grad_scope")
            _lib.check(lib.edv_backward(C.c_void_p(nat.handle), C.c_uint64(ctx.generation), disp0.data_ptr(), ptrs, stream), "edv_backward")
            grads: List[Optional[torch.Tensor]] = fg.views()
            if held:
                if len(held) == len(grads):
                    fg.flat.add_(prev)
                else:
                    pv = fg.views(prev)
                    for i in held:
                        grads[i].add_(pv[i])
                for i in held:
                    grads[i] = None  # p.grad already is this slice, and it now holds old + new
        return (None, None, None, *grads)


class endodav(nn.Module):
    """Drop-in for ``models.endodav.endodav`` (constructor: endodav.py:53-73)."""

    def __init__(
        self,
        encoder="vitl",
        features=256,
        out_channels=[256, 512, 1024, 1024],
        use_bn=False,
        use_clstoken=False,
        num_frames=32,
        pe="ape",
        r=4,
        image_shape=(224, 280),
        lora_type="lora",
        pretrained_path=None,
        residual_block_indexes=[],
        include_cls_token=True,
        inv_sigmoid=False,
        temporal_lora=False,
        disable_conv_head=False,
        out_sigmoid=False,
    ):
        super().__init__()
        dim, depth, heads, img_size, taps = ENCODERS[encoder]  # KeyError for an unknown encoder, like endodav.py:92
        residual_block_indexes = [int(i) for i in residual_block_indexes]
        self.encoder = encoder
        self.intermediate_layer_idx = {encoder: list(taps)}
        self.image_shape = tuple(image_shape)
        self.r = r
        self.lora_type = lora_type
        self.num_frames = num_frames
        self.include_cls_token = include_cls_token
        self.inv_sigmoid, self.out_sigmoid = inv_sigmoid, out_sigmoid
        self.temporal_lora, self.disable_conv_head = temporal_lora, disable_conv_head
        self.features, self.out_channels = features, list(out_channels)
        self.use_clstoken = bool(use_clstoken)
        self.use_bn, self.pe = bool(use_bn), pe
        self.residual_block_indexes = [i for i in residual_block_indexes if 0 <= i < depth]
        self._dash_calls = 0  # DashLinear.FLAG of the reference, one shared count (every Dash layer sees every forward)

        self.pretrained = _Backbone(dim, depth, heads, img_size, self.residual_block_indexes)
        self.head = _Head(dim, features, out_channels, num_frames, disable_conv_head, self.use_clstoken, self.use_bn, pe)

        if lora_type != "none":  # endodav.py:102-137
            for blk in self.pretrained.blocks:
                blk.mlp.fc1 = _make_lora(lora_type, dim, 4 * dim, r)
                blk.mlp.fc2 = _make_lora(lora_type, 4 * dim, dim, r)
            if temporal_lora:
                for tm in self.head.motion_modules:
                    for blk in tm.temporal_transformer.transformer_blocks:
                        old = blk.ff.net[2]
                        blk.ff.net[2] = _make_lora(lora_type, old.in_features, old.out_features, r)

        if pretrained_path is not None:  # endodav.py:139-144
            print("load pretrained weight from {}\n".format(pretrained_path))
            path = os.path.join(pretrained_path, "video_depth_anything_{}.pth".format(self.encoder))
            self.load_state_dict(torch.load(path, map_location="cpu", weights_only=True), strict=False)

        mark_only_part_as_trainable(self.pretrained)  # endodav.py:146-148
        mark_only_part_as_trainable(self.head)
        mark_only_part_as_trainable(self.head.motion_modules, is_trainable=False)

        # ---- native contexts, one per device, created lazily.  The dict is shared (by reference) with
        # nn.DataParallel replicas, each of which then finds or creates the context of its own device.
        self._native: Dict[str, "_NativeCtx"] = {}
        self._capture = False
        # Arithmetic of the encoder's linears in inference: "f32" (fp32 products on the fp32 matrix pipe) or "bf16x6" (each operand split into three
        # bf16 terms, six bf16 MFMAs, fp32 accumulate: per-term error below fp32's unit roundoff -- include/endodav_hip.h, edv_set_products).
        # Inputs, outputs and accumulation are fp32 either way; training forwards always run "f32".  Initial value: EDV_PRODUCTS, else PRODUCTS_DEFAULT.
        self.products = os.environ.get("EDV_PRODUCTS", PRODUCTS_DEFAULT)

    # -------------------------------------------------------------------------------------
    def _config(self) -> _lib.EdvConfig:
        dim, depth, heads, _, taps = ENCODERS[self.encoder]
        cfg = _lib.EdvConfig()
        cfg.abi_version = _lib.ABI_VERSION
        cfg.embed_dim, cfg.depth, cfg.num_heads = dim, depth, heads
        cfg.taps = (C.c_int32 * 4)(*taps)
        cfg.features = self.features
        cfg.out_channels = (C.c_int32 * 4)(*self.out_channels)
        cfg.image_h, cfg.image_w = int(self.image_shape[0]), int(self.image_shape[1])
        cfg.num_frames = self.num_frames
        cfg.pos_tokens = int(self.pretrained.pos_embed.shape[1])
        cfg.lora_type = _lib.LORA_TYPES[self.lora_type]
        cfg.lora_rank = self.r
        cfg.include_cls_token = int(bool(self.include_cls_token))
        cfg.conv_head = int(not self.disable_conv_head)
        cfg.inv_sigmoid, cfg.out_sigmoid = int(bool(self.inv_sigmoid)), int(bool(self.out_sigmoid))
        cfg.temporal_lora = int(bool(self.temporal_lora))
        cfg.dash_active = int(self.lora_type == "dash" and self._dash_calls > DashLinear.WARMUP)
        cfg.use_clstoken = int(self.use_clstoken)
        cfg.use_bn, cfg.pe_rope = int(self.use_bn), int(self.pe == "rope")
        mask = 0
        for i in self.residual_block_indexes:
            mask |= 1 << i
        cfg.residual_mask = mask
        return cfg

    def _ensure_ctx(self, device: torch.device, lane: int = 0) -> int:
        """The engine context of ``device`` (created, bound and prepared on first use; re-folded when trainable tensors changed).  ``lane`` > 0:
        further contexts of the same device -- own workspace, same bound parameters -- for clips in flight beside each other
        (``pipeline.ClipsInFlight``); profiling / stage taps / the training path stay on lane 0."""
        lib = _lib.load()
        key = str(device) + ("+dash" if (self.lora_type == "dash" and self._dash_calls > DashLinear.WARMUP) else "") + (f"#{lane}" if lane else "")
        nat = self._native.get(key)
        if nat is None:
            h = C.c_void_p()
            cfg = self._config()
            _lib.check(lib.edv_create(C.byref(cfg), C.byref(h)), "edv_create")
            nat = self._native[key] = _NativeCtx(h.value, device)
        lib.edv_set_capture(C.c_void_p(nat.handle), int(self._capture))
        if lane == 0:
            self._last = nat
        # (re)bind + repack whenever a tensor moved or was written (optimizer step, load_state_dict)
        sd = self.state_dict(keep_vars=True)
        if self.pe == "rope":  # the rotary tables are not state (motion_module.py:221-225) but the engine reads them like weights
            sd.update({k: v for k, v in self.named_buffers() if k.endswith(".freqs_cis")})
        sig = tuple((k, v.data_ptr(), v._version) for k, v in sd.items())
        if sig != nat.sig and nat.sig is not None and len(sig) == len(nat.sig) and all(
                a[:2] == b[:2] and (a[2] == b[2] or _is_lora_factor(a[0]) or _is_head_conv(a[0]) or ".residual_." in a[0]) for a, b in zip(sig, nat.sig)):
            # the fine-tune loop: same tensors, only trainable ones written (optimizer.step) -> re-fold / re-pack those only
            _lib.check(lib.edv_refresh_lora(C.c_void_p(nat.handle), C.c_void_p(_lib.stream_ptr(device))), "edv_refresh_lora")
            nat.sig = sig
        if sig != nat.sig:
            for k, v in sd.items():
                if not v.is_floating_point():
                    continue
                if v.device != device or v.dtype != torch.float32 or not v.is_contiguous():
                    raise RuntimeError(f"parameter {k} must be a contiguous float32 tensor on {device} (got {v.dtype} on {v.device})")
                shape = (C.c_int64 * v.dim())(*v.shape)
                _lib.check(lib.edv_bind_param(C.c_void_p(nat.handle), k.encode(), v.data_ptr(), shape, v.dim()), f"edv_bind_param({k})")
            _lib.check(lib.edv_prepare(C.c_void_p(nat.handle), C.c_void_p(_lib.stream_ptr(device))), "edv_prepare")
            nat.sig = sig
        if self.products not in PRODUCTS:
            raise ValueError(f"products must be one of {sorted(PRODUCTS)} (got {self.products!r})")
        if lib.edv_get_products(C.c_void_p(nat.handle)) != PRODUCTS[self.products]:
            _lib.check(lib.edv_set_products(C.c_void_p(nat.handle), PRODUCTS[self.products], C.c_void_p(_lib.stream_ptr(device))), "edv_set_products")
        return nat.handle

    def _new_lane(self) -> int:
        """A lane id no one else uses (lane 0 is ``model(x)``'s own): one engine context per (device, lane), and a context must never run on two
        streams at once -- its workspace, stream-K counters and kept state are single-user."""
        self._lane_seq = getattr(self, "_lane_seq", 0) + 1
        return self._lane_seq

    def _drop_lane(self, lane: int) -> None:
        for key in [k for k in self._native if k.endswith(f"#{lane}")]:
            del self._native[key]  # _NativeCtx.__del__ destroys the engine context (and frees its workspace)

    def _dash_layers(self):
        return [m for m in self.modules() if isinstance(m, DashLinear)]

    @torch.no_grad()
    def _dash_step(self) -> None:
        """DashLinear's call counter (mylora/layers.py:558-583).  Calls 1..100 are plain LoRA; on call 101 every layer
        picks the 8 singular directions of W whose singular values the LoRA update changes most (relative change
        |diag(Uᵀ ΔW V)| / σ), stores them in weight_u_top / weight_vt_top and frees lora_index; from then on the engine
        folds U_top diag(lora_index) Vt_top into W as well.  The SVD is a one-off host-side selection step."""
        self._dash_calls += 1
        for m in self._dash_layers():
            m.FLAG = self._dash_calls
        if self._dash_calls == 1:
            for m in self._dash_layers():
                m.lora_index.requires_grad = m.weight_u_top.requires_grad = m.weight_vt_top.requires_grad = False
        if self._dash_calls == DashLinear.WARMUP + 1:
            for m in self._dash_layers():
                # on the host CPU: the ranking below is a discrete choice, and LAPACK's SVD is what the reference
                # goldens were captured with (rocSOLVER's rounding picks different directions on near-ties)
                dev = m.weight.device
                w = m.weight.detach().float().cpu()
                delta = (m.lora_B.detach().float().cpu() @ m.lora_A.detach().float().cpu()) * m.scaling
                u, sig, vt = torch.linalg.svd(w, full_matrices=False)
                dsig = torch.diag(u.T @ delta @ vt.T)
                top = torch.topk(dsig.abs() / sig.abs(), m.index).indices
                m.weight_u_top.data = u[:, top].contiguous().to(dev)
                m.weight_vt_top.data = vt[top, :].contiguous().to(dev)
                m.lora_index.requires_grad = True

    def output_shapes(self) -> List[Tuple[int, int]]:
        ph, pw = self.image_shape[0] // PATCH, self.image_shape[1] // PATCH
        if self.disable_conv_head:
            h, w = self.image_shape
            out = []
            for _ in range(4):
                out.append((h, w))
                h, w = h // 2, w // 2
            return out
        return [(ph * m, pw * m) for m in (16, 8, 4, 2)]

    # -------------------------------------------------------------------------------------
    def forward(self, x: torch.Tensor, lane: int = 0) -> Dict[Tuple[str, int], torch.Tensor]:
        """``x``: [B, T, 3, H, W] float32 in [0, 1] on a ROCm device → {("disp", s): [B*T, 1, h_s, w_s]}.  ``lane`` (not in the reference): which
        engine context of the device runs the clip -- inference only, used by ``pipeline.ClipsInFlight`` to keep consecutive clips in flight."""
        if lane and (torch.is_grad_enabled() or self.lora_type == "dash"):
            raise RuntimeError("lanes > 0 run inference under torch.no_grad() only (one set of kept activations / one dash counter per model)")
        if x.dim() != 5 or x.shape[2] != 3:
            raise ValueError(f"expected a clip [B, T, 3, H, W], got {tuple(x.shape)}")
        if not x.is_cuda:
            raise RuntimeError("endodav_amd runs on MI355X only: the clip must be a CUDA/ROCm tensor (there is no CPU fallback)")
        if self.use_bn and self.training:
            raise NotImplementedError("use_bn=True is built for eval() only: train-mode BatchNorm (batch statistics, running-average updates, "
                                      "util/blocks.py:80-86) is not; no reference script sets use_bn")
        train_names: List[str] = []
        if torch.is_grad_enabled() and lane == 0:
            if x.requires_grad:
                raise NotImplementedError("libendodav_hip does not produce the gradient of the input clip (nothing in the reference asks for it)")
            train_names = self._trainable_names()
        B, T, _, H, W = x.shape
        assert self.image_shape[0] % PATCH == 0, f"Input image height {self.image_shape[0]} is not a multiple of patch height {PATCH}"
        assert self.image_shape[1] % PATCH == 0, f"Input image width {self.image_shape[1]} is not a multiple of patch width: {PATCH}"
        if T > self.num_frames:  # motion_module.py:197: pe[:, :T] cannot broadcast
            raise RuntimeError(f"The size of tensor a ({T}) must match the size of tensor b ({self.num_frames}) at non-singleton dimension 1")
        x = x.detach().contiguous().float()
        if self.lora_type == "dash":
            self._dash_step()
        if train_names:
            sd = self.state_dict(keep_vars=True)
            outs = _EdvFunction.apply(self, x, tuple(train_names), *[sd[n] for n in train_names])
        else:
            outs = self._run_native(x, train=False, lane=lane)
        return {("disp", s): outs[s] for s in range(4)}

    def _run_native(self, x: torch.Tensor, train: bool, lane: int = 0):
        B, T, _, H, W = x.shape
        with torch.cuda.device(x.device):
            ctx = self._ensure_ctx(x.device, lane)
            lib = _lib.load()
            _lib.check(lib.edv_set_train(C.c_void_p(ctx), int(train)), "edv_set_train")
            outs = [torch.empty((B * T, 1, h, w), device=x.device, dtype=torch.float32) for (h, w) in self.output_shapes()]
            ptrs = (C.c_void_p * 4)(*[o.data_ptr() for o in outs])
            _lib.check(lib.edv_forward(C.c_void_p(ctx), x.data_ptr(), B, T, H, W, ptrs, C.c_void_p(_lib.stream_ptr(x.device))), "edv_forward")
        return outs

    def _trainable_names(self) -> List[str]:
        """state_dict names of the parameters that require grad, in state_dict order.  The HIP backward produces the
        gradients of the LoRA factors of mlp.fc1 / mlp.fc2 (what ``mark_only_part_as_trainable`` leaves trainable for
        lora / dvlora, endodav/layers.py:5-34); a trainable parameter outside that set is refused, not silently frozen."""
        names = [n for n, p in self.state_dict(keep_vars=True).items() if p.requires_grad]
        bad = [n for n in names if not (_is_lora_factor(n) or _is_head_conv(n) or (n.startswith("pretrained.blocks.") and ".residual_." in n))]
        if bad:
            raise NotImplementedError(f"libendodav_hip has no gradient for {bad[:4]}{' ...' if len(bad) > 4 else ''}: the HIP backward covers the LoRA "
                                      "factors of the encoder MLPs (and, with temporal_lora, of ff.net.2 in the motion modules), the residual bottleneck "
                                      "blocks residual_* and the output-head convolutions conv_depth_* / scratch.output_conv* (SURVEY.md §8f rank 3)")
        if any(_is_lora_factor(n) for n in names) and self.lora_type not in ("lora", "dvlora", "ssb", "dash"):
            raise NotImplementedError(f"the HIP backward supports lora_type 'lora', 'dvlora', 'ssb' and 'dash', not {self.lora_type!r}")
        return names

    # ---- debug taps for the per-stage parity tests --------------------------------------------
    def set_capture(self, on: bool) -> None:
        self._capture = bool(on)

    def stage(self, name: str) -> torch.Tensor:
        """Flat copy of an internal stage of the last forward (see ``edv_stage_copy``)."""
        lib = _lib.load()
        nat = self._last
        n = C.c_size_t()
        _lib.check(lib.edv_stage_copy(C.c_void_p(nat.handle), name.encode(), None, C.byref(n), None), "edv_stage_copy")
        out = torch.empty(n.value, device=nat.device, dtype=torch.float32)
        with torch.cuda.device(nat.device):
            _lib.check(lib.edv_stage_copy(C.c_void_p(nat.handle), name.encode(), out.data_ptr(), C.byref(n), C.c_void_p(_lib.stream_ptr(nat.device))), "edv_stage_copy")
        return out

    def launch_count(self) -> int:
        nat = getattr(self, "_last", None)
        return int(_lib.load().edv_last_launch_count(C.c_void_p(nat.handle))) if nat else 0

    def profile_enable(self, classes: Sequence[str]) -> None:
        """Bracket every launch of the named kernel classes (``_lib.KERNEL_CLASSES``) with HIP events."""
        nat = self._last
        mask = 0
        for c in classes:
            mask |= 1 << _lib.KERNEL_CLASSES[c]
        _lib.check(_lib.load().edv_profile_enable(C.c_void_p(nat.handle), mask), "edv_profile_enable")

    def set_encoder_streams(self, n: int) -> None:
        """0 = automatic, 1..4 = that many frame groups on internal streams, -1 = the initial setting (automatic unless
        EDV_ENC_STREAMS was set) -- see ``edv_set_encoder_streams``."""
        _lib.check(_lib.load().edv_set_encoder_streams(C.c_void_p(self._last.handle), int(n)), "edv_set_encoder_streams")

    def profile_set(self, classes: Sequence[str]) -> None:
        """Change the bracketed kernel classes without dropping what was recorded so far."""
        mask = 0
        for c in classes:
            mask |= 1 << _lib.KERNEL_CLASSES[c]
        _lib.check(_lib.load().edv_profile_set_mask(C.c_void_p(self._last.handle), mask), "edv_profile_set_mask")

    def profile_read(self, cls: str) -> Tuple[int, float]:
        """(launches, summed milliseconds) of one kernel class since the last read."""
        n, ms = C.c_int32(), C.c_double()
        _lib.check(_lib.load().edv_profile_read(C.c_void_p(self._last.handle), _lib.KERNEL_CLASSES[cls], C.byref(n), C.byref(ms)), "edv_profile_read")
        return n.value, ms.value

    def profile_work(self, cls: str) -> Tuple[float, float]:
        """(FLOP, algorithmic bytes) of the launches of one kernel class bracketed since the last read."""
        fl, by = C.c_double(), C.c_double()
        _lib.check(_lib.load().edv_profile_work(C.c_void_p(self._last.handle), _lib.KERNEL_CLASSES[cls], C.byref(fl), C.byref(by)), "edv_profile_work")
        return fl.value, by.value

    def flat_gradients(self, params: Sequence[torch.nn.Parameter]) -> Optional[torch.Tensor]:
        """The flat gradient buffer of the last backward if ``params`` are exactly the tensors it covers and every ``p.grad`` IS its
        slice of that buffer (then one in-place all-reduce of the buffer reduces every gradient); None otherwise."""
        nat = getattr(self, "_last", None)
        fg = nat.flat if nat is not None else None
        if fg is None:
            return None
        sd = self.state_dict(keep_vars=True)
        mine = [sd.get(n) for n in fg.names]
        wanted = {id(p) for p in params if p.requires_grad}
        if {id(p) for p in mine} != wanted or any(p is None for p in mine):
            return None
        if not all(fg.aliases(i, p.grad) for i, p in enumerate(mine)):
            return None
        return fg.flat

    def device_bytes(self) -> int:
        nat = getattr(self, "_last", None)
        return int(_lib.load().edv_device_bytes(C.c_void_p(nat.handle))) if nat else 0

    # -------------------------------------------------------------------------------------
    def infer_video_depth(self, frames, input_size=518, device="cuda", shard_windows=False):
        """Sliding-window inference over a whole video (endodav.py:162-254).

        ``frames``: uint8 [N, H, W, 3].  Returns float32 [N, H, W].  Windows of 32 frames, step 22; the
        first 10 slots of every later window are refilled with key frames of the previous window's
        INPUT; windows are stitched by a least-squares scale/shift on the overlap and a linear
        cross-fade over 8 frames.  ``shard_windows=True`` (not in the reference): every rank of the process group calls this with the SAME
        video and runs a share of its windows; rank 0 gets the result, the others None (``video.infer_video_depth``).
        """
        from .video import infer_video_depth as _impl

        return _impl(self, frames, input_size=input_size, device=device, shard_windows=shard_windows)
