"""Freeze policy of the reference, ``models/endodav/layers.py:5-34``.

A parameter stays trainable iff its name contains one of the LoRA tags of the current phase
(``lora_A``/``lora_B`` during warm-up, ``lora_U``/``lora_V`` afterwards), ``residual_`` or
``conv_depth_``; ``train_output_conv`` additionally unfreezes ``output_conv*``.  The set this
produces is also the gradient all-reduce set of the data-parallel fine-tune path.
"""
from __future__ import annotations

import torch.nn as nn

_ALWAYS = ("residual_", "conv_depth_")


def mark_only_part_as_trainable(model: nn.Module, bias: str = "none", warm_up: bool = True, is_trainable: bool = True,
                                train_output_conv: bool = False) -> None:
    tags = (("lora_A", "lora_B") if warm_up else ("lora_U", "lora_V")) + _ALWAYS
    for name, p in model.named_parameters():
        p.requires_grad = bool(is_trainable) if any(t in name for t in tags) else False
        if train_output_conv and "output_conv" in name:
            p.requires_grad = True
    if bias == "none":
        return
    if bias == "all":
        for name, p in model.named_parameters():
            if "bias" in name:
                p.requires_grad = True
        return
    if bias == "lora_only":
        # The reference tests isinstance(m, backbones.galora.LoRALayer) (layers.py:27-32); the model is
        # built from mylora layers, which never match, so this branch changes nothing there either.
        return
    raise NotImplementedError(bias)


def trainable_parameters(model: nn.Module):
    """The parameters a fine-tune step updates — and, under data parallelism, all-reduces."""
    return [p for p in model.parameters() if p.requires_grad]
