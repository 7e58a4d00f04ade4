"""Parity cases shared by the golden generator, the oracle tests and the GPU tests.

Each case = reference constructor kwargs (``models/endodav/endodav.py:53-73``) + a synthetic
clip shape.  Weights and clips come from ``endodav_amd.synth`` (name-keyed, portable), so a
fixture only has to hold outputs.
"""
from __future__ import annotations

VITS = dict(encoder="vits", features=64, out_channels=[48, 96, 192, 384])
VITS_SMALL_HEAD = dict(encoder="vits", features=32, out_channels=[32, 32, 64, 64])
VITL_SMALL_HEAD = dict(encoder="vitl", features=32, out_channels=[32, 64, 64, 64])
VITB = dict(encoder="vitb", features=128, out_channels=[96, 192, 384, 768])      # extension, SURVEY.md section 0.5 (head widths: endodac.py:184-199)
VITL = dict(encoder="vitl", features=256, out_channels=[256, 512, 1024, 1024])   # the reference's constructor defaults

# name -> (ctor kwargs, (B, T, H, W) of the input clip, clip kind, store)
# store: "full" keeps whole outputs; "strided" keeps every 7th pixel of disp 0 + stats.
CASES = {
    # --- micro cases: tiny grids, every code path, full outputs stored -----------------
    "micro_vda_dvlora": (dict(VITS_SMALL_HEAD, image_shape=(42, 56), lora_type="dvlora", disable_conv_head=True), (1, 3, 42, 56), "uniform", "full"),
    "micro_vda_lora_b2": (dict(VITS_SMALL_HEAD, image_shape=(42, 42), lora_type="lora", disable_conv_head=True), (2, 2, 42, 42), "uniform", "full"),
    "micro_conv_dvlora": (dict(VITS_SMALL_HEAD, image_shape=(42, 56), lora_type="dvlora"), (1, 3, 42, 56), "uniform", "full"),
    "micro_conv_invsig_ssb": (dict(VITS_SMALL_HEAD, image_shape=(42, 56), lora_type="ssb", inv_sigmoid=True), (1, 2, 42, 56), "tissue", "full"),
    "micro_vda_none_outsig": (dict(VITS_SMALL_HEAD, image_shape=(42, 42), lora_type="none", disable_conv_head=True, out_sigmoid=True), (1, 4, 42, 42), "uniform", "full"),
    "micro_vda_nocls": (dict(VITS_SMALL_HEAD, image_shape=(42, 56), lora_type="dvlora", disable_conv_head=True, include_cls_token=False), (1, 2, 42, 56), "uniform", "full"),
    "micro_vda_temporal_lora": (dict(VITS_SMALL_HEAD, image_shape=(42, 56), lora_type="lora", disable_conv_head=True, temporal_lora=True), (1, 3, 42, 56), "uniform", "full"),
    "micro_resize_in": (dict(VITS_SMALL_HEAD, image_shape=(42, 56), lora_type="dvlora", disable_conv_head=True), (1, 2, 64, 80), "tissue", "full"),
    "micro_t1": (dict(VITS_SMALL_HEAD, image_shape=(42, 56), lora_type="dvlora", disable_conv_head=True), (1, 1, 42, 56), "uniform", "full"),
    "micro_t32": (dict(VITS_SMALL_HEAD, image_shape=(42, 42), lora_type="dvlora", disable_conv_head=True), (1, 32, 42, 42), "tissue", "full"),
    # num_frames > 32 (the constructor takes any, dpt_temporal.py:35-40; the reference's scripts keep 32): clips longer than one 32-frame window
    "micro_t48": (dict(VITS_SMALL_HEAD, image_shape=(42, 42), lora_type="dvlora", disable_conv_head=True, num_frames=64), (1, 48, 42, 42), "tissue", "full"),
    "micro_vitl": (dict(VITL_SMALL_HEAD, image_shape=(42, 56), lora_type="dvlora", disable_conv_head=True), (1, 2, 42, 56), "uniform", "full"),
    # --- options no reference script sets, still part of the constructor surface --------------------------
    "micro_clstoken": (dict(VITS_SMALL_HEAD, image_shape=(42, 56), lora_type="dvlora", disable_conv_head=True, use_clstoken=True), (1, 2, 42, 56), "uniform", "full"),
    "micro_clstoken_nocls": (dict(VITS_SMALL_HEAD, image_shape=(42, 56), lora_type="lora", use_clstoken=True, include_cls_token=False), (1, 2, 42, 56), "uniform", "full"),
    "micro_dash": (dict(VITS_SMALL_HEAD, image_shape=(42, 56), lora_type="dash", disable_conv_head=True), (1, 2, 42, 56), "uniform", "full"),
    "micro_dash_active": (dict(VITS_SMALL_HEAD, image_shape=(42, 56), lora_type="dash", disable_conv_head=True), (1, 2, 42, 56), "uniform", "full"),
    "micro_bn": (dict(VITS_SMALL_HEAD, image_shape=(42, 56), lora_type="dvlora", disable_conv_head=True, use_bn=True), (1, 2, 42, 56), "uniform", "full"),
    "micro_rope": (dict(VITS_SMALL_HEAD, image_shape=(42, 56), lora_type="dvlora", disable_conv_head=True, pe="rope"), (1, 5, 42, 56), "tissue", "full"),
    "micro_rope_bn_conv": (dict(VITS_SMALL_HEAD, image_shape=(42, 42), lora_type="lora", use_bn=True, pe="rope"), (2, 3, 42, 42), "uniform", "full"),
    # residual bottleneck blocks: the reference hard-wires their grid to 16x20 patches, i.e. image_shape (224, 280)
    "resblock_224x280": (dict(VITS_SMALL_HEAD, image_shape=(224, 280), lora_type="dvlora", residual_block_indexes=[2, 5, 8, 11]), (1, 2, 224, 280), "tissue", "strided"),
    # --- reference default geometry (224x280 from 256x320 frames, trainer_end_to_end_video.py:61)
    "vits_224x280_t2": (dict(VITS, image_shape=(224, 280), lora_type="dvlora", disable_conv_head=True), (1, 2, 256, 320), "tissue", "strided"),
    "vits_224x280_conv_t2": (dict(VITS, image_shape=(224, 280), lora_type="dvlora"), (1, 2, 256, 320), "tissue", "strided"),
    # --- BASELINE config 1: ViT-S 518x518 T=4 -------------------------------------------
    "vits_518_t4": (dict(VITS, image_shape=(518, 518), lora_type="dvlora", disable_conv_head=True), (1, 4, 518, 518), "uniform", "strided"),
    # --- BASELINE configs 3 and 5 at full size (round 2): ViT-B 518x518 T=16 and ViT-L 518x518 T=32, and ViT-L's real widths with
    # T=32 on a small non-square grid (every temporal-attention width of config 5: C = 1024, 1024, 256, 256).  "vitb" is not
    # constructible through the reference's endodav (KeyError): make_golden.py composes the reference's OWN
    # vit_base (vision_transformer.py:368-382) with its DPTHeadPyramid through the 'vits' slot of the constructor.
    "vitb_518_t16": (dict(VITB, image_shape=(518, 518), lora_type="dvlora", disable_conv_head=True), (1, 16, 518, 518), "tissue", "strided"),
    "vitl_518_t32": (dict(VITL, image_shape=(518, 518), lora_type="dvlora", disable_conv_head=True), (1, 32, 518, 518), "tissue", "strided"),
    "vitl_126x182_t32": (dict(VITL, image_shape=(126, 182), lora_type="dvlora", disable_conv_head=True), (1, 32, 126, 182), "tissue", "strided"),
}

# Too large for the CPU suite's replay (minutes of oracle on 8 threads): for these the CPU test only checks the oracle-vs-reference
# agreement recorded when the fixture was made; the GPU suite runs the oracle on vitb_518_t16 at full size on the box's 16 threads.
CPU_REPLAY_SKIP = ("vitb_518_t16", "vitl_518_t32")

DASH_WARMUP_CALLS = 100  # mylora/layers.py:542 (self.warmup)
# cases replayed by the (CPU) oracle test on every run; the 518 case takes ~5 s
STAGE_KEYS = ("tokens", "block0", "tap0", "tap3", "mm0", "mm1", "path4", "path3", "path2", "path1")
