"""CPU: host-side logic and the drop-in boundary (no GPU compute calls)."""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest
import torch

import endodav_amd
from endodav_amd import _lib, synth, video
from tests import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---- C ABI ---------------------------------------------------------------------------------------
def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "endodav_hip.h")).read()
    declared = set(re.findall(r"\b(edv_[a-z0-9_]+)\s*\(", header))
    declared -= {"edv_lora_type"}
    assert len(declared) >= 25
    lib = C.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/endodav_hip.h but not exported"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)


def test_abi_version_and_config_layout():
    lib = _lib.load()
    assert lib.edv_abi_version() == _lib.ABI_VERSION
    assert C.sizeof(_lib.EdvConfig) == 4 * 29  # 29 32-bit slots incl. the two int[4] arrays


def test_create_validates_config_without_gpu():
    lib = _lib.load()
    cfg = _lib.EdvConfig()
    h = C.c_void_p()
    assert lib.edv_create(C.byref(cfg), C.byref(h)) != 0
    assert b"ABI" in lib.edv_last_error()
    m = endodav_amd.endodav(encoder="vits", features=64, out_channels=[48, 96, 192, 384], image_shape=(518, 518), disable_conv_head=True)
    cfg = m._config()
    assert lib.edv_create(C.byref(cfg), C.byref(h)) == 0
    oh, ow = C.c_int32(), C.c_int32()
    for s, want in enumerate([(518, 518), (259, 259), (129, 129), (64, 64)]):
        assert lib.edv_output_shape(h, s, C.byref(oh), C.byref(ow)) == 0 and (oh.value, ow.value) == want
    cfg.image_h = 500  # not a multiple of 14 (patch_embed.py:72)
    h2 = C.c_void_p()
    assert lib.edv_create(C.byref(cfg), C.byref(h2)) != 0 and b"14" in lib.edv_last_error()


# ---- drop-in surface --------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def ref_keys():
    with open(os.path.join(H.GOLDEN, "state_keys.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("combo", ["vits_dvlora_vda", "vits_lora_conv", "vits_ssb_vda_tlora", "vits_dash_conv", "vits_none_vda", "vitl_dvlora_vda",
                                   "vits_clstoken_resblocks", "vits_bn_rope"])
def test_state_dict_keys_shapes_and_trainable_set_match_reference(ref_keys, combo):
    entry = ref_keys[combo]
    m = endodav_amd.endodav(**entry["kwargs"], pretrained_path=None)
    got = [[k, list(v.shape)] for k, v in m.state_dict().items()]
    assert sorted(map(tuple, map(lambda kv: (kv[0], tuple(kv[1])), got))) == sorted((k, tuple(s)) for k, s in entry["keys"])
    assert [k for k, _ in got] == [k for k, _ in entry["keys"]], "key ORDER differs from the reference"
    assert sorted(n for n, p in m.named_parameters() if p.requires_grad) == entry["trainable"]


def test_constructor_errors_follow_the_reference():
    with pytest.raises(KeyError):
        endodav_amd.endodav(encoder="vitg")
    with pytest.raises(AssertionError):
        endodav_amd.endodav(encoder="vits", features=64, out_channels=[48, 96, 192, 384], num_frames=0)
    with pytest.raises(FileNotFoundError):
        endodav_amd.endodav(encoder="vits", features=64, out_channels=[48, 96, 192, 384], pretrained_path="/nonexistent")
    m = endodav_amd.endodav(encoder="vitb", features=128, out_channels=[96, 192, 384, 768])  # extension, SURVEY.md §0.5
    assert m.pretrained.embed_dim == 768 and len(m.pretrained.blocks) == 12


def test_forward_refuses_cpu_tensors():
    m = endodav_amd.endodav(encoder="vits", features=32, out_channels=[32, 32, 64, 64], image_shape=(42, 56), disable_conv_head=True)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.rand(1, 2, 3, 42, 56))


def test_mark_only_part_as_trainable_phases():
    m = endodav_amd.endodav(encoder="vits", features=64, out_channels=[48, 96, 192, 384], lora_type="dvlora")
    names = lambda: {n for n, p in m.named_parameters() if p.requires_grad}
    warm = names()
    assert any("lora_A" in n for n in warm) and any("conv_depth_" in n for n in warm) and not any("lora_U" in n for n in warm)
    assert not any("motion_modules" in n for n in warm)
    endodav_amd.mark_only_part_as_trainable(m.pretrained, warm_up=False)
    after = names()
    assert any("lora_U" in n for n in after) and not any("lora_A" in n for n in after)
    endodav_amd.mark_only_part_as_trainable(m.head, train_output_conv=True)
    with pytest.raises(NotImplementedError):
        endodav_amd.mark_only_part_as_trainable(m, bias="nope")


def test_synth_is_portable_and_stable():
    # known answers: any host must regenerate bit-identical weights
    a = synth.uniform01("w:pretrained.cls_token", 4)
    assert a.dtype == np.float32 and np.all((a >= 0) & (a < 1))
    assert synth.fnv1a64("abc") == 0xE71FA2190541574B
    b = synth.uniform01("w:pretrained.cls_token", 4)
    assert np.array_equal(a, b) and not np.array_equal(a, synth.uniform01("w:pretrained.pos_embed", 4))
    sd = synth.synth_state({"head.motion_modules.0.temporal_transformer.proj_out.weight": (8, 8), "x.ls1.gamma": (4,), "a.pos_encoder.pe": (1, 2, 2)})
    assert "a.pos_encoder.pe" not in sd and np.abs(sd["head.motion_modules.0.temporal_transformer.proj_out.weight"]).max() > 0
    assert sd["x.ls1.gamma"].min() >= 0.2


# ---- whole-video host logic ------------------------------------------------------------------------
def test_window_plan_and_resize_rule():
    assert video.window_plan(60) == (76, [0, 22, 44])  # endodav.py:188-189: pad to k*22 + 10
    assert video.window_plan(22) == (32, [0])
    assert video.window_plan(23) == (54, [0, 22])
    assert video.window_plan(1) == (32, [0])
    assert video.lower_bound_size(1280, 1024, 280, 224) == (280, 224)
    assert video.lower_bound_size(518, 518, 518, 518) == (518, 518)
    w, h = video.lower_bound_size(640, 480, 518, 518)
    assert (w, h) == (686, 518) and w % 14 == 0 and h % 14 == 0


def test_stitching_matches_reference_golden():
    from tests.golden.make_golden import VIDEO_CASE, fake_window_disp

    g = H.load_golden("video_stitch")
    n, h, w = VIDEO_CASE["n_frames"], VIDEO_CASE["h"], VIDEO_CASE["w"]
    total, starts = video.window_plan(n)
    assert len(starts) == g["window_input_means"].shape[0]
    wins = [fake_window_disp(i, h, w)[:, 0] for i in range(len(starts))]
    out = video.stitch_windows(wins, n)
    assert out.shape == g["out"].shape and out.dtype == np.float32
    assert np.abs(out - g["out"]).max() <= 1e-6 * np.abs(g["out"]).max()
