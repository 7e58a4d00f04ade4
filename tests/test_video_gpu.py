"""infer_video_depth on MI355X: windowing, key-frame reuse and stitching against the reference golden,
then one real end-to-end run (frames -> HIP forward -> stitched depth)."""
import numpy as np
import pytest
import torch

import endodav_amd
from endodav_amd import synth
from tests import helpers as H
from tests.golden.make_golden import VIDEO_CASE, fake_window_disp

pytestmark = pytest.mark.gpu


def _model(h, w, cuda):
    m = endodav_amd.endodav(encoder="vits", features=32, out_channels=[32, 32, 64, 64], image_shape=(h, w), lora_type="none",
                            disable_conv_head=True).eval()
    synth.fill_module_(m)
    return m.to(cuda)


def test_windowing_keyframes_and_stitching_match_reference(cuda):
    g = H.load_golden("video_stitch")
    n, h, w = VIDEO_CASE["n_frames"], VIDEO_CASE["h"], VIDEO_CASE["w"]
    model = _model(h, w, cuda)
    frames = (synth.uniform("video:frames", (n, h, w, 3), 0.0, 1.0) * 255).astype(np.uint8)
    seen = []

    def recorder(x, lane=0):  # same stand-in forward the reference ran when the golden was made (lane: which engine context, pipeline.ClipsInFlight)
        assert x.is_cuda and x.shape == (1, 32, 3, h, w)
        seen.append(x[0].mean(dim=(1, 2, 3)).double().cpu().numpy())
        return {("disp", 0): torch.from_numpy(fake_window_disp(len(seen) - 1, h, w)).to(x.device)}

    model.forward = recorder
    out = model.infer_video_depth(frames, device="cuda:0")
    assert out.shape == (n, h, w) and out.dtype == np.float32
    assert np.abs(np.stack(seen) - g["window_input_means"]).max() < 1e-6  # padding + key-frame substitution
    assert np.abs(out - g["out"]).max() <= 2e-6 * np.abs(g["out"]).max()  # resize-to-native (identity) + stitching


def test_real_video_run_with_native_resolution_resize(cuda):
    """Frames larger than image_shape: bicubic pre-resize (unpinned), forward, bilinear back, stitch."""
    h, w = 42, 56
    model = _model(h, w, cuda)
    frames = (synth.synth_clip(1, 25, 60, 80, seed=4, kind="tissue")[0].transpose(0, 2, 3, 1) * 255).astype(np.uint8)
    out = model.infer_video_depth(frames, device="cuda:0")
    assert out.shape == (25, 60, 80) and np.isfinite(out).all() and (out >= 0).all() and out.max() > 0
    # 25 frames = two windows (the second one exercises the pipelined upload / download).  Stitching rewrites only the last INTERP_LEN = 8
    # slots of window 0, so frames 0..23 must equal the direct forward on the HIP-resized clip of window 0, brought back to the frame size
    # with torch's own bilinear (align_corners=True, endodav.py:203-204) -- the streams, the uint8 upload and the stitching add nothing
    from endodav_amd import video

    runner = video.HipWindowRunner(model, np.ascontiguousarray(frames), torch.device("cuda:0"))
    sources = video.window_sources(25)
    assert len(sources) == 2
    x = runner.resized_clip(sources[0])
    assert x.shape == (1, 32, 3, h, w)
    with torch.no_grad():
        disp = model(x)[("disp", 0)]  # [32, 1, h, w]
        direct = torch.nn.functional.interpolate(disp, size=(60, 80), mode="bilinear", align_corners=True)[:24, 0].cpu().numpy()
    assert np.abs(out[:24] - direct).max() <= 2e-6 * np.abs(direct).max()
    out2 = model.infer_video_depth(frames, device="cuda:0")
    assert np.array_equal(out, out2)  # and the pipelined run is reproducible bit for bit


def test_evaluate_video_end_to_end(cuda):
    """evaluate_depth_video.py's loop on synthetic clips through the real HIP model: plumbing + report format."""
    from endodav_amd import evaluate as ev

    model = _model(42, 56, cuda)
    ds = ev.SyntheticVideos(n_clips=2, n_frames=5, height=42, width=56)
    res = ev.evaluate_video(model, ds, depth_align="scale_shift", device="cuda:0")
    assert res["errors"].shape == (10, 7) and res["temporal"].shape == (8, 2) and np.isfinite(res["errors"]).all()
    assert len(res["inference_times"]) == 2 and res["aligns"].shape == (2, 4)
    txt = ev.format_results(res)
    assert txt.startswith("    abs_rel") and "average inference time" in txt
