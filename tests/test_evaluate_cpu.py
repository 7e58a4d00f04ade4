"""CPU: metric helpers against known answers captured from the reference's own functions, and the evaluation
harness end to end with a stand-in depther (no GPU)."""
import numpy as np
import pytest

from endodav_amd import evaluate as ev
from tests import helpers as H
from tests.golden.make_golden import metrics_inputs


@pytest.fixture(scope="module")
def kat():
    return H.load_golden("metrics_kat"), metrics_inputs()


def test_compute_errors_and_disp_to_depth(kat):
    g, x = kat
    valid = (x["gt"] > 1e-3) & (x["gt"] < 150)
    assert np.allclose(ev.compute_errors(x["gt"], x["pred"], valid), g["compute_errors"], rtol=1e-6)
    sd, d = ev.disp_to_depth(x["disp"], 0.1, 150.0)
    assert np.allclose(sd, g["scaled_disp"], rtol=1e-6) and np.allclose(d, g["depth"], rtol=1e-6)


def test_alignments(kat):
    g, x = kat
    ms, ratio = ev.median_scaling(x["gt"], x["pred"].copy())
    assert np.isclose(ratio, g["median_ratio"], rtol=1e-6) and np.allclose(ms, g["median_scaled"], rtol=1e-6)
    al, *params = ev.align_shift_and_scale(x["gt"], x["pred"].copy())
    assert np.allclose(params, g["align_params"], rtol=1e-6) and np.allclose(al, g["aligned"], rtol=1e-5)


def test_temporal_metrics(kat):
    g, x = kat
    mask = np.ones_like(x["depth_a"], dtype=bool)
    mask[:3] = False
    i2w_a, i2w_b = np.linalg.inv(x["K"] @ x["pose_a"]), np.linalg.inv(x["K"] @ x["pose_b"])
    assert np.isclose(ev.tae(x["depth_a"], mask, i2w_a, x["depth_b"], mask, i2w_b), g["tae"], rtol=1e-6)
    assert np.isclose(ev.tas(x["depth_a"], mask, i2w_a, x["depth_b"], mask, i2w_b), g["tas"], rtol=1e-6)


class _PerfectDepther:
    """infer_video_depth stand-in: returns the disparity whose depth is 1.1 x the ground truth."""

    def __init__(self, ds):
        self.gt = {i["filename"]: i["depths"] for i in ds}
        self.cur = iter(ds)

    def infer_video_depth(self, colors, device=None):
        item = next(self.cur)
        depth = 1.1 * item["depths"]
        return ((1.0 / depth - 1.0 / 150.0) / (1.0 / 0.1 - 1.0 / 150.0)).astype(np.float32)


def test_harness_with_scaled_prediction_is_exact_after_median_scaling():
    ds = ev.SyntheticVideos(n_clips=2, n_frames=6, height=28, width=42)
    res = ev.evaluate_video(_PerfectDepther(ds), ds, depth_align="scale", device=None)
    assert res["errors"].shape == (12, 7) and res["temporal"].shape == (10, 2)
    assert np.allclose(res["ratios"], 1 / 1.1, rtol=1e-4)
    assert res["errors"][:, 0].max() < 1e-4 and np.allclose(res["errors"][:, 4:], 1.0)  # abs_rel ~ 0, a1..a3 = 1
    txt = ev.format_results(res)
    assert txt.splitlines()[0].split("|")[0].strip() == "abs_rel" and "average inference time" in txt
    assert txt.count("&") == 18
