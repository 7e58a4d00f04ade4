"""pipeline.ClipsInFlight on MI355X: several consecutive clips in flight on one GPU (one engine context + stream per lane) must give, clip by clip,
exactly the bits of the one-at-a-time forward -- different clips on the lanes, the weights changed between two batches (every lane re-folds), the
result handed to the caller's stream in order.  Kernel-level co-residency of two forwards is what found round 2's LDS race, so this doubles as a
concurrency soak of the whole kernel set at small sizes."""
import pytest
import torch

import endodav_amd
from endodav_amd import synth
from endodav_amd.pipeline import ClipsInFlight

pytestmark = pytest.mark.gpu


def _model(cuda, image=(70, 98)):
    m = endodav_amd.endodav(encoder="vits", features=64, out_channels=[48, 96, 192, 384], image_shape=image, lora_type="dvlora", disable_conv_head=True).eval()
    synth.fill_module_(m)
    return m.to(cuda)


@pytest.mark.parametrize("depth", [1, 2, 3])
def test_clips_in_flight_equal_the_one_at_a_time_forward(cuda, depth):
    model = _model(cuda)
    clips = [torch.from_numpy(synth.synth_clip(1, 3 + (i % 2), 70, 98, seed=10 + i, kind="tissue")).to(cuda) for i in range(7)]  # T alternates 3 / 4
    with torch.no_grad():
        ref = [[o.clone() for o in model(x).values()] for x in clips]
    flight = ClipsInFlight(model, cuda, depth=depth)
    for rnd in range(3):
        got = list(flight.run(clips))
        assert len(got) == len(clips)
        for i, (g, r) in enumerate(zip(got, ref)):
            assert list(g.keys()) == [("disp", s) for s in range(4)]
            for a, b in zip(g.values(), r):
                assert torch.equal(a, b), f"round {rnd}, clip {i}"
    # weights change (what an optimizer step or load_state_dict does): every lane must pick the new values up
    with torch.no_grad():
        model.pretrained.blocks[0].mlp.fc1.lora_B.mul_(1.5)
        model.head.scratch.output_conv2[2].bias.add_(0.01)
        ref2 = [[o.clone() for o in model(x).values()] for x in clips[:4]]
    assert not torch.equal(ref2[0][0], ref[0][0])
    for g, r in zip(flight.run(clips[:4]), ref2):
        for a, b in zip(g.values(), r):
            assert torch.equal(a, b)


def test_submit_orders_against_the_callers_stream(cuda):
    """The clip is produced on the caller's stream right before submit(), and consumed right after result(): no explicit synchronisation."""
    model = _model(cuda)
    base = torch.from_numpy(synth.synth_clip(1, 3, 70, 98, seed=3, kind="tissue")).to(cuda)
    with torch.no_grad():
        want = [model((base * s).clamp(0, 1))[("disp", 0)].clone().mean() for s in (1.0, 0.7, 0.4, 0.9)]
    flight = ClipsInFlight(model, cuda, depth=3)
    hs = []
    for s in (1.0, 0.7, 0.4, 0.9):
        x = (base * s).clamp(0, 1)  # enqueued on the current stream; submit() must wait for it
        hs.append(flight.submit(x))
    got = [h.result()[("disp", 0)].mean() for h in hs]
    for a, b in zip(got, want):
        assert torch.equal(a, b)


def test_lanes_are_inference_only(cuda):
    model = _model(cuda)
    x = torch.from_numpy(synth.synth_clip(1, 3, 70, 98, seed=3)).to(cuda)
    with pytest.raises(RuntimeError, match="inference"):
        model(x, lane=1)  # grad mode is on
    assert ClipsInFlight.auto_depth(_model(cuda, (518, 518)), 8) == 3 and ClipsInFlight.auto_depth(_model(cuda, (518, 518)), 32) == 1


def test_two_flights_and_the_plain_forward_run_beside_each_other(cuda):
    """An engine context is single-user (workspace, stream-K arrival counters): every ClipsInFlight owns its lanes and model(x) keeps lane 0, so
    all of them may be in flight at once without a synchronisation in between -- and releasing a flight frees its contexts."""
    model = _model(cuda, (126, 182))
    x = torch.from_numpy(synth.synth_clip(1, 8, 126, 182, seed=5, kind="tissue")).to(cuda)
    with torch.no_grad():
        ref = [o.clone() for o in model(x).values()]
    a, b = ClipsInFlight(model, cuda, depth=3), ClipsInFlight(model, cuda, depth=2)
    for rnd in range(4):
        ha = [a.submit(x, resident=True) for _ in range(5)]
        hb = [b.submit(x, resident=True) for _ in range(4)]
        with torch.no_grad():
            plain = [list(model(x).values()) for _ in range(2)]
        for h in ha + hb:
            for u, v in zip(h.result().values(), ref):
                assert torch.equal(u, v), f"round {rnd}"
        for o in plain:
            for u, v in zip(o, ref):
                assert torch.equal(u, v)
    assert sorted(a.lanes + b.lanes) == [1, 2, 3, 4, 5]
    n_ctx = len(model._native)
    a.close()
    assert len(model._native) == n_ctx - 3
