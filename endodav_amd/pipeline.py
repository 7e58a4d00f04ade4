"""Host <-> HBM transfers off the critical path of the per-clip forward.

The reference moves every clip to the device synchronously before the forward and every map back synchronously after it
(``cur_input.to(device)`` endodav.py:197, ``.cpu()`` :206; ``inputs[key].to(self.device)`` trainer_end_to_end_video.py:729-730), so the
PCIe time adds to the step.  On MI355X the clip (25.8 MB at 518x518 T=8) and the four maps cross PCIe Gen5 in about a millisecond,
against a ~10 ms forward: with the upload of clip k+1 on a copy stream while clip k computes, and the download of clip k-1's
maps on a second copy stream, the transfers vanish behind the forward.  Two pinned host buffers and two device buffers per
direction; the host blocks only when it is about to reuse one.
"""
from __future__ import annotations

from typing import Callable, Dict, Iterator, List, Optional, Sequence, Tuple

import torch


class InFlight:
    """One submitted clip: ``result()`` makes the caller's current stream wait for it and returns its output dict."""

    def __init__(self, out: Dict, done: torch.cuda.Event, pool: List[torch.cuda.Event]):
        self._out, self._done, self._pool = out, done, pool

    def result(self) -> Dict:
        cur = torch.cuda.current_stream(next(iter(self._out.values())).device)
        cur.wait_event(self._done)
        for o in self._out.values():
            o.record_stream(cur)  # allocated on the lane's stream, consumed on this one
        return self._out

    def synchronize(self) -> Dict:
        self._done.synchronize()
        return self._out

    def __del__(self):
        # the event goes back to its lane's pool instead of being destroyed: destroying an event whose work is still in flight made the host wait
        # for it on this platform (the loop then kept one clip fewer in flight than asked for)
        try:
            self._pool.append(self._done)
        except Exception:
            pass


class ClipsInFlight:
    """Consecutive clips of an inference stream are independent, so clip k+1's encoder can start while clip k is still in its
    DPT head: ``depth`` engine contexts of one model (own workspaces, the same bound parameters) take the clips round-robin, each on its own HIP
    stream.  What that buys is what a short clip leaves idle on 256 CUs -- launch ramps, tails, the head's small grids: ViT-S 518x518 T=8
    +5.4 % with 2 and +8.9 % with 3 clips in flight, every output bit-identical to the one-at-a-time result; at ViT-B T=16 the kernels fill the
    part alone and co-residency costs 1-2 % (profiles/r03_notes.txt), hence ``auto_depth``.  Memory: one workspace per lane (1.6 GB at ViT-S T=8),
    released by ``close()``.  The lanes are this object's own: an engine context is single-user (workspace, stream-K arrival counters), so neither
    ``model(x)`` (lane 0) nor another ClipsInFlight ever shares one, and they may all run beside each other.
    Full overlap needs the clips to come from a stream other than PyTorch's default one (or ``resident=True``): an event recorded on the default
    (null) stream orders it against every other stream on this platform, which keeps one clip fewer in flight (measured: no gain at depth 3).

        flight = ClipsInFlight(model, device)            # depth: auto_depth(model, T) on the first clip
        handles = [flight.submit(x) for x in clips]      # returns at once; x must stay unmodified until the result is taken
        maps = handles[0].result()                       # the current stream now waits for clip 0 only

    The reference has no counterpart (one synchronous forward per clip, evaluate_depth_video.py:163-168)."""

    def __init__(self, model, device, depth: Optional[int] = None):
        self.model, self.dev = model, torch.device(device)
        self.depth = None if depth is None else max(1, int(depth))
        self.streams: List[torch.cuda.Stream] = []
        self.lanes: List[int] = []                      # this object's own engine contexts of the model (never lane 0, never shared)
        self._ready: List[torch.cuda.Event] = []        # one per lane, re-recorded by every submit
        self._pool: List[List[torch.cuda.Event]] = []   # completion events per lane, recycled by InFlight.__del__
        self._k = 0

    def close(self) -> None:
        """Wait for the clips in flight and release the lanes' engine contexts (their workspaces)."""
        for s in self.streams:
            s.synchronize()
        for lane in self.lanes:
            self.model._drop_lane(lane)
        self.lanes, self.streams, self._ready, self._pool = [], [], [], []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def auto_depth(model, frames: int) -> int:
        """Lanes while one clip is too small to fill the part (token rows x width of the encoder), 1 otherwise -- the measured sign change lies
        at about ViT-S T=16 / ViT-B T=8 (21 920 rows x 384 resp. 10 960 x 768: +3 % / +-0); larger clips lose 1-2 %: profiles/r03_notes.txt."""
        ph, pw = model.image_shape[0] // 14, model.image_shape[1] // 14
        rows = frames * (ph * pw + 1)
        return 3 if rows * model.pretrained.embed_dim <= 22_000 * 384 else 1

    def next_lane(self, frames: int):
        """(stream, lane id) for the next clip of ``frames`` frames, round-robin: for pipelines that put more than the forward on the lane's stream
        (``video.HipWindowRunner``: uint8 -> float, pre-resize, forward, resize back).  The caller runs ``model(x, lane=lane)`` under
        ``torch.cuda.stream(stream)`` and orders its own inputs / outputs with events."""
        if self.depth is None:
            self.depth = self.auto_depth(self.model, frames)
        while len(self.streams) < self.depth:
            self.streams.append(torch.cuda.Stream(device=self.dev))
            self.lanes.append(self.model._new_lane())
            self._ready.append(torch.cuda.Event())
            self._pool.append([])
        k = self._k % self.depth
        self._k += 1
        return self.streams[k], self.lanes[k]

    def submit(self, x: torch.Tensor, resident: bool = False) -> InFlight:
        """Enqueue one clip [B, T, 3, H, W] (device tensor).  ``resident``: the clip's producer finished long ago (a dataset tensor already in HBM),
        so the lane need not wait for the caller's stream."""
        if self.depth is None:
            self.depth = self.auto_depth(self.model, x.shape[0] * x.shape[1])
        while len(self.streams) < self.depth:
            self.streams.append(torch.cuda.Stream(device=self.dev))
            self.lanes.append(self.model._new_lane())
            self._ready.append(torch.cuda.Event())
            self._pool.append([])
        lane = self._k % self.depth
        self._k += 1
        s = self.streams[lane]
        with torch.cuda.device(self.dev), torch.no_grad():
            if not resident:
                self._ready[lane].record(torch.cuda.current_stream(self.dev))
                s.wait_event(self._ready[lane])  # the clip was produced on the caller's stream
            with torch.cuda.stream(s):
                out = self.model(x, lane=self.lanes[lane])  # the lane's own context and workspace; its previous clip is ordered before this one by the stream
                done = self._pool[lane].pop() if self._pool[lane] else torch.cuda.Event()
                done.record(s)
            x.record_stream(s)
        return InFlight(out, done, self._pool[lane])

    def run(self, clips, resident: bool = False) -> Iterator[Dict]:
        """Outputs of ``clips`` (device tensors) in order, ``depth`` of them in flight."""
        pending: List[InFlight] = []
        for x in clips:
            pending.append(self.submit(x, resident))
            if len(pending) >= self.depth:
                yield pending.pop(0).result()
        while pending:
            yield pending.pop(0).result()


class ClipPipeline:
    """``run(clips)``: feed host clips [B, T, 3, H, W] (float32, any iterable) through ``forward`` and yield, in order, the maps of
    each clip as pinned host tensors (valid until the next result is taken).  ``forward`` is the model (or any callable
    mapping a device clip to a dict / sequence of device tensors); it runs on the current stream of ``device``."""

    def __init__(self, forward: Callable, device: torch.device, depth: int = 2, copy_streams: int = 2):
        self.forward, self.dev, self.depth = forward, torch.device(device), max(2, int(depth))
        self._d_in: List[Optional[torch.Tensor]] = [None] * self.depth
        self._h_out: List[Optional[List[torch.Tensor]]] = [None] * self.depth
        # copy_streams: 2 = uploads and downloads on separate streams (default; measured 10.90 ms/clip against 10.55 resident), 1 = one
        # shared copy stream (11.37), 0 = downloads on the compute stream itself (10.90)   (profiles/r02_pcie_pipeline.txt)
        self.s_in = torch.cuda.Stream(device=self.dev)
        self.s_out = torch.cuda.Stream(device=self.dev) if copy_streams >= 2 else (self.s_in if copy_streams == 1 else None)

    def _stage_in(self, slot: int, clip: torch.Tensor, up_done, used) -> None:
        if self._d_in[slot] is None or self._d_in[slot].shape != clip.shape:
            self._d_in[slot] = torch.empty(clip.shape, dtype=torch.float32, device=self.dev)
        # A pinned clip is uploaded asynchronously from where it lies.  A pageable one goes up with a blocking copy on the copy stream (the
        # driver stages it): the host waits ~2 ms, the GPU keeps computing the previous clip.  Staging it through a pinned buffer by hand is
        # far slower -- host writes into pinned memory run at ~2 GB/s on this platform (profiles/r02_pcie_pipeline.txt).
        with torch.cuda.stream(self.s_in):
            if used[slot] is not None:
                self.s_in.wait_event(used[slot])  # the forward that read the device buffer has finished
            self._d_in[slot].copy_(clip, non_blocking=clip.is_pinned())
            up_done[slot] = torch.cuda.Event()
            up_done[slot].record(self.s_in)

    def run(self, clips) -> Iterator[List[torch.Tensor]]:
        # (a generator: no context manager may stay entered across a yield -- torch.no_grad / torch.cuda.device are thread state and
        # would leak into the consumer's code -- so each step enters them itself)
        dev, D = self.dev, self.depth
        up_done: List[Optional[torch.cuda.Event]] = [None] * D
        used: List[Optional[torch.cuda.Event]] = [None] * D
        out_done: List[Optional[torch.cuda.Event]] = [None] * D
        it = iter(clips)
        nxt = next(it, None)
        if nxt is None:
            return
        with torch.cuda.device(dev):
            self._stage_in(0, nxt, up_done, used)
        k = 0
        pending: List[int] = []
        while nxt is not None:
            slot = k % D
            nxt = next(it, None)
            ready = None
            with torch.cuda.device(dev), torch.no_grad():
                compute = torch.cuda.current_stream(dev)
                if nxt is not None:
                    self._stage_in((k + 1) % D, nxt, up_done, used)  # overlaps clip k's forward
                compute.wait_event(up_done[slot])
                out = self.forward(self._d_in[slot])
                maps = list(out.values()) if isinstance(out, dict) else list(out)
                used[slot] = torch.cuda.Event()
                used[slot].record(compute)
                if out_done[slot] is not None:  # the slot's previous result is handed out before its buffers are overwritten
                    out_done[slot].synchronize()
                    pending.remove(slot)
                    ready = self._h_out[slot]
            if ready is not None:
                yield ready  # valid until the consumer asks for the next result (the D2H of clip k is enqueued after this yield)
            with torch.cuda.device(dev), torch.no_grad():
                if self._h_out[slot] is None or any(h.shape != m.shape for h, m in zip(self._h_out[slot], maps)):
                    self._h_out[slot] = [torch.empty(m.shape, dtype=m.dtype).pin_memory() for m in maps]
                s_out = self.s_out if self.s_out is not None else torch.cuda.current_stream(dev)
                with torch.cuda.stream(s_out):
                    if self.s_out is not None:
                        s_out.wait_event(used[slot])
                    for h, m in zip(self._h_out[slot], maps):
                        h.copy_(m, non_blocking=True)
                        if self.s_out is not None:
                            m.record_stream(s_out)
                    out_done[slot] = torch.cuda.Event()
                    out_done[slot].record(s_out)
            pending.append(slot)
            k += 1
        for slot in list(pending):  # oldest first
            out_done[slot].synchronize()
            out_done[slot] = None
            yield self._h_out[slot]
