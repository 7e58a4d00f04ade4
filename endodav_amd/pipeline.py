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
