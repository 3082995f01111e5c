"""Several whole batches in flight on ONE GPU (HIP streams; no reference counterpart — the reference's ``forward`` is synchronous).

Why.  With adaptive depth / width most pairs of a batch end early (``check_if_stop``, ref lightglue.py:645-656), and the launches of the late
layers then fill only a fraction of the chip: at BASELINE config 3 (16 pairs of N = M = 2 048) four pairs are alive behind layer 3, one round of
workgroups on a quarter of the CUs.  A second batch's early layers, issued on another HIP stream, run in that space: measured +3 … +4 % at config
3 and +6 % at config 5 with two or three batches in flight, −1 % at config 2 and ±0 at config 4, where every launch already fills the chip
(``tools/ab_inflight.py``, ``profiles/r06t_ab_inflight.log``).
A stream of single pairs (B = 1 per call) gains most — x2.5 ... x3.5 with four lanes at N = 512 / 1 024 (``profiles/r06v_ab_inflight_b1.log``); beyond four lanes the runtime's hardware
queues are shared and the rate falls again.

How.  ``InflightMatcher(model, depth)`` holds ``depth`` LANES: a lane = one ``LightGlue`` instance (its own engine, i.e. its own workspace — the
first lane is ``model`` itself, the others are deep copies made at construction) + one HIP stream.  ``submit(data)`` deals batches to the lanes round
robin: the lane's stream first waits for the caller's current stream (the producer of ``data``), then the forward is enqueued there WITHOUT a
host synchronisation (``LightGlue.forward_deferred``).  The returned handle's ``result()`` waits for that forward only and returns the same dict as
``LightGlue.forward``; the output tensors are handed to the stream that is current when ``result()`` is called (``record_stream``), so the caching
allocator does not recycle them under a consumer on another stream.  Two forwards never share a workspace: a lane's next forward is ordered
behind its previous one by its stream.  Weights are copied when the lanes are built; after changing the model's parameters call ``refresh()``.
"""
from __future__ import annotations

import copy
from typing import List

import torch

from .lightglue import DeferredMatches, LightGlue


def _tensors(obj):
    if torch.is_tensor(obj):
        yield obj
    elif isinstance(obj, dict):
        for v in obj.values():
            yield from _tensors(v)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            yield from _tensors(v)


class InflightResult:
    """Handle of one submitted batch."""

    def __init__(self, deferred: DeferredMatches, data, lane: int):
        self._deferred, self._data, self.lane, self._out = deferred, data, lane, None

    def result(self) -> dict:
        if self._out is None:
            buffers = self._deferred.buffers
            out = self._deferred.result()            # waits for THIS forward's event; raises on a non-zero status like forward()
            cur = torch.cuda.current_stream()
            for t in buffers:                        # every output tensor is a view of one of these allocations
                t.record_stream(cur)
            self._out, self._deferred, self._data = out, None, None   # the inputs may go now
        return self._out


class InflightMatcher:
    def __init__(self, model: LightGlue, depth: int = 2, device=None):
        if depth < 1:
            raise ValueError("depth must be >= 1")
        self.device = torch.device(device if device is not None else "cuda")
        if self.device.type != "cuda":
            raise RuntimeError("lightglue_amd runs on MI355X (ROCm device type 'cuda') only; there is no CPU fallback")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.model = model
        self.depth = depth
        self._lanes: List[LightGlue] = []
        self._streams = [torch.cuda.Stream(self.device) for _ in range(depth)]
        self._next = 0
        self.refresh()

    def refresh(self) -> None:
        """(Re)build the lanes from the model's current parameters and options."""
        torch.cuda.synchronize(self.device)
        self._lanes = [self.model] + [copy.deepcopy(self.model).eval() for _ in range(self.depth - 1)]
        for lane in self._lanes[1:]:
            lane.track_inplace_weight_edits = False   # private copies: nobody edits their parameters, so the per-forward walk over 251 version counters (10 - 30 us) is skipped

    def reserve(self, batch: int, n0: int, n1: int) -> None:
        """Pre-size every lane's workspace (avoids a synchronising re-allocation inside the first forwards)."""
        for lane in self._lanes:
            lane.reserve(batch, n0, n1, self.device)

    def submit(self, data: dict) -> InflightResult:
        k = self._next
        self._next = (k + 1) % self.depth
        stream = self._streams[k]
        stream.wait_stream(torch.cuda.current_stream(self.device))     # `data` was produced on the caller's stream
        for t in _tensors(data):
            if t.is_cuda:
                t.record_stream(stream)
        with torch.cuda.stream(stream):
            deferred = self._lanes[k].forward_deferred(data)
        return InflightResult(deferred, data, k)

    def map(self, batches):
        """Results of an iterable of batches, in order, with up to `depth` of them in flight."""
        pending: List[InflightResult] = []
        for data in batches:
            pending.append(self.submit(data))
            if len(pending) == self.depth:
                yield pending.pop(0).result()
        for p in pending:
            yield p.result()
