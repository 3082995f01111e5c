"""Pair-sharded data parallelism over the GPUs of one node.

Image pairs are fully independent (no cross-pair op anywhere in the reference's ``_forward``), so the
batch shards by contiguous blocks of pairs with NO data-path collective; one process per GPU
(``torchrun``), weights replicated.  The only exchange is the result gather: ONE fixed-shape
``all_gather_into_tensor`` of a packed int32 buffer per batch (match indices; scores are carried as
their fp32 bit patterns in the same buffer) — RCCL over xGMI on the GPU (backend "nccl"), gloo in
the CPU tests.  Payload is a few MB at most, i.e. latency-bound on the xGMI ring; it is issued on the
side stream right after the local forward (an event orders it behind the compute stream), so that the caller can start the
next batch while the gather is on the wire (`issue_local` / `Pending.wait`; `forward` waits at once).  The HIP matcher hands
over its int32 / fp32 output buffers as they are (`LightGlue.forward_raw`): no int64 round trip, no host synchronisation
before the collective.

The ragged ``matches`` lists are rebuilt from ``matches0`` after the gather (no variable-size
collective).

Ragged batches (``num_keypoints`` per image) have unequal cost per pair, so for them the pairs are dealt to
the ranks by estimated work (``balanced_shards``, SURVEY.md §8e "balance by expected work") instead of by
contiguous blocks; the gather is the same single collective, followed by a row permutation.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_range(batch: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block of pairs owned by ``rank``: sizes differ by at most one."""
    base, rem = divmod(batch, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def pair_cost(n0: int, n1: int, d: int = 256) -> float:
    """Algorithmic FLOPs of one transformer layer for a pair with n0 / n1 keypoints: linear layers
    2 490 368 per point, self-attention 4·d·n² per image, cross-attention 6·d·n0·n1."""
    return 2490368.0 * (n0 + n1) + 4.0 * d * (n0 * n0 + n1 * n1) + 6.0 * d * n0 * n1


def balanced_shards(costs: Sequence[float], world: int) -> List[List[int]]:
    """Deal pair indices to ``world`` ranks by longest-processing-time-first: heaviest remaining pair to the
    currently lightest rank (ties: lower rank / lower index), with every rank holding ceil(B / world) pairs at
    most so the gather buffer keeps its fixed shape.  Deterministic, identical on every rank."""
    cap = (len(costs) + world - 1) // world
    order = sorted(range(len(costs)), key=lambda i: (-float(costs[i]), i))
    load, shards = [0.0] * world, [[] for _ in range(world)]
    for i in order:
        r = min((r for r in range(world) if len(shards[r]) < cap), key=lambda r: (load[r], r))
        shards[r].append(i)
        load[r] += float(costs[i])
    return [sorted(sh) for sh in shards]


def _take(data: dict, idx) -> dict:
    """Rows `idx` (slice or index tensor) of every per-pair tensor of the nested input dict."""
    pick = lambda t: t[idx] if isinstance(idx, slice) else t.index_select(0, idx.to(t.device))
    return {k: ({kk: pick(vv) for kk, vv in v.items()} if isinstance(v, dict) else v) for k, v in data.items()}


class PairShardedMatcher:
    """Wraps a per-process matcher (``LightGlue`` on this rank's GPU, or any callable with the same dict
    contract) and returns full-batch results on every rank.

    ``forward(data)``: ``data`` holds the FULL batch on every rank (the usual case when a loader feeds
    identical manifests) — each rank matches its shard and the results are all-gathered;
    ``forward_local(local_data, global_batch)`` takes the already-sharded local pairs.
    """

    def __init__(self, matcher: Callable[[dict], dict], group=None, always_gather: bool = False):
        self.matcher = matcher
        self.group = group
        self.always_gather = always_gather   # issue the collective even in a world of one (bench.py measures its cost against no gather that way)
        self._src_cache: dict = {}
        self._side = None      # side stream of the result gather (created on first use on a GPU)

    @property
    def world(self) -> int:
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    @property
    def rank(self) -> int:
        return dist.get_rank(self.group) if dist.is_initialized() else 0

    def assignment(self, data: dict, balance: Optional[bool] = None) -> List[List[int]]:
        """Pair indices per rank: contiguous blocks, or work-balanced when the batch is ragged (default) / on request."""
        batch = data["image0"]["keypoints"].shape[0]
        num0, num1 = data["image0"].get("num_keypoints"), data["image1"].get("num_keypoints")
        if balance is None:
            balance = num0 is not None or num1 is not None
        if not balance or self.world == 1:
            return [list(range(*shard_range(batch, r, self.world))) for r in range(self.world)]
        full0, full1 = data["image0"]["keypoints"].shape[1], data["image1"]["keypoints"].shape[1]
        n0 = [full0] * batch if num0 is None else [int(v) for v in torch.as_tensor(num0).tolist()]
        n1 = [full1] * batch if num1 is None else [int(v) for v in torch.as_tensor(num1).tolist()]
        return balanced_shards([pair_cost(a, b) for a, b in zip(n0, n1)], self.world)

    def forward(self, data: dict, balance: Optional[bool] = None) -> Dict[str, torch.Tensor]:
        batch = data["image0"]["keypoints"].shape[0]
        shards = self.assignment(data, balance)
        mine = shards[self.rank]
        contiguous = all(sh == list(range(sh[0], sh[0] + len(sh))) for sh in shards if sh)
        idx = slice(mine[0], mine[-1] + 1) if (mine and contiguous) else torch.tensor(mine, dtype=torch.long)
        return self.forward_local(_take(data, idx), batch, shards)

    __call__ = forward

    def _row_source(self, shards: List[List[int]], per_rank: int, global_batch: int, device):
        """(gathered-buffer row of every pair of the batch, destination pair of every gathered row or -1 for the padding rows of short shards):
        rank r's k-th row is pair shards[r][k]; built once per shard assignment and kept on the device"""
        key = (tuple(tuple(sh) for sh in shards), per_rank, str(device))
        hit = self._src_cache.get(key)
        if hit is None:
            src = torch.empty(global_batch, dtype=torch.long)
            dest = torch.full((len(shards) * per_rank,), -1, dtype=torch.int32)
            for r, sh in enumerate(shards):
                src[torch.tensor(sh, dtype=torch.long)] = r * per_rank + torch.arange(len(sh))
                dest[r * per_rank: r * per_rank + len(sh)] = torch.tensor(sh, dtype=torch.int32)
            if len(self._src_cache) > 64:
                self._src_cache.clear()
            hit = self._src_cache[key] = (src.to(device), dest.to(device))
        return hit

    def issue_local(self, local: dict, global_batch: int, shards: Optional[List[List[int]]] = None) -> "Pending":
        """Match the local shard and put the result gather in flight; `Pending.wait()` returns the full-batch dict."""
        m = local["image0"]["keypoints"].shape[1]
        n = local["image1"]["keypoints"].shape[1]
        world, rank = self.world, self.rank
        if shards is None:
            shards = [list(range(*shard_range(global_batch, r, world))) for r in range(world)]
        nloc = len(shards[rank])
        assert local["image0"]["keypoints"].shape[0] == nloc, "local shard size does not match the pair assignment"
        dev = local["image0"]["keypoints"].device
        raw = getattr(self.matcher, "forward_raw", None)
        # ---- one row per pair, [pairs_max][2m + 2n + 1] int32: matches0 | bits(scores0) | matches1 | bits(scores1) | stop.  The HIP matcher's last
        # kernel packs the rows itself (lg_forward_io.wire): no framework kernel between the forward and the collective.
        per_rank = (global_batch + world - 1) // world
        width = 2 * m + 2 * n + 1
        buf = torch.empty((per_rank, width), dtype=torch.int32, device=dev)
        if nloc < per_rank:
            buf[nloc:].zero_()
        if nloc > 0 and raw is not None and dev.type == "cuda":
            raw(local, wire=buf)
        elif nloc > 0:   # any other matcher with the dict contract (the CPU tests' stand-in): pack here
            out = self.matcher(local)
            stop = out["stop"]
            stop_t = torch.full((nloc,), int(stop), dtype=torch.int32, device=dev) if not torch.is_tensor(stop) else stop.to(dev, torch.int32).reshape(nloc)
            as_i32 = lambda t: t if t.dtype is torch.int32 else t.to(torch.int32)
            buf[:nloc, 0:m] = as_i32(out["matches0"])
            buf[:nloc, m:2 * m] = out["matching_scores0"].to(torch.float32).contiguous().view(torch.int32)
            buf[:nloc, 2 * m:2 * m + n] = as_i32(out["matches1"])
            buf[:nloc, 2 * m + n:2 * m + 2 * n] = out["matching_scores1"].to(torch.float32).contiguous().view(torch.int32)
            buf[:nloc, -1] = stop_t
        if world == 1 and not (self.always_gather and dist.is_initialized()):
            return Pending(self, buf[:nloc], None, None, m, n)
        # RCCL ("nccl") gathers device buffers directly; gloo (CPU tests, or a debugging run of several ranks
        # on one GPU) goes through host copies
        via_host = dist.get_backend(self.group) == "gloo" and buf.is_cuda
        src, dest = self._row_source(shards, per_rank, global_batch, dev)
        if buf.is_cuda and not via_host:
            if self._side is None:
                self._side = torch.cuda.Stream(device=dev)
            ready = torch.cuda.Event(); ready.record(torch.cuda.current_stream(dev))
            gathered = torch.empty((world * per_rank, width), dtype=torch.int32, device=dev)
            with torch.cuda.stream(self._side):
                self._side.wait_event(ready)
                dist.all_gather_into_tensor(gathered, buf, group=self.group)
                done = torch.cuda.Event(); done.record(self._side)
            buf.record_stream(self._side); gathered.record_stream(self._side)
            return Pending(self, gathered, src, done, m, n, dest)
        send = buf.cpu() if via_host else buf
        gathered = torch.empty((world * per_rank, width), dtype=torch.int32, device=send.device)
        dist.all_gather_into_tensor(gathered, send, group=self.group)
        if via_host:
            gathered = gathered.to(dev)
        return Pending(self, gathered, src, None, m, n, dest)

    def forward_local(self, local: dict, global_batch: int, shards: Optional[List[List[int]]] = None) -> Dict[str, torch.Tensor]:
        return self.issue_local(local, global_batch, shards).wait()

    @staticmethod
    def ragged(result: Dict[str, torch.Tensor]):
        """Rebuild the reference's ragged `matches` / `scores` lists (ref lightglue.py:593-602) from the
        gathered fixed-shape tensors."""
        matches, scores = [], []
        for k in range(result["matches0"].shape[0]):
            valid = result["matches0"][k] > -1
            i0 = torch.where(valid)[0]
            matches.append(torch.stack([i0, result["matches0"][k][valid]], -1))
            scores.append(result["matching_scores0"][k][valid])
        return matches, scores


class Pending:
    """A result gather in flight.  `wait()` orders the current stream behind it and unpacks the full-batch tensors."""

    def __init__(self, owner: PairShardedMatcher, gathered: torch.Tensor, src, done, m: int, n: int, dest=None):
        self.owner, self.gathered, self.src, self.done, self.m, self.n, self.dest = owner, gathered, src, done, m, n, dest

    def wait(self) -> Dict[str, torch.Tensor]:
        if self.done is not None:
            torch.cuda.current_stream(self.gathered.device).wait_event(self.done)
        m, n, g = self.m, self.n, self.gathered
        if g.is_cuda:   # ONE engine kernel: row permutation, int64 widening, score bit patterns back to fp32 (lg_unpack_wire, include/lightglue_amd.h)
            import ctypes as C
            from . import _cabi
            rows = g.shape[0] if self.src is None else self.src.shape[0]
            new = lambda shape, dt: torch.empty(shape, dtype=dt, device=g.device)
            out = {"matches0": new((rows, m), torch.int64), "matches1": new((rows, n), torch.int64), "matching_scores0": new((rows, m), torch.float32),
                   "matching_scores1": new((rows, n), torch.float32), "stop": new((rows,), torch.int64)}
            ptr = lambda t: None if t is None or t.numel() == 0 else t.data_ptr()
            with torch.cuda.device(g.device):
                _cabi.check(_cabi.load().lg_unpack_wire(ptr(g), g.stride(0), g.shape[0], m, n, ptr(self.dest) if self.src is not None else None,
                                                        ptr(out["matches0"]), ptr(out["matching_scores0"]), ptr(out["matches1"]), ptr(out["matching_scores1"]),
                                                        ptr(out["stop"]), C.c_void_p(torch.cuda.current_stream(g.device).cuda_stream)))
            return out
        full = g if self.src is None else g.index_select(0, self.src)
        return {"matches0": full[:, 0:m].long(), "matches1": full[:, 2 * m:2 * m + n].long(),
                "matching_scores0": full[:, m:2 * m].contiguous().view(torch.float32),
                "matching_scores1": full[:, 2 * m + n:2 * m + 2 * n].contiguous().view(torch.float32),
                "stop": full[:, -1].long()}
