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
        """Match the local shard and put the result gather in flight; `Pending.wait()` returns the full-batch dict — the SAME dict `LightGlue.forward` returns
        (ref lightglue.py:604-629: int64 matches0/1, scores, `stop`, the ragged `matches` / `scores` lists, prune0/1), on every rank."""
        m = local["image0"]["keypoints"].shape[1]
        n = local["image1"]["keypoints"].shape[1]
        world, rank = self.world, self.rank
        if shards is None:
            shards = [list(range(*shard_range(global_batch, r, world))) for r in range(world)]
        nloc = len(shards[rank])
        assert local["image0"]["keypoints"].shape[0] == nloc, "local shard size does not match the pair assignment"
        dev = local["image0"]["keypoints"].device
        raw = getattr(self.matcher, "forward_raw", None)
        # ---- one row per pair, [pairs_max][3m + 3n + 2] int32: matches0 | bits(scores0) | matches1 | bits(scores1) | stop | status | prune0 | prune1
        # (include/lightglue_amd.h lg_forward_io.wire).  The HIP matcher's last kernel packs the rows itself: no framework kernel between the forward and
        # the collective.  The prune block holds int counters when the matcher prunes, else the bit patterns of the reference's float fill; which of the two
        # is a property of the matcher's configuration, identical on every rank (weights and conf are replicated).
        per_rank = (global_batch + world - 1) // world
        width = wire_width(m, n)
        buf = torch.empty((per_rank, width), dtype=torch.int32, device=dev)
        if nloc < per_rank:
            buf[nloc:].zero_()
        will_prune = getattr(self.matcher, "will_prune", None)
        with_prune = bool(will_prune(m, n)) if will_prune is not None else bool(getattr(self.matcher, "wire_prunes", False))
        if nloc > 0 and raw is not None and dev.type == "cuda":
            assert bool(raw(local, wire=buf)["pruning"]) == with_prune
        elif nloc > 0:   # any other matcher with the dict contract (the CPU tests' stand-in): pack here
            out = self.matcher(local)
            _pack_rows(buf[:nloc], out, m, n)
            with_prune = "prune0" in out and not out["prune0"].dtype.is_floating_point
        if world == 1 and not (self.always_gather and dist.is_initialized()):
            outs, info, pairs = _unpack(buf[:nloc], None, nloc, m, n, with_prune)
            if not buf.is_cuda:
                return Pending(outs, info, pairs, None)
            host = torch.empty((3, pairs), dtype=torch.int32, pin_memory=True)     # as LightGlue.forward_deferred: the sizes travel behind an event
            host.copy_(info, non_blocking=True)
            done = torch.cuda.Event(); done.record(torch.cuda.current_stream(dev))
            return Pending(outs, host, pairs, done)
        # RCCL ("nccl") gathers device buffers directly; gloo (CPU tests, or a debugging run of several ranks
        # on one GPU) goes through host copies
        via_host = dist.get_backend(self.group) == "gloo" and buf.is_cuda
        _, dest = self._row_source(shards, per_rank, global_batch, dev)
        if buf.is_cuda and not via_host:
            # the gather, its unpack kernel and the copy of the [3][B] host block all run on the SIDE stream behind an event, so that the caller's next forward
            # (already enqueued on the compute stream when wait() is called) never stands between the host and the sizes it waits for
            if self._side is None:
                self._side = torch.cuda.Stream(device=dev)
            cur = torch.cuda.current_stream(dev)
            # everything the side stream writes is allocated (torch.empty: no kernel) BEFORE the event, and nothing on the compute stream touches it afterwards
            gathered = torch.empty((world * per_rank, width), dtype=torch.int32, device=dev)
            outs, info = _alloc_outputs(global_batch, m, n, with_prune, dev)
            ready = torch.cuda.Event(); ready.record(cur)
            with torch.cuda.stream(self._side):
                self._side.wait_event(ready)
                dist.all_gather_into_tensor(gathered, buf, group=self.group)
                _unpack_cuda(gathered, dest, global_batch, m, n, with_prune, outs, info, self._side)
                host = torch.empty((3, global_batch), dtype=torch.int32, pin_memory=True)
                host.copy_(info, non_blocking=True)
                done = torch.cuda.Event(); done.record(self._side)
            for t in (buf, gathered, info, *[v for v in outs.values() if torch.is_tensor(v)]):
                t.record_stream(self._side)
            return Pending(outs, host, global_batch, done)
        send = buf.cpu() if via_host else buf
        gathered = torch.empty((world * per_rank, width), dtype=torch.int32, device=send.device)
        dist.all_gather_into_tensor(gathered, send, group=self.group)
        if via_host:
            gathered = gathered.to(dev)
        return Pending(*_unpack(gathered, dest, global_batch, m, n, with_prune), None)

    def forward_local(self, local: dict, global_batch: int, shards: Optional[List[List[int]]] = None) -> Dict[str, torch.Tensor]:
        return self.issue_local(local, global_batch, shards).wait()

    @staticmethod
    def ragged(result: Dict[str, torch.Tensor]):
        """The reference's ragged `matches` / `scores` lists (ref lightglue.py:593-602) from fixed-shape `matches0` / `matching_scores0`: B `torch.where`
        calls, i.e. B host synchronisations on a GPU — what `Pending.wait()` avoids (its lists come from lg_unpack_wire's match list + ONE host copy);
        kept as the independent restatement the tests compare those lists with."""
        matches, scores = [], []
        for k in range(result["matches0"].shape[0]):
            valid = result["matches0"][k] > -1
            i0 = torch.where(valid)[0]
            matches.append(torch.stack([i0, result["matches0"][k][valid]], -1))
            scores.append(result["matching_scores0"][k][valid])
        return matches, scores


def wire_width(m: int, n: int) -> int:
    """int32 elements of one wire row (LG_WIRE_WIDTH, include/lightglue_amd.h)"""
    return 3 * m + 3 * n + 2


def _pack_rows(rows: torch.Tensor, out: dict, m: int, n: int) -> None:
    """Generic (framework-op) form of the wire row for matchers without `forward_raw`: same layout as the engine's write_outputs_kernel."""
    k, dev = rows.shape[0], rows.device
    stop = out["stop"]
    stop_t = torch.full((k,), int(stop), dtype=torch.int32, device=dev) if not torch.is_tensor(stop) else stop.to(dev, torch.int32).reshape(k)
    as_i32 = lambda t: t if t.dtype is torch.int32 else t.to(torch.int32)
    bits = lambda t: t.to(torch.float32).contiguous().view(torch.int32)
    rows[:, 0:m] = as_i32(out["matches0"]); rows[:, m:2 * m] = bits(out["matching_scores0"])
    rows[:, 2 * m:2 * m + n] = as_i32(out["matches1"]); rows[:, 2 * m + n:2 * m + 2 * n] = bits(out["matching_scores1"])
    wp = 2 * m + 2 * n + 2
    rows[:, wp - 2] = stop_t
    status = out.get("status")
    rows[:, wp - 1] = 0 if status is None else as_i32(torch.as_tensor(status)).to(dev).reshape(k)
    for key, lo, cnt in (("prune0", wp, m), ("prune1", wp + m, n)):
        p = out.get(key)
        if p is None:
            rows[:, lo:lo + cnt] = 0
        else:
            rows[:, lo:lo + cnt] = bits(p) if p.dtype.is_floating_point else as_i32(p)


def _alloc_outputs(pairs: int, m: int, n: int, with_prune: bool, dev):
    new = lambda shape, dt: torch.empty(shape, dtype=dt, device=dev)
    kmax = min(m, n)
    pdt = torch.int64 if with_prune else torch.float32
    outs = {"matches0": new((pairs, m), torch.int64), "matches1": new((pairs, n), torch.int64),
            "matching_scores0": new((pairs, m), torch.float32), "matching_scores1": new((pairs, n), torch.float32),
            "stop": new((pairs,), torch.int64), "prune0": new((pairs, m), pdt), "prune1": new((pairs, n), pdt),
            "_mlist": new((pairs, kmax, 2), torch.int64), "_mscores": new((pairs, kmax), torch.float32)}
    # the host block: NOT torch.zeros — a fill kernel on the compute stream could run after the side stream's unpack kernel has written the block; every pair of
    # the batch sits in exactly one shard, so lg_unpack_wire (and the CPU form below) writes all 3 x pairs entries
    return outs, torch.empty((3, pairs), dtype=torch.int32, device=dev)


def _unpack_cuda(g: torch.Tensor, dest, pairs: int, m: int, n: int, with_prune: bool, outs: dict, info: torch.Tensor, stream) -> None:
    """ONE engine kernel on `stream`: row permutation, int64 widening, score / fill bit patterns back to fp32, the sorted match list and the host block
    (lg_unpack_wire, include/lightglue_amd.h)"""
    import ctypes as C
    from . import _cabi
    ptr = lambda t: None if t is None or t.numel() == 0 else t.data_ptr()
    io = _cabi.LgUnpackIO(ptr(g), g.stride(0) if g.dim() == 2 else 0, g.shape[0], m, n, int(with_prune), pairs, ptr(dest),
                          ptr(outs["matches0"]), ptr(outs["matches1"]), ptr(outs["stop"]), ptr(outs["matching_scores0"]), ptr(outs["matching_scores1"]),
                          ptr(outs["prune0"]) if with_prune else None, ptr(outs["prune1"]) if with_prune else None,
                          None if with_prune else ptr(outs["prune0"]), None if with_prune else ptr(outs["prune1"]),
                          ptr(outs["_mlist"]), ptr(outs["_mscores"]), ptr(info))
    with torch.cuda.device(g.device):
        _cabi.check(_cabi.load().lg_unpack_wire(C.byref(io), C.c_void_p(stream.cuda_stream)))


def _unpack(g: torch.Tensor, dest, pairs: int, m: int, n: int, with_prune: bool):
    """(outputs, [3][pairs] info block, pairs) of gathered rows `g` on the CURRENT stream / on the CPU."""
    outs, info = _alloc_outputs(pairs, m, n, with_prune, g.device)
    if g.is_cuda:
        _unpack_cuda(g, dest, pairs, m, n, with_prune, outs, info, torch.cuda.current_stream(g.device))
        return outs, info, pairs
    # CPU (gloo tests): the same unpack with framework ops
    if dest is not None:
        keep = dest >= 0
        g, d = g[keep], dest[keep].long()
    else:
        d = torch.arange(g.shape[0])
    wp = 2 * m + 2 * n + 2
    fl = lambda t: t.contiguous().view(torch.float32)
    outs["matches0"][d] = g[:, 0:m].long(); outs["matching_scores0"][d] = fl(g[:, m:2 * m])
    outs["matches1"][d] = g[:, 2 * m:2 * m + n].long(); outs["matching_scores1"][d] = fl(g[:, 2 * m + n:2 * m + 2 * n])
    outs["stop"][d] = g[:, wp - 2].long()
    outs["prune0"][d] = g[:, wp:wp + m].long() if with_prune else fl(g[:, wp:wp + m])
    outs["prune1"][d] = g[:, wp + m:wp + m + n].long() if with_prune else fl(g[:, wp + m:wp + m + n])
    info[0, d] = g[:, wp - 2]; info[2, d] = g[:, wp - 1]
    for k in d.tolist():
        valid = outs["matches0"][k] > -1
        c = int(valid.sum())
        outs["_mlist"][k, :c, 0] = torch.where(valid)[0]; outs["_mlist"][k, :c, 1] = outs["matches0"][k][valid]
        outs["_mscores"][k, :c] = outs["matching_scores0"][k][valid]
        info[1, k] = c
    return outs, info, pairs


class Pending:
    """A result gather in flight.  `wait()` returns the full-batch output dict of `LightGlue.forward` (same keys, dtypes and list semantics) and raises — on
    EVERY rank — if any pair of any rank carries a non-zero status (LG_ERR_RANGE / LG_ERR_DEVICE)."""

    def __init__(self, outs: dict, info: torch.Tensor, pairs: int, done=None):
        self.outs, self.info, self.pairs, self.done = outs, info, pairs, done
        self._result = None

    def wait(self) -> Dict[str, torch.Tensor]:
        if self._result is not None:    # idempotent
            return self._result
        outs = self.outs
        if self.done is not None:   # side-stream path: the host waits for gather + unpack + the copy of the host block ONLY; the compute stream is ordered behind them
            self.done.synchronize()
            torch.cuda.current_stream(outs["matches0"].device).wait_event(self.done)
        host = self.info.tolist()       # THE host synchronisation of the step (a no-op on the side-stream path: `info` is pinned host memory there)
        from .lightglue import LightGlue
        LightGlue._raise_on_status(host[2])
        counts = host[1]
        mlist, mscores = outs.pop("_mlist"), outs.pop("_mscores")
        outs["matches"] = [row[:c] for row, c in zip(mlist.unbind(0), counts)]
        outs["scores"] = [row[:c] for row, c in zip(mscores.unbind(0), counts)]
        if self.pairs == 1:
            outs["stop"] = int(host[0][0])
        self._result = outs
        return outs
