"""Pair-sharded data parallelism over the GPUs of one node.

Image pairs are fully independent (no cross-pair op anywhere in the reference's ``_forward``), so the
batch shards by contiguous blocks of pairs with NO data-path collective; one process per GPU
(``torchrun``), weights replicated.  The only exchange is the result gather: ONE fixed-shape
``all_gather_into_tensor`` of a packed int32 buffer per batch (match indices; scores are carried as
their fp32 bit patterns in the same buffer) — RCCL over xGMI on the GPU (backend "nccl"), gloo in
the CPU tests.  Payload is a few MB at most, i.e. latency-bound on the xGMI ring; it is issued on the
caller's stream right after the local forward.

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

    def __init__(self, matcher: Callable[[dict], dict], group=None):
        self.matcher = matcher
        self.group = group

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

    def forward_local(self, local: dict, global_batch: int, shards: Optional[List[List[int]]] = None) -> Dict[str, torch.Tensor]:
        m = local["image0"]["keypoints"].shape[1]
        n = local["image1"]["keypoints"].shape[1]
        world, rank = self.world, self.rank
        if shards is None:
            shards = [list(range(*shard_range(global_batch, r, world))) for r in range(world)]
        nloc = len(shards[rank])
        assert local["image0"]["keypoints"].shape[0] == nloc, "local shard size does not match the pair assignment"
        dev = local["image0"]["keypoints"].device
        out = self.matcher(local) if nloc > 0 else None
        # ---- pack [pairs_max][2m + 2n + 1] int32: matches0 | bits(scores0) | matches1 | bits(scores1) | stop
        per_rank = (global_batch + world - 1) // world
        width = 2 * m + 2 * n + 1
        buf = torch.zeros((per_rank, width), dtype=torch.int32, device=dev)
        if nloc > 0:
            stop = out["stop"]
            stop_t = torch.full((nloc,), int(stop), dtype=torch.int32, device=dev) if not torch.is_tensor(stop) else stop.to(dev, torch.int32).reshape(nloc)
            buf[:nloc, 0:m] = out["matches0"].to(torch.int32)
            buf[:nloc, m:2 * m] = out["matching_scores0"].to(torch.float32).contiguous().view(torch.int32)
            buf[:nloc, 2 * m:2 * m + n] = out["matches1"].to(torch.int32)
            buf[:nloc, 2 * m + n:2 * m + 2 * n] = out["matching_scores1"].to(torch.float32).contiguous().view(torch.int32)
            buf[:nloc, -1] = stop_t
        if world > 1:
            # RCCL ("nccl") gathers device buffers directly; gloo (CPU tests, or a debugging run of several ranks
            # on one GPU) goes through host copies
            via_host = dist.get_backend(self.group) == "gloo" and buf.is_cuda
            send = buf.cpu() if via_host else buf
            gathered = torch.empty((world * per_rank, width), dtype=torch.int32, device=send.device)
            dist.all_gather_into_tensor(gathered, send, group=self.group)
            if via_host:
                gathered = gathered.to(dev)
            # rank r's k-th row is pair shards[r][k]: one gather of rows puts the batch back in input order
            src = torch.empty(global_batch, dtype=torch.long)
            for r, sh in enumerate(shards):
                src[torch.tensor(sh, dtype=torch.long)] = r * per_rank + torch.arange(len(sh))
            full = gathered.index_select(0, src.to(gathered.device))
        else:
            full = buf[:nloc]
        m0 = full[:, 0:m].long()
        ms0 = full[:, m:2 * m].contiguous().view(torch.float32)
        m1 = full[:, 2 * m:2 * m + n].long()
        ms1 = full[:, 2 * m + n:2 * m + 2 * n].contiguous().view(torch.float32)
        stop = full[:, -1].long()
        return {"matches0": m0, "matches1": m1, "matching_scores0": ms0, "matching_scores1": ms1, "stop": stop}

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
