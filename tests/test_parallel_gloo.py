"""Pair sharding + the result all-gather on 2 CPU processes (gloo).  The per-rank matcher here is the
oracle (no GPU in this container); what is under test is shard_range / PairShardedMatcher: same batch
on 1 vs 2 ranks must be bit-equal after the gather."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from lightglue_amd.parallel import PairShardedMatcher, balanced_shards, pair_cost, shard_range
from oracle import lightglue_oracle as O
from lightglue_amd import synthetic as synth


def test_shard_range_partitions():
    for batch in (1, 2, 7, 32, 256):
        for world in (1, 2, 3, 8):
            spans = [shard_range(batch, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == batch
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


class _OracleMatcher:
    def __init__(self):
        self.sd = synth.make_state_dict(0, recipe="A", n_layers=2)
        self.conf = O.make_conf(depth_confidence=-1, width_confidence=-1, n_layers=2)

    def __call__(self, data):
        npd = {k: {kk: vv.numpy() for kk, vv in v.items()} for k, v in data.items()}
        if "num_keypoints" in npd["image0"]:       # ragged batch = loop of B = 1 calls on each pair's own rows
            return self.ragged(npd)
        out = O.forward(self.sd, self.conf, npd)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
        return {"matches0": t(out["matches0"]), "matches1": t(out["matches1"]), "matching_scores0": t(out["matching_scores0"]),
                "matching_scores1": t(out["matching_scores1"]), "stop": torch.tensor(out["stop"])}


def _ragged_call(self, npd):
    d0, d1 = npd["image0"], npd["image1"]
    B, m, n = d0["keypoints"].shape[0], d0["keypoints"].shape[1], d1["keypoints"].shape[1]
    m0, m1 = -np.ones((B, m), np.int64), -np.ones((B, n), np.int64)
    s0, s1 = np.zeros((B, m), np.float32), np.zeros((B, n), np.float32)
    stop = []
    for b in range(B):
        c0, c1 = int(d0["num_keypoints"][b]), int(d1["num_keypoints"][b])
        r = O.forward_pair(self.sd, self.conf, d0["keypoints"][b][:c0], d1["keypoints"][b][:c1], d0["descriptors"][b][:c0],
                           d1["descriptors"][b][:c1], d0["image_size"][b], d1["image_size"][b])
        m0[b, :c0], m1[b, :c1], s0[b, :c0], s1[b, :c1] = r["matches0"], r["matches1"], r["matching_scores0"], r["matching_scores1"]
        stop.append(r["stop"])
    t = torch.from_numpy
    return {"matches0": t(m0), "matches1": t(m1), "matching_scores0": t(s0), "matching_scores1": t(s1), "stop": torch.tensor(stop)}


_OracleMatcher.ragged = _ragged_call


def test_balanced_shards_properties():
    costs = [pair_cost(a, b) for a, b in [(2048, 2048), (100, 90), (1500, 1024), (64, 64), (1024, 1024), (700, 2000), (10, 10)]]
    for world in (1, 2, 3, 4):
        shards = balanced_shards(costs, world)
        assert sorted(i for sh in shards for i in sh) == list(range(len(costs)))            # a partition
        assert max(len(sh) for sh in shards) <= -(-len(costs) // world)                      # fixed gather shape
        assert shards == balanced_shards(costs, world)                                       # deterministic
    two = balanced_shards(costs, 2)
    load = [sum(costs[i] for i in sh) for sh in two]
    block = [sum(costs[:4]), sum(costs[4:])]
    assert max(load) / min(load) < max(block) / min(block)                                   # better than contiguous blocks


def _batch(B=5, n=48, m=40, ragged=False):
    if ragged:
        data = synth.make_batch(3, B, n, m)
        out = {k: {kk: torch.from_numpy(vv) for kk, vv in v.items()} for k, v in data.items()}
        out["image0"]["num_keypoints"] = torch.tensor([48, 44, 7, 30, 12], dtype=torch.int32)[:B]
        out["image1"]["num_keypoints"] = torch.tensor([40, 40, 9, 5, 33], dtype=torch.int32)[:B]
        return out
    data = synth.make_batch(3, B, n, m)
    return {k: {kk: torch.from_numpy(vv) for kk, vv in v.items()} for k, v in data.items()}


FORWARD_KEYS = {"matches0", "matches1", "matching_scores0", "matching_scores1", "stop", "matches", "scores", "prune0", "prune1"}   # ref lightglue.py:619-629


def _np(res):
    """output dict -> picklable numpy form (the ragged lists as lists of arrays)"""
    conv = lambda v: [x.numpy() for x in v] if isinstance(v, list) else (v.numpy() if torch.is_tensor(v) else v)
    return {k: conv(v) for k, v in res.items()}


def _assert_same(a, b, msg=""):
    assert set(a) == set(b), (msg, set(a) ^ set(b))
    for k in a:
        if isinstance(a[k], list):
            assert len(a[k]) == len(b[k]), (msg, k)
            for x, y in zip(a[k], b[k]):
                np.testing.assert_array_equal(x, y, err_msg=f"{msg} {k}")
        else:
            np.testing.assert_array_equal(a[k], b[k], err_msg=f"{msg} {k}")


class _PruningMatcher(_OracleMatcher):
    """the oracle with int64 prune counters in its output (what a matcher with width_confidence > 0 returns): the wire row's prune block carries counters"""
    wire_prunes = True

    def __call__(self, data):
        out = super().__call__(data)
        out["prune0"] = (out["matches0"] % 7 + 1).long()          # any per-pair deterministic counters >= 1
        out["prune1"] = (out["matches1"] % 5 + 1).long()
        return out


class _PoisonedMatcher(_OracleMatcher):
    """reports LG_ERR_RANGE for global pair 3 (rank 1's shard in a world of two)"""

    def __call__(self, data):
        out = super().__call__(data)
        first = float(data["image0"]["keypoints"][0, 0, 0])
        ref = _batch()["image0"]["keypoints"][:, 0, 0].tolist()
        lo = ref.index(first)                                     # which global pair this shard starts at
        B = out["matches0"].shape[0]
        out["status"] = torch.tensor([4 if lo + k == 3 else 0 for k in range(B)], dtype=torch.int32)
        return out


def _worker(rank, world, port, q, ragged=False, kind="oracle"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sharded = PairShardedMatcher({"oracle": _OracleMatcher, "prune": _PruningMatcher, "poison": _PoisonedMatcher}[kind]())
        batch = _batch(ragged=ragged)
        try:
            res = sharded(batch)
        except Exception as ex:   # the poisoned case: every rank must raise
            q.put((rank, {"error": f"{type(ex).__name__}: {ex}"}))
            return
        if ragged:   # the work-balanced assignment must actually be non-contiguous here, or the test shows nothing
            assert sharded.assignment(batch) != [[0, 1, 2], [3, 4]]
        q.put((rank, _np(res)))
    finally:
        dist.destroy_process_group()


def _run_two_ranks(ragged=False, kind="oracle"):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, ragged, kind)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return got


@pytest.mark.parametrize("ragged", [False, True])
def test_two_ranks_equal_single_rank(ragged):
    """Same batch on 1 vs 2 ranks: the full-batch dict is identical on every rank — and it is the dict `LightGlue.forward` returns:
    same key set, int64 indices / stop, the ragged `matches` / `scores` lists equal to the reference's construction from matches0 (ref :593-602)."""
    single = _np(PairShardedMatcher(_OracleMatcher())(_batch(ragged=ragged)))
    assert set(single) == FORWARD_KEYS
    assert single["matches0"].dtype == np.int64 and single["stop"].dtype == np.int64 and single["prune0"].dtype == np.float32
    got = _run_two_ranks(ragged)
    for rank in (0, 1):
        _assert_same(got[rank], single, f"rank {rank}")
    t = lambda a: torch.from_numpy(a)
    matches, scores = PairShardedMatcher.ragged({"matches0": t(single["matches0"]), "matching_scores0": t(single["matching_scores0"])})
    assert len(matches) == 5 and all(mm.shape[1] == 2 for mm in matches)
    for k in range(5):
        np.testing.assert_array_equal(single["matches"][k], matches[k].numpy())
        np.testing.assert_array_equal(single["scores"][k], scores[k].numpy())


def test_prune_counters_travel_on_the_wire():
    single = _np(PairShardedMatcher(_PruningMatcher())(_batch()))
    assert single["prune0"].dtype == np.int64 and (single["prune0"] >= 1).all()
    got = _run_two_ranks(kind="prune")
    for rank in (0, 1):
        _assert_same(got[rank], single, f"rank {rank}")


def test_a_poisoned_pair_on_rank_1_raises_on_every_rank():
    """the per-pair status travels on the wire row; a pair flagged on rank 1 must raise on rank 0 as well."""
    got = _run_two_ranks(kind="poison")
    for rank in (0, 1):
        assert "error" in got[rank] and "pair 3" in got[rank]["error"] and "LG_ERR_RANGE" in got[rank]["error"], got[rank]


def test_empty_world_of_one_batch():
    """an empty batch in a world of one returns empty outputs instead of failing."""
    b = _batch()
    empty = {k: {kk: vv[:0] for kk, vv in v.items()} for k, v in b.items()}
    res = PairShardedMatcher(_OracleMatcher()).forward_local(empty, 0)
    assert res["matches0"].shape == (0, 48) and res["matches"] == [] and set(res) == FORWARD_KEYS
