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


def _worker(rank, world, port, q, ragged=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sharded = PairShardedMatcher(_OracleMatcher())
        batch = _batch(ragged=ragged)
        res = sharded(batch)
        if ragged:   # the work-balanced assignment must actually be non-contiguous here, or the test shows nothing
            assert sharded.assignment(batch) != [[0, 1, 2], [3, 4]]
        q.put((rank, {k: v.numpy() for k, v in res.items()}))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("ragged", [False, True])
def test_two_ranks_equal_single_rank(ragged):
    single = PairShardedMatcher(_OracleMatcher())(_batch(ragged=ragged))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, ragged)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank in (0, 1):
        for k, v in single.items():
            np.testing.assert_array_equal(got[rank][k], v.numpy(), err_msg=f"rank {rank} {k}")
    matches, scores = PairShardedMatcher.ragged(single)
    assert len(matches) == 5 and all(mm.shape[1] == 2 for mm in matches)
