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

from lightglue_amd.parallel import PairShardedMatcher, shard_range
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
        out = O.forward(self.sd, self.conf, npd)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
        return {"matches0": t(out["matches0"]), "matches1": t(out["matches1"]), "matching_scores0": t(out["matching_scores0"]),
                "matching_scores1": t(out["matching_scores1"]), "stop": torch.tensor(out["stop"])}


def _batch(B=5, n=48, m=40):
    data = synth.make_batch(3, B, n, m)
    return {k: {kk: torch.from_numpy(vv) for kk, vv in v.items()} for k, v in data.items()}


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        res = PairShardedMatcher(_OracleMatcher())(_batch())
        q.put((rank, {k: v.numpy() for k, v in res.items()}))
    finally:
        dist.destroy_process_group()


def test_two_ranks_equal_single_rank():
    single = PairShardedMatcher(_OracleMatcher())(_batch())
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
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
