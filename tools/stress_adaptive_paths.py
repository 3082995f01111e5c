#!/usr/bin/env python3
"""Randomised soak of the round-6 paths on a GPU: for seeded random shapes / thresholds / ragged counts / recipes, (1) the gather projection (adapt_gather = 1, the
product path) against the in-place compaction kernel (adapt_gather = 0) — every output bit for bit —, (2) PairShardedMatcher's world-of-one step (engine-packed wire
row -> lg_unpack_wire -> dict) against LightGlue.forward — same keys, dtypes, values, ragged lists —, (3) the same forward twice (buffer sets flip inside a forward), (4) the similarity matrix on f16 planes (sim_planes = 1) against the generic kernel.
usage: python tools/stress_adaptive_paths.py [cases=60] [seed=0]"""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import gpu_util  # noqa: E402
from lightglue_amd import PairShardedMatcher  # noqa: E402
from lightglue_amd import synthetic as synth  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.Generator(np.random.PCG64(int(sys.argv[2]) if len(sys.argv) > 2 else 0))
KEYS = ("matches0", "matches1", "matching_scores0", "matching_scores1", "prune0", "prune1")
bad = 0
pruned_cases = 0
for c in range(cases):
    recipe = str(rng.choice(["B", "C", "C", "D"]))
    B = int(rng.integers(1, 6)); n = int(rng.integers(40, 2600)); m = int(rng.integers(40, 2600))
    if n * m * B > 3_000_000:
        m = max(40, 3_000_000 // (n * B))
    kw = dict(pruning_min_kpts=int(rng.choice([-1, 64, 300, 1536])))
    if rng.random() < 0.25:
        kw["depth_confidence"] = -1          # pruning without early stop
    if rng.random() < 0.15:
        kw["width_confidence"] = -1          # early stop without pruning
    prec = str(rng.choice(["f16x3", "f16x3", "f16x3/fp16", "fp32"]))
    sd = synth.make_state_dict(int(rng.integers(0, 5)), recipe=recipe)
    data = gpu_util.to_torch(synth.make_batch(int(rng.integers(0, 10_000)), B, n, m, **(synth.RECIPE_D_DATA if recipe == "D" else {})))
    if rng.random() < 0.4:   # ragged, now and then with an empty image
        c0 = rng.integers(0 if rng.random() < 0.2 else 1, n + 1, B); c1 = rng.integers(1, m + 1, B)
        data["image0"]["num_keypoints"] = torch.as_tensor(c0, dtype=torch.int32, device="cuda")
        data["image1"]["num_keypoints"] = torch.as_tensor(c1, dtype=torch.int32, device="cuda")
    model = gpu_util.make_model(sd, prec, **kw)
    on = model(data)
    again = model(data)
    sharded = PairShardedMatcher(model)(data)
    model.set_option("adapt_gather", 0)
    off = model(data)
    model.set_option("adapt_gather", 1)
    model.set_option("sim_planes", 0)        # (4) split-f16: the similarity matrix on f16 planes (lg_sim.hip, the product path) against the generic fp32-row kernel
    generic = model(data)
    msgs = []
    for k in KEYS:
        if not torch.equal(on[k], generic[k]): msgs.append(f"sim on planes != generic sim kernel: {k}")
    for k in KEYS:
        if not torch.equal(on[k], off[k]): msgs.append(f"gather != compaction: {k}")
        if not torch.equal(on[k], again[k]): msgs.append(f"second forward differs: {k}")
        if not (sharded[k].dtype == on[k].dtype and torch.equal(sharded[k], on[k])): msgs.append(f"sharded step differs: {k}")
    if not torch.equal(torch.as_tensor(on["stop"]), torch.as_tensor(off["stop"])): msgs.append("stop differs (compaction)")
    if not torch.equal(torch.as_tensor(on["stop"]).cpu(), torch.as_tensor(sharded["stop"]).cpu()): msgs.append("stop differs (sharded)")
    if set(sharded) != set(on): msgs.append(f"key sets differ: {set(sharded) ^ set(on)}")
    for a, b, x, y in zip(on["matches"], sharded["matches"], on["scores"], sharded["scores"]):
        if not (torch.equal(a, b) and torch.equal(x, y)): msgs.append("ragged lists differ"); break
    did_prune = bool(on["prune0"].dtype == torch.int64 and ((on["prune0"] > 0) & (on["prune0"] < torch.as_tensor(on["stop"]).reshape(-1, 1).to(on["prune0"].device))).any())
    pruned_cases += did_prune
    print(f"case {c:3d} recipe {recipe} {prec:11s} B={B} n={n:4d} m={m:4d} {kw} ragged={'num_keypoints' in data['image0']} stop={on['stop'] if not torch.is_tensor(on['stop']) else on['stop'].tolist()} pruned={did_prune}:",
          "ok" if not msgs else "; ".join(msgs), flush=True)
    bad += bool(msgs)
print(f"{cases} cases, {pruned_cases} with rows actually pruned, {bad} FAILED" if bad else f"{cases} cases, {pruned_cases} with rows actually pruned: all bit-identical")
sys.exit(1 if bad else 0)
