#!/usr/bin/env python
"""The ring attention (engine option attn_ring) against the one-barrier kernel: whole-forward outputs within fp32 summation-order noise (NOT bit-identical: the
online softmax steps in 32-key tiles), on shapes that end mid-tile, unequal cross sets, one-tile key sets, adaptive runs, recipe D, NaN-poisoned workspace; many
repetitions of one case to catch a rare race."""
import sys
import torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import gpu_util
from lightglue_amd import synthetic as synth

cases = ((300, 333, "A", dict(depth_confidence=-1, width_confidence=-1)), (130, 520, "B", dict(pruning_min_kpts=64)),
         (1024, 1024, "A", dict(depth_confidence=-1, width_confidence=-1)), (40, 700, "C", dict()), (64, 65, "A", dict(depth_confidence=-1, width_confidence=-1)),
         (31, 33, "A", dict(depth_confidence=-1, width_confidence=-1)), (700, 900, "D", dict(depth_confidence=-1, width_confidence=-1)), (2048, 2048, "C", dict()),
         (4096, 4096, "A", dict(depth_confidence=-1, width_confidence=-1)))
bad = 0
for (n0, n1, recipe, kw) in cases:
    sd = synth.make_state_dict(0, recipe=recipe)
    model = gpu_util.make_model(sd, "f16x3", **kw)
    model.check_finite = False
    B = 1 if n0 >= 4096 else 2
    poison = gpu_util.to_torch(synth.make_batch(5, B, max(n0, 384), max(n1, 384)))
    poison["image0"]["descriptors"][:] = float("nan"); poison["image1"]["descriptors"][:] = float("nan")
    data = gpu_util.to_torch(synth.make_batch(23, B, n0, n1, **(synth.RECIPE_D_DATA if recipe == "D" else {})))
    model(poison)
    base = model(data)
    model.set_option("attn_ring", 1)
    model(poison)
    outs = [model(data) for _ in range(20 if n0 <= 1024 else 4)]
    for i, out in enumerate(outs):
        d = float((base["matching_scores0"] - out["matching_scores0"]).abs().max())
        flips = int((base["matches0"] != out["matches0"]).sum())
        same_as_first = all(torch.equal(out[k], outs[0][k]) for k in ("matches0", "matching_scores0", "matching_scores1"))
        stop_eq = torch.equal(torch.as_tensor(base["stop"]), torch.as_tensor(out["stop"]))
        if d > 2e-5 or flips or not same_as_first or not stop_eq or not torch.isfinite(out["matching_scores0"]).all():
            bad += 1
            print("PROBLEM", n0, n1, recipe, "rep", i, "max |dscore|", d, "flips", flips, "deterministic", same_as_first, "stop equal", stop_eq)
    print("case", n0, n1, recipe, "max |dscore| vs one-barrier kernel", float((base["matching_scores0"] - outs[0]["matching_scores0"]).abs().max()), flush=True)
print("ring attention:", "FAILED" if bad else "within summation-order noise and deterministic on all cases")
sys.exit(1 if bad else 0)
