#!/usr/bin/env python
"""Bit-identity of the split-attention variants (engine option attn_pp) against the one-barrier kernel: whole-forward outputs on shapes that end
mid-tile, unequal cross sets, one-tile key sets, adaptive runs with compaction, recipe-D statistics, NaN-poisoned workspace."""
import sys
import torch
sys.path.insert(0, ".")
from lightglue_amd import synthetic as synth
from tests import gpu_util

variants = [int(x) for x in sys.argv[1:]] or [1, 2, 3, 4, 5, 6]
cases = ((300, 333, "A", dict(depth_confidence=-1, width_confidence=-1)), (130, 520, "B", dict(pruning_min_kpts=64)),
         (1024, 1024, "A", dict(depth_confidence=-1, width_confidence=-1)), (40, 700, "C", dict()), (64, 65, "A", dict(depth_confidence=-1, width_confidence=-1)),
         (700, 900, "D", dict(depth_confidence=-1, width_confidence=-1)), (2048, 2048, "C", dict()))
bad = 0
for (n0, n1, recipe, kw) in cases:
    sd = synth.make_state_dict(0, recipe=recipe)
    model = gpu_util.make_model(sd, "f16x3", **kw)
    model.set_option("attn_rows", 32)
    poison = gpu_util.to_torch(synth.make_batch(5, 2, max(n0, 384), max(n1, 384)))
    poison["image0"]["descriptors"][:] = float("nan"); poison["image1"]["descriptors"][:] = float("nan")
    data = gpu_util.to_torch(synth.make_batch(23, 2, n0, n1, **(synth.RECIPE_D_DATA if recipe == "D" else {})))
    model(poison)
    base = model(data)
    for v in variants:
        model.set_option("attn_pp", v)
        model(poison)
        out = model(data)
        for key in ("matches0", "matches1", "matching_scores0", "matching_scores1", "prune0", "prune1"):
            if not torch.equal(base[key], out[key]):
                bad += 1
                print("MISMATCH", n0, n1, recipe, "variant", v, key, float((base[key].float() - out[key].float()).abs().max()))
        if not torch.isfinite(out["matching_scores0"]).all():
            bad += 1; print("NONFINITE", n0, n1, recipe, v)
    model.set_option("attn_pp", 0)
    print("case", n0, n1, recipe, "done", flush=True)
print("attn variants", variants, "FAILED" if bad else "bit-identical on all cases")
sys.exit(1 if bad else 0)
