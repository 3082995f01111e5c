#!/usr/bin/env python3
"""GPU diagnostic: one golden case in one precision — where do indices / scores differ from the reference fixture?"""
import sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests")); sys.path.insert(0, str(ROOT / "tools"))
import gpu_util, make_golden
from conftest import load_golden
name, prec = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "bf16x3")
meta, gold = load_golden(name); case = meta["case"]
sd, data = make_golden.case_inputs(case)
kw = dict(case["conf"])
if "prune_th" in case: kw["pruning_min_kpts"] = case["prune_th"]
model = gpu_util.make_model(sd, prec, **kw)
if case.get("static_lengths"): model.static_lengths = list(case["static_lengths"])
out = model(gpu_util.to_torch(data)); torch.cuda.synchronize()
for side in ("0", "1"):
    m, s = out["matches" + side].cpu().numpy(), out["matching_scores" + side].cpu().numpy()
    gm, gs = gold["matches" + side], gold["matching_scores" + side]
    d = np.abs(s - gs)
    print(f"side {side}: index mismatches {int((m != gm).sum())}  max|dscore| {d.max(initial=0):.3e}  #>1e-3: {int((d > 1e-3).sum())}")
    for b, i in zip(*np.where((m != gm) | (d > 1e-3))):
        print(f"   pair {b} kpt {i}: got ({m[b, i]}, {s[b, i]:.6f}) ref ({gm[b, i]}, {gs[b, i]:.6f})")
print("stop", out["stop"], gold["stop"].tolist())
