#!/usr/bin/env python3
"""GPU diagnostic: run one golden case N times in one process; report run-to-run differences (race detector)."""
import sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests")); sys.path.insert(0, str(ROOT / "tools"))
import gpu_util, make_golden
from conftest import load_golden
name, prec, reps = sys.argv[1], sys.argv[2], int(sys.argv[3])
meta, gold = load_golden(name); case = meta["case"]
sd, data = make_golden.case_inputs(case)
kw = dict(case["conf"])
if "prune_th" in case: kw["pruning_min_kpts"] = case["prune_th"]
model = gpu_util.make_model(sd, prec, **kw)
for opt in sys.argv[4:]:
    k, v = opt.split("="); model.set_option(k, int(v))
t = gpu_util.to_torch(data)
first = None
for i in range(reps):
    out = model(t); torch.cuda.synchronize()
    s0, s1 = out["matching_scores0"].cpu().numpy(), out["matching_scores1"].cpu().numpy()
    d0, d1 = np.abs(s0 - gold["matching_scores0"]).max(), np.abs(s1 - gold["matching_scores1"]).max()
    if first is None: first = (s0.copy(), s1.copy())
    print(f"run {i}: max|ds0| {d0:.3e} max|ds1| {d1:.3e}  vs run 0: {np.abs(s0 - first[0]).max():.3e} {np.abs(s1 - first[1]).max():.3e}  idx mism {(out['matches0'].cpu().numpy() != gold['matches0']).sum()}")
