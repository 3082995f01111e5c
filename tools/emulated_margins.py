#!/usr/bin/env python3
"""Per-fixture PREDICTION (CPU emulation of the operand rounding, no GPU): max / rms |dscore| against the reference's golden
vectors for the default precision ("f16x3", split attention) and for the fast opt-in ("f16x3/fp16": one f16 plane for q / k / v / P), to be
compared with what tools/gpu_lab.py measures on the GPU.
usage: tools/emulated_margins.py [max keypoints per image, default 2048] > profiles/<round>_emulated_margins.md"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests")); sys.path.insert(0, str(ROOT / "tools"))
from conftest import golden_names, load_golden, oracle_conf_for  # noqa: E402
import make_golden  # noqa: E402
from oracle import lightglue_oracle as O  # noqa: E402

nmax = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
modes = {"f16x3": O.DEFAULT_PRECISION_QUANT, "f16x3/fp16": O.FAST_ATTENTION_QUANT}
print("| fixture | " + " | ".join(f"{m}: idx flips / max / rms" for m in modes) + " |")
print("|---|" + "---|" * len(modes))
tot = {m: [0, 0.0, []] for m in modes}
for name in golden_names():
    meta, gold = load_golden(name)
    case = meta["case"]
    if max(case["n"], case["m"]) > nmax:
        continue
    sd, data = make_golden.case_inputs(case)
    cells = []
    for m, q in modes.items():
        t = time.time()
        out = O.forward(sd, oracle_conf_for(case), data, quant=q)
        d = np.abs(np.asarray(out["matching_scores0"]) - gold["matching_scores0"]).ravel()
        flips = int((np.asarray(out["matches0"]) != gold["matches0"]).sum())
        same = (np.asarray(out["matches0"]) == gold["matches0"]).ravel()
        dmax = float(d[same].max()) if same.any() else 0.0
        rms = float(np.sqrt(np.mean(d[same] ** 2))) if same.any() else 0.0
        cells.append(f"{flips} / {dmax:.2e} / {rms:.2e}")
        tot[m][0] += flips; tot[m][1] = max(tot[m][1], dmax); tot[m][2].append(rms)
    print(f"| {name} | " + " | ".join(cells) + " |", flush=True)
print("| **all** | " + " | ".join(f"{tot[m][0]} / {tot[m][1]:.2e} / {np.mean(tot[m][2]):.2e}" for m in modes) + " |")
