#!/usr/bin/env python3
"""CPU emulation study (round 6): which linear layers tolerate ONE f16 plane for the WEIGHTS (activations still hi + lo: 2 MFMAs per product instead of 3,
and half the L2 -> VGPR weight stream).  Prints max / rms |dscore| and index flips against the reference's golden vectors.
usage: tools/study_weight_planes.py <fixture> [<fixture> ...]"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests")); sys.path.insert(0, str(ROOT / "tools"))
from conftest import load_golden, oracle_conf_for  # noqa: E402
import make_golden  # noqa: E402
from oracle import lightglue_oracle as O  # noqa: E402

W1 = ("fp16x2", "fp16")
base = dict(O.DEFAULT_PRECISION_QUANT)
modes = {
    "default": base,
    "ffn0+out w1": {**base, "lin_ffn0": W1, "lin_out": W1},
    "ffn3 w1": {**base, "lin_ffn3": W1},
    "qkv w1": {**base, "lin_qkv": W1},
    "ffn0+out+ffn3 w1": {**base, "lin_ffn0": W1, "lin_out": W1, "lin_ffn3": W1},
    "all lin w1": {**base, "lin": W1},
    "ffn0+out x1": {**base, "lin_ffn0": ("fp16", "fp16x2"), "lin_out": ("fp16", "fp16x2")},
    "ffn3 x1": {**base, "lin_ffn3": ("fp16", "fp16x2")},
}
sel = [a for a in sys.argv[1:] if not a.startswith("--")]
only = [a[2:] for a in sys.argv[1:] if a.startswith("--")]
if only:
    modes = {k: v for k, v in modes.items() if k in only or k == "default"}
print("| fixture | " + " | ".join(f"{m}: flips / max / rms" for m in modes) + " |")
print("|---|" + "---|" * len(modes))
for name in sel:
    meta, gold = load_golden(name)
    case = meta["case"]
    sd, data = make_golden.case_inputs(case)
    cells = []
    for m, q in modes.items():
        out = O.forward(sd, oracle_conf_for(case), data, quant=q)
        d = np.abs(np.asarray(out["matching_scores0"]) - gold["matching_scores0"]).ravel()
        flips = int((np.asarray(out["matches0"]) != gold["matches0"]).sum())
        same = (np.asarray(out["matches0"]) == gold["matches0"]).ravel()
        cells.append(f"{flips} / {d[same].max():.2e} / {np.sqrt(np.mean(d[same] ** 2)):.2e}")
    print(f"| {name} | " + " | ".join(cells) + " |", flush=True)
