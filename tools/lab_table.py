#!/usr/bin/env python3
"""gpurun_out/.../lab.json (tools/gpu_lab.py) -> the per-fixture markdown table of profiles/rNN_fixture_errors.md.   usage: tools/lab_table.py lab.json [prec ...]"""
import json
import sys

d = json.load(open(sys.argv[1]))["golden"]
precs = sys.argv[2:] or ["f16x3", "f16x3/fp16", "fp32"]
names = sorted({k[len(p) + 1:] for p in precs for k in d if k.startswith(p + "/") and not k[len(p) + 1:].startswith("fp16/")})
print("| fixture | " + " | ".join(f"{p}: idx 0 / 1, max, rms" for p in precs) + " |")
print("|---|" + "---|" * len(precs))
for n in names:
    cells = []
    for p in precs:
        r = d.get(f"{p}/{n}")
        cells.append("—" if r is None else ("EXC" if isinstance(r, str) else f"{r['idx_mismatch0']} / {r['idx_mismatch1']}, {r['max_dscore']:.2e}, {r['rms_dscore']:.2e}"))
    print(f"| {n} | " + " | ".join(cells) + " |")
