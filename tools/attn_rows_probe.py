#!/usr/bin/env python3
"""Which attention workgroup shape for which grid?  attn_rows = 16 (64-row workgroups) vs 32 (128-row, LDS-DMA kernel)."""
import sys, time
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import gpu_util
from lightglue_amd import synthetic as synth
sd = synth.make_state_dict(0, recipe="A")
for b, n in ((1, 2048), (1, 4096), (2, 4096), (4, 1024), (8, 1024), (16, 1024)):
    data = gpu_util.to_torch(synth.make_batch(1, b, n, n))
    res = []
    for rows in (16, 32):
        model = gpu_util.make_model(sd, "bf16x3", depth_confidence=-1, width_confidence=-1)
        model.set_option("attn_rows", rows)
        for _ in range(5): model(data)
        model.profile(True); reps = 30
        for _ in range(reps): model(data)
        prof = model.profile_read(); model.profile(False)
        res.append((rows, sum(v[0] for k, v in prof.items() if k.startswith("attn")) / reps, sum(v[0] for v in prof.values()) / reps))
    print(f"B={b} N={n} (128-row workgroups: {b * 2 * n // 128 * 4}):", ", ".join(f"rows {r}: attn {a:.3f} ms, all kernels {t:.3f} ms" for r, a, t in res))
