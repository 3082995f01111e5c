#!/usr/bin/env python3
"""B = 1 latency: wall time per forward vs the sum of kernel times (HIP events) — how launch-bound is it?"""
import sys, time
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import gpu_util
from lightglue_amd import synthetic as synth
import itertools
for n, fused_next, variant, shape in [(512, 1, 0, 0), (512, 1, 0, 4), (1024, 1, 0, 0), (1024, 1, 0, 4), (1024, 1, 0, 2), (1024, 1, 0, 1), (2048, 1, 0, 0), (2048, 1, 0, 4), (2048, 1, 0, 1), (4096, 1, 0, 0), (4096, 1, 0, 4)]:
    sd = synth.make_state_dict(0, recipe="A")
    model = gpu_util.make_model(sd, "f16x3", depth_confidence=-1, width_confidence=-1)
    model.set_option("fused_next", fused_next)
    model.set_option("tail_row_tiles", shape)
    if shape == 4: model.set_option("attn_rows", 32)   # the round-2 operating point: 64-row tail workgroups, 128-row attention workgroups
    data = gpu_util.to_torch(synth.make_batch(1, 1, n, n))
    for _ in range(5): model(data)
    torch.cuda.synchronize(); t0 = time.perf_counter(); reps = 50
    for _ in range(reps): model(data)
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / reps * 1e3
    model.profile(True); 
    for _ in range(reps): model(data)
    prof = model.profile_read(); model.profile(False)
    ksum = sum(v[0] for v in prof.values()) / reps
    top = sorted(((v[0] / reps, k) for k, v in prof.items() if v[1]), reverse=True)[:4]
    print(f"N={n} tail_row_tiles={shape}: wall {wall:.3f} ms/forward, kernel sum {ksum:.3f} ms, top: " + ", ".join(f"{k} {t:.3f}" for t, k in top))
