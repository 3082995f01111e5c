#!/usr/bin/env python3
"""Occupancy timeline of the attention kernel: every workgroup's life span on the 100 MHz wall clock and where it ran.
Needs a library built with -DLG_ATTN_WALL (tools/build_variant.sh attn_wall -DLG_ATTN_WALL) selected through LIGHTGLUE_AMD_LIB."""
import sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import gpu_util
from lightglue_amd import synthetic as synth
sd = synth.make_state_dict(0, recipe="A")
model = gpu_util.make_model(sd, "f16x3", depth_confidence=-1, width_confidence=-1)
data = gpu_util.to_torch(synth.make_batch(1, 32, 1024, 1024))
import os
NW = int(os.environ.get("LG_ATTN_WALL_NW", "8"))      # waves per workgroup of the kernel under test (split attention: 8; attn_dma_kernel: 4)
for kv in os.environ.get("LG_BENCH_OPTS", "").split():
    k, v = kv.split("="); model.set_option(k, int(v))
for _ in range(3): model(data)
model.set_option("tail_timing", 3); model(data); torch.cuda.synchronize()
d = model.debug_read("TAILDBG", np.int64).reshape(-1, NW, 8)
d = d[d[:, 0, 7] == 1][:, 0, :]          # wave 0 of each workgroup
t0, t1, cyc, hw, xcc = d[:, 0], d[:, 1], d[:, 2], d[:, 3], d[:, 4]
base = t0.min()
print("workgroups", len(d), " kernel span (first start -> last end) %.1f us" % ((t1.max() - base) / 100.0))
life = (t1 - t0) / 100.0
print("workgroup life us: median %.1f  p10 %.1f  p90 %.1f  max %.1f;  shader cycles per life: median %.0f -> clock %.0f MHz" %
      (np.median(life), np.percentile(life, 10), np.percentile(life, 90), life.max(), np.median(cyc), np.median(cyc / life)))
assert 0 < t1.max() - base < 10_000_000
ts = np.arange(0, (t1.max() - base), 200)   # every 2 us
live = [(int(((t0 - base) <= t) & ((t1 - base) > t)).sum()) if False else int((((t0 - base) <= t) & ((t1 - base) > t)).sum()) for t in ts]
print("live workgroups every 2 us:", live)
cu = (hw >> 8) & 0xF; se = (hw >> 13) & 0x7; sh = (hw >> 12) & 1
key = (xcc & 0xF) * 10000 + se * 1000 + sh * 100 + cu
u, c = np.unique(key, return_counts=True)
print("distinct (xcc, se, sh, cu):", len(u), " workgroups per CU: min %d max %d" % (c.min(), c.max()))
starts = np.sort((t0 - base) / 100.0)
print("start times us: 25%% %.1f  50%% %.1f  75%% %.1f  100%% %.1f" % tuple(np.percentile(starts, [25, 50, 75, 100])))
first = (t0 - base) < 300          # started in the first 3 us
for name, sel in (("first round", first), ("later", ~first)):
    print(f"{name}: n={sel.sum()}  life us median {np.median(life[sel]):.1f}  cycles median {np.median(cyc[sel]):.0f}  clock MHz median {np.median(cyc[sel] / life[sel]):.0f}")
order = np.argsort(t0)
print("per start-time decile: start us / life us / cycles / MHz")
for q in range(10):
    sel = order[q * len(order) // 10:(q + 1) * len(order) // 10]
    print(f"  {np.median((t0[sel] - base) / 100.0):6.1f} {np.median(life[sel]):6.1f} {np.median(cyc[sel]):8.0f} {np.median(cyc[sel] / life[sel]):6.0f}")
