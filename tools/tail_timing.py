#!/usr/bin/env python3
"""Per-phase shader-clock breakdown of the fused tail kernel (profiling tap)."""
import sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import gpu_util
from lightglue_amd import synthetic as synth
prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
which = int(sys.argv[2]) if len(sys.argv) > 2 else 1   # 1 = fused tail (the forward's last launch: no fused projection), 2 = self projection,
# 5 / 6 = layer 0's CrossBlock / SelfBlock tail WITH its fused next projection (768 / 512 columns; the projection's stamps come from TAILDBG2)
sd = synth.make_state_dict(0, recipe="A")
model = gpu_util.make_model(sd, prec, depth_confidence=-1, width_confidence=-1)
data = gpu_util.to_torch(synth.make_batch(1, 32, 1024, 1024))
for kv in sys.argv[3:]:
    k, v = kv.split("="); model.set_option(k, int(v))
model(data); model.set_option("tail_timing", which); model(data); torch.cuda.synchronize()
nst = 6
d = model.debug_read("TAILDBG", np.int64).reshape(-1, 8, 8)[:, :, :nst]
d = d[d[:, 0, 0] != 0]   # 128-row workgroups fill only half of the 64-row slots
dt = np.diff(d, axis=2).astype(np.float64)
names = ["phaseA", "LN", "GELU0", "phaseB+GELU", "epilogue"] if which in (1, 5, 6) else ["A tile -> LDS", "pass0 MFMA", "pass0 epilogue", "pass1 MFMA", "pass1 epilogue"]
print(prec, "which", which, "clock ticks per wave (median / p10 / p90) over", dt.shape[0], "blocks x 8 waves; s_memtime ticks at 100 MHz => x ~21 shader cycles")
for i, n in enumerate(names):
    v = dt[:, :, i].ravel()
    print(f"  {n:10s} {np.median(v):9.0f} {np.percentile(v,10):9.0f} {np.percentile(v,90):9.0f}")
tot = (d[:, :, nst - 1] - d[:, :, 0]).ravel()
print(f"  total      {np.median(tot):9.0f}")
starts = d[:, 0, 0]; print("  block start spread (ticks):", int(starts.max() - starts.min()), " distinct rounds ~", np.unique(np.round((starts - starts.min()) / max(np.median(tot), 1))).size)
if which in (5, 6):   # the fused next projection: stamps 1..5 of TAILDBG2 continue the tail's stamp 5
    p = model.debug_read("TAILDBG2", np.int64).reshape(-1, 8, 8)
    full = model.debug_read("TAILDBG", np.int64).reshape(-1, 8, 8)
    keep = full[:, 0, 0] != 0
    p, full = p[keep], full[keep]
    seq = np.concatenate([full[:, :, 5:6], p[:, :, 1:6]], axis=2)
    dp = np.diff(seq, axis=2).astype(np.float64)
    for i, n in enumerate(["x tile -> LDS + barrier", "pass0 MFMA", "pass0 epilogue", "pass1 MFMA", "pass1 epilogue"]):
        v = dp[:, :, i].ravel()
        print(f"  proj {n:24s} {np.median(v):9.0f} {np.percentile(v,10):9.0f} {np.percentile(v,90):9.0f}")
    print(f"  tail + projection total {np.median((p[:, :, 5] - full[:, :, 0]).ravel()):9.0f}")
