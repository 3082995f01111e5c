#!/usr/bin/env python3
"""Adaptive run for profiling the pruning kernels: prints per-layer kept counts and the algorithmic bytes the in-place
compaction moved ((N_cur + N_keep) * (256*4 + 2*32*4) + 8*N_cur per pruned image-layer)."""
import sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import gpu_util
from lightglue_amd import synthetic as synth
B, n = 16, 2048
sd = synth.make_state_dict(0, recipe="B")
model = gpu_util.make_model(sd, "f16x3", depth_confidence=-1, pruning_min_kpts=-1)   # pruning every layer, no early stop
data = gpu_util.to_torch(synth.make_batch(1, B, n, n))
for _ in range(3): out = model(data)
torch.cuda.synchronize()
tot = 0.0
for img in ("prune0", "prune1"):
    p = out[img].cpu().numpy()                         # value v: survived v-1 prunings (pruning happens after layers 0..7)
    for layer in range(8):
        cur = (p >= layer + 1).sum(1)                   # live before the pruning of this layer
        keep = (p >= layer + 2).sum(1)
        tot += float(((cur + keep) * (256 * 4 + 2 * 32 * 4) + 8 * cur).sum())
    print(img, "kept after each layer (pair 0):", [(int((p[0] >= l + 2).sum())) for l in range(8)])
print(f"algorithmic compaction bytes per forward: {tot / 1e6:.1f} MB over 8 layers x 2 images x {B} pairs")
