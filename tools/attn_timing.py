#!/usr/bin/env python3
"""Per-phase shader-clock breakdown of the attention tile loop.  Needs a library built with -DLG_ATTN_TIMING
(tools/build_variant.sh attn_timing -DLG_ATTN_TIMING) selected through LIGHTGLUE_AMD_LIB."""
import sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import gpu_util
from lightglue_amd import synthetic as synth
prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
sd = synth.make_state_dict(0, recipe="A")
model = gpu_util.make_model(sd, prec, depth_confidence=-1, width_confidence=-1)
data = gpu_util.to_torch(synth.make_batch(1, 32, 1024, 1024))
model(data); model.set_option("tail_timing", 3); model(data); torch.cuda.synchronize()
NW = 8 if prec == "f16x3" else 4   # waves per workgroup: the split attention runs 8 x 16 rows, the single-plane kernels 4 x 32
d = model.debug_read("TAILDBG", np.int64).reshape(-1, NW, 8)
d = d[d[:, 0, 7] == 1]
assert d.shape[0] == 32 * 2048 // (128 if NW == 8 else 128) * 4, f"expected every (row tile, head) workgroup to report, got {d.shape[0]}"   # the tap buffer is sized for the split kernel now
tiles = d[:, :, 6].astype(np.float64)
dma = (len(sys.argv) <= 2 or sys.argv[2] != "staged")
if not dma: model.set_option("attn_dma", 0); model(data); torch.cuda.synchronize(); d = model.debug_read("TAILDBG", np.int64).reshape(-1, NW, 8); d = d[d[:, 0, 7] == 1]; tiles = d[:, :, 6].astype(np.float64)
names = ["barrier 1 (wait for all waves' PV)", "store tile + barrier 2", "next-tile loads + K frags + QK MFMA issue",
         "QK drain + max + shfl + any", "rescale + exp", "pack + V frags + PV MFMA issue"] if not dma else \
        ["vmcnt(0) + barrier", "issue next tile's DMA", "K frags + QK MFMA issue", "QK drain + max + swaps + rescale", "exp", "pack + V frags + PV MFMA issue"]
print(prec, "dma" if dma else "staged", "s_memtime ticks PER TILE per wave; median / p10 / p90 over", d.shape[0], "blocks x", NW, "waves")
tot = 0
for i, n in enumerate(names):
    v = (d[:, :, i] / tiles).ravel()
    tot += np.median(v)
    print(f"  {n:45s} {np.median(v):7.2f} {np.percentile(v,10):7.2f} {np.percentile(v,90):7.2f}")
print(f"  sum of medians {tot:.2f} ticks per tile")
