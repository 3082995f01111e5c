#!/usr/bin/env python3
"""Shader-clock split of lse_sweep_kernel: streaming loop vs cross-lane row reduction (profiling tap, tail_timing = 4)."""
import sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import gpu_util
from lightglue_amd import synthetic as synth
sd = synth.make_state_dict(0, recipe="A")
model = gpu_util.make_model(sd, "bf16x3", depth_confidence=-1, width_confidence=-1)
data = gpu_util.to_torch(synth.make_batch(1, 32, 1024, 1024))
model(data); model.set_option("tail_timing", 4); model(data); torch.cuda.synchronize()
d = model.debug_read("TAILDBG", np.int64)[: 32 * 32 * 4].reshape(-1, 4)
loop, red = (d[:, 1] - d[:, 0]).astype(float), (d[:, 2] - d[:, 1]).astype(float)
print("lse_sweep per workgroup (shader clocks): loop median %.0f p10 %.0f p90 %.0f | reduction median %.0f" % (np.median(loop), np.percentile(loop, 10), np.percentile(loop, 90), np.median(red)))
print("start spread (clocks): %d" % (d[:, 0].max() - d[:, 0].min()), " end spread:", int(d[:, 2].max() - d[:, 0].min()))
