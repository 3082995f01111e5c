#!/usr/bin/env python3
"""Phase-A sub-stamps of the fused tail (diagnostic builds only: tools/gpu_call_r05c.sh builds them from patched copies of csrc/).
TAILDBG slot 0 = kernel start, slot 1 = end of phase A; TAILDBG2 slots = the diagnostic stamps named on the command line."""
import sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import gpu_util
from lightglue_amd import synthetic as synth
names = sys.argv[1].split(",")
sd = synth.make_state_dict(0, recipe="A")
model = gpu_util.make_model(sd, "f16x3", depth_confidence=-1, width_confidence=-1)
data = gpu_util.to_torch(synth.make_batch(1, 32, 1024, 1024))
model(data); model.set_option("tail_timing", 5); model(data); torch.cuda.synchronize()
d = model.debug_read("TAILDBG", np.int64).reshape(-1, 8, 8)
p = model.debug_read("TAILDBG2", np.int64).reshape(-1, 8, 8)
keep = d[:, 0, 0] != 0
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0      # only workgroups with id >= first (e.g. 256: the later rounds of the grid)
keep[:first] = False
d, p = d[keep], p[keep]
order = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else list(range(len(names)))   # time order of the TAILDBG2 slots
seq = np.concatenate([d[:, :, 0:1], p[:, :, order], d[:, :, 1:2]], axis=2)
dt = np.diff(seq, axis=2).astype(np.float64)
for i, n in enumerate(names + ["-> end of phase A"]):
    v = dt[:, :, i].ravel()
    print(f"  {n:28s} {np.median(v):9.0f} {np.percentile(v,10):9.0f} {np.percentile(v,90):9.0f}")
print(f"  phase A total {np.median((d[:, :, 1] - d[:, :, 0]).ravel()):9.0f}")
