#!/usr/bin/env python3
"""GPU diagnostic: stop the pipeline after a given step several times and diff the q/k/v^T buffers run to run."""
import sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import gpu_util
from lightglue_amd import synthetic as synth
step = int(sys.argv[1]) if len(sys.argv) > 1 else 1
sd = synth.make_state_dict(0, recipe="A")
model = gpu_util.make_model(sd, "bf16x3", depth_confidence=-1, width_confidence=-1)
for opt in sys.argv[2:]:
    k, v = opt.split("="); model.set_option(k, int(v))
t = gpu_util.to_torch(synth.make_batch(1, 4, 1024, 1024))
ref = None
for i in range(6):
    model.debug_stop_after(step); model(t)
    cur = {n: gpu_util.read_attn_buf(model, n).copy() for n in ("Q", "K", "VT")}
    cur["X"] = model.debug_read("X").copy()
    if ref is None: ref = cur; continue
    msg = []
    for n in cur:
        d = cur[n] != ref[n]
        nanboth = np.isnan(cur[n]) & np.isnan(ref[n])
        d &= ~nanboth
        if d.any():
            idx = np.flatnonzero(d.ravel())
            msg.append(f"{n}: {idx.size} differing elements, first flat idx {idx[:6].tolist()} shape {cur[n].shape} vals {cur[n].ravel()[idx[:3]]} vs {ref[n].ravel()[idx[:3]]}")
    print(f"run {i}:", "; ".join(msg) if msg else "identical")
