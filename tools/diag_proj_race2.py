#!/usr/bin/env python3
"""GPU diagnostic: q of the first SelfBlock projection vs an fp64 host reference, run by run."""
import sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import gpu_util
from lightglue_amd import synthetic as synth
sd = synth.make_state_dict(0, recipe="A")
model = gpu_util.make_model(sd, "bf16x3", depth_confidence=-1, width_confidence=-1)
t = gpu_util.to_torch(synth.make_batch(1, 4, 1024, 1024))
model.debug_stop_after(0); model(t)
X = model.debug_read("X").reshape(-1, 256).astype(np.float64); COS = model.debug_read("COS").reshape(-1, 32).astype(np.float64); SIN = model.debug_read("SIN").reshape(-1, 32).astype(np.float64)
W = sd["transformers.0.self_attn.Wqkv.weight"].astype(np.float64); b = sd["transformers.0.self_attn.Wqkv.bias"].astype(np.float64)
y = X @ W.T + b                       # [R, 768], channel = h*192 + d*3 + which
R = X.shape[0]
q = y.reshape(R, 4, 64, 3)[..., 0].transpose(1, 0, 2)   # [H, R, 64]
k = y.reshape(R, 4, 64, 3)[..., 1].transpose(1, 0, 2)
def rope(z):
    z2 = z.reshape(4, R, 32, 2); c = COS[None, :, :]; s = SIN[None, :, :]
    out = np.empty_like(z2); out[..., 0] = z2[..., 0] * c - z2[..., 1] * s; out[..., 1] = z2[..., 1] * c + z2[..., 0] * s
    return out.reshape(4, R, 64)
qr, kr = rope(q), rope(k)
for i in range(5):
    model.debug_stop_after(1); model(t)
    Q = gpu_util.read_attn_buf(model, "Q").reshape(4, R, 64); K = gpu_util.read_attn_buf(model, "K").reshape(4, R, 64)
    dq, dk = np.abs(Q - qr), np.abs(K - kr)
    bad = np.argwhere(dq > 2e-3)
    print(f"run {i}: max|dq| {dq.max():.3e} max|dk| {dk.max():.3e}  #q>2e-3: {len(bad)}  first {bad[:4].tolist()}")
    if len(bad):
        h, r, d = bad[0]; print("   got", Q[h, r, d-2:d+2], "ref", qr[h, r, d-2:d+2], "unrotated", q[h, r, d-2:d+2], "cos/sin", COS[r, d//2], SIN[r, d//2])
