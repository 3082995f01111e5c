#!/usr/bin/env python3
"""CPU emulation on a checkpoint (tools/train_synthetic_checkpoint.py's, or any): which operand of which contraction needs its second f16 plane THERE?  The same rounding
modes as tools/study_weight_planes.py / DESIGN.md section 1, applied through the oracle's `quant` switch, against the oracle's own fp32 result on the same inputs.
usage: study_trained_checkpoint.py <checkpoint.pth> [features.npz] [--n 512]"""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tools"))
from lightglue_amd import synthetic as synth  # noqa: E402
from oracle import lightglue_oracle as O  # noqa: E402
from verify_pretrained import load_checkpoint  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
n = int(sys.argv[sys.argv.index("--n") + 1]) if "--n" in sys.argv else 512
sd, arch, _ = load_checkpoint(args[0])
nsd = {k: v.numpy() for k, v in sd.items()}
cases = [(f"synthetic N=M={n}", synth.make_batch(4000 + n, 1, n, n, arch["input_dim"]))]
if len(args) > 1:
    z = np.load(args[1])
    img = lambda i: {"keypoints": z[f"keypoints{i}"][None].astype(np.float32), "descriptors": z[f"descriptors{i}"][None].astype(np.float32), "image_size": z[f"image_size{i}"][None].astype(np.float32)}
    cases.append((Path(args[1]).name, {"image0": img(0), "image1": img(1)}))
W1, X1 = ("fp16x2", "fp16"), ("fp16", "fp16x2")
base = dict(O.DEFAULT_PRECISION_QUANT)
modes = {"default f16x3": base, "fast opt-in (attention on one plane)": dict(O.FAST_ATTENTION_QUANT), "q k^T on one plane": {**base, "attn_qk": "fp16"}, "P V on one plane": {**base, "attn_pv": "fp16"},
         "all linear WEIGHTS on one plane": {**base, "lin": W1}, "all linear ACTIVATIONS on one plane": {**base, "lin": X1}, "q/k/v projection weights on one plane": {**base, "lin_qkv": W1},
         "ffn.0 + out_proj weights on one plane": {**base, "lin_ffn0": W1, "lin_out": W1}, "ffn.3 weights on one plane": {**base, "lin_ffn3": W1},
         "everything on one f16 plane": {"lin": "fp16", "attn": "fp16", "final": "fp16"}, "everything on one bf16 plane": {"lin": "bf16", "attn": "bf16", "final": "bf16"}}
torch.set_num_threads(8)
conf = O.make_conf(depth_confidence=-1, width_confidence=-1, **arch)
print("| operand rounding | " + " | ".join(f"{c[0]}: flips / max \\|d score\\|" for c in cases) + " |")
print("|---|" + "---|" * len(cases))
refs = [O.forward(nsd, conf, d, backend="torch") for _, d in cases]
for name, q in modes.items():
    cells = []
    for (label, data), ref in zip(cases, refs):
        out = O.forward(nsd, conf, data, quant=q)
        same = np.asarray(out["matches0"]) == np.asarray(ref["matches0"])
        d = np.abs(np.asarray(out["matching_scores0"]) - np.asarray(ref["matching_scores0"]))[same]
        cells.append(f"{int((~same).sum())} / {d.max():.1e}")
    print(f"| {name} | " + " | ".join(cells) + " |", flush=True)
