#!/usr/bin/env python3
"""Throughput of the SuperPoint extractor on the MI355X: conv stack alone and the full extract
(conv stack + NMS / top-k + descriptor head), VGA and 1024x768 images.  Algorithmic FLOPs: 2 * k*k * Cin * Cout per output pixel."""
import sys, time
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tools"))
import make_golden_superpoint as G
from lightglue_amd import SuperPoint


def flops(h, w):
    L = [(1, 64, 9, 1), (64, 64, 9, 1), (64, 64, 9, 4), (64, 64, 9, 4), (64, 128, 9, 16), (128, 128, 9, 16), (128, 128, 9, 64), (128, 128, 9, 64),
         (128, 256, 9, 64), (256, 65, 1, 64), (128, 256, 9, 64), (256, 256, 1, 64)]
    return sum(2.0 * k * ci * co * h * w / d for ci, co, k, d in L)


for prec in ("fp32", "f16x3"):
    model = SuperPoint(weights=G.encoder_state_dict(0), max_num_keypoints=2048, conv_precision=prec).cuda().eval()
    for (b, h, w) in ((8, 480, 640), (8, 768, 1024)):
        img = torch.rand(b, 1, h, w, device="cuda")
        for name, fn in (("conv stack", lambda: model.encode(img)), ("full extract", lambda: model({"image": img}))):
            for _ in range(3): fn()
            torch.cuda.synchronize(); t0 = time.perf_counter(); reps = 10
            for _ in range(reps): fn()
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
            peak = 157.3e12 if prec == "fp32" else 2500e12
            print(f"{prec:5s} {name:13s} B={b} {h}x{w}: {dt * 1e3:7.2f} ms/batch  {b / dt:8.1f} images/s  {flops(h, w) * b / dt / 1e12:6.1f} TFLOP/s algorithmic "
                  f"({flops(h, w) * b / dt / peak * 100:4.1f} % of the {'157 TF f32' if prec == 'fp32' else '2 500 TF f16 (x3 MFMAs per product issued)'} MFMA peak)")
