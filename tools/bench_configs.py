#!/usr/bin/env python3
"""Time the HIP path on the five BASELINE.json configs (1 GPU).  Writes gpurun_out/configs.md.
    tools/bench_configs.py [tag ...]      only the configs whose name starts with one of the tags, e.g.  "#2 " "#3' " "#5' "
(the library is picked by LIGHTGLUE_AMD_LIB as everywhere: a tile map or kernel variant is A/B'd on the adaptive / ragged configs too)"""
import sys, time
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import gpu_util
from lightglue_amd import synthetic as synth

CFGS = [
    ("#1 SuperPoint N=M=512 B=1 fp32 non-adaptive", dict(B=1, n=512, m=512, dim=256, prec="fp32", recipe="A", kw=dict(depth_confidence=-1, width_confidence=-1))),
    ("#1' same, f16x3", dict(B=1, n=512, m=512, dim=256, prec="f16x3", recipe="A", kw=dict(depth_confidence=-1, width_confidence=-1))),
    ("#2 SuperPoint N=M=1024 B=32 f16x3 non-adaptive", dict(B=32, n=1024, m=1024, dim=256, prec="f16x3", recipe="A", kw=dict(depth_confidence=-1, width_confidence=-1))),
    ("#3 SuperPoint N=M=2048 adaptive (0.95/0.99) B=1 f16x3, recipe C (mixed stop depths)", dict(B=1, n=2048, m=2048, dim=256, prec="f16x3", recipe="C", kw=dict())),
    ("#3' same, B=16", dict(B=16, n=2048, m=2048, dim=256, prec="f16x3", recipe="C", kw=dict())),
    ("#3b recipe B (every pair stops after 3 layers), B=16", dict(B=16, n=2048, m=2048, dim=256, prec="f16x3", recipe="B", kw=dict())),
    ("#3'' N=M=2048 NON-adaptive B=16 f16x3 (for comparison)", dict(B=16, n=2048, m=2048, dim=256, prec="f16x3", recipe="C", kw=dict(depth_confidence=-1, width_confidence=-1))),
    ("#4 DISK 128-d N=M=4096 B=32 (one GPU's shard of 256) f16x3", dict(B=32, n=4096, m=4096, dim=128, prec="f16x3", recipe="A", kw=dict(depth_confidence=-1, width_confidence=-1, input_dim=128))),
    ("#5 ALIKED 128-d N=2048 M=512 B=64 fp16 adaptive, recipe C", dict(B=64, n=2048, m=512, dim=128, prec="fp16", recipe="C", wseed=2, kw=dict(input_dim=128))),
    ("#5' same in f16x3 (the parity-holding mode)", dict(B=64, n=2048, m=512, dim=128, prec="f16x3", recipe="C", wseed=2, kw=dict(input_dim=128))),
]
lines = ["| config | pairs/s | ms/batch | mean stop | stop histogram (layers 1..9) | mean kept pts img0 (last layer) | matches/pair |", "|---|---|---|---|---|---|---|"]
ONLY = sys.argv[1:]
for name, c in CFGS:
    if ONLY and not any(name.startswith(t) for t in ONLY):
        continue
    sd = synth.make_state_dict(c.get("wseed", 0), recipe=c["recipe"], input_dim=c["dim"])
    model = gpu_util.make_model(sd, c["prec"], **c["kw"])
    data = gpu_util.to_torch(synth.make_batch(1, c["B"], c["n"], c["m"], c["dim"]))
    for _ in range(3): out = model(data)
    torch.cuda.synchronize(); t0 = time.perf_counter(); reps = 10
    for _ in range(reps): out = model(data)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    stop = out["stop"]; st = np.atleast_1d(np.asarray(stop if not torch.is_tensor(stop) else stop.cpu().numpy())).astype(int)
    hist = np.bincount(st, minlength=10)[1:].tolist(); stop = float(st.mean())
    p0 = out["prune0"].float(); kept = float((p0 >= p0.max(dim=1, keepdim=True).values).float().sum(1).mean())
    nm = float(np.mean([int(x.shape[0]) for x in out["matches"]]))
    lines.append(f"| {name} | {c['B'] / dt:.1f} | {dt * 1e3:.2f} | {stop:.2f} | {hist} | {kept:.0f} | {nm:.0f} |")
    print(lines[-1], flush=True)
    del model
Path(ROOT / "gpurun_out").mkdir(exist_ok=True)
if not ONLY:
    (ROOT / "gpurun_out" / "configs.md").write_text("\n".join(lines) + "\n")
