#!/usr/bin/env python3
"""Is a captured hipGraph of the forward worth anything?  (SURVEY §2.1 names hipGraph replay as the mechanism behind `compile()`, ref :439-454.)
Per shape: wall time per `forward_raw` issued launch by launch vs the same forward captured once and replayed (torch.cuda.CUDAGraph = hipGraph
on ROCm; the engine launches on the capturing stream, its workspace is reserved beforehand, no host synchronisation inside), and that the
replay's outputs are bit-identical."""
import sys, time
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import gpu_util
from lightglue_amd import synthetic as synth

def wall(fn, reps):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3

cases = [(1, 512, "A", False), (1, 1024, "A", False), (1, 2048, "A", False), (1, 2048, "C", True), (4, 1024, "A", False), (16, 2048, "C", True), (32, 1024, "A", False)]
for b, n, recipe, adaptive in cases:
    sd = synth.make_state_dict(0, recipe=recipe)
    kw = {} if adaptive else dict(depth_confidence=-1, width_confidence=-1)
    model = gpu_util.make_model(sd, "f16x3", **kw)
    model.track_inplace_weight_edits = False
    data = gpu_util.to_torch(synth.make_batch(1, b, n, n))
    model.reserve(b, n, n)
    ref = model.forward_raw(data); torch.cuda.synchronize()
    ref = {k: v.clone() for k, v in ref.items() if torch.is_tensor(v)}
    reps = 200 if b == 1 else 30
    plain = wall(lambda: model.forward_raw(data), reps)
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        model.forward_raw(data)
    torch.cuda.current_stream().wait_stream(side)
    with torch.cuda.graph(g):
        out = model.forward_raw(data)
    g.replay(); torch.cuda.synchronize()
    same = all(torch.equal(out[k], ref[k]) for k in ref)
    replay = wall(g.replay, reps)
    print(f"B={b} N={n} recipe {recipe}{' adaptive' if adaptive else ''}: launches {plain:.3f} ms, graph replay {replay:.3f} ms ({(replay / plain - 1) * 100:+.1f} %), outputs identical: {same}", flush=True)
