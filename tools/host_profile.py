#!/usr/bin/env python3
"""Where does the HOST time of a B = 1 forward go?  cProfile of `InflightMatcher.submit(...)` / `.result()` with 8 lanes at N = 512 (the GPU is then never
the bottleneck: the step time IS the host time per forward)."""
import cProfile, pstats, sys, time
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import gpu_util
from lightglue_amd import InflightMatcher, synthetic as synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 8
model = gpu_util.make_model(synth.make_state_dict(0, recipe="A"), "f16x3", depth_confidence=-1, width_confidence=-1)
model.track_inplace_weight_edits = False
data = gpu_util.to_torch(synth.make_batch(1, 1, n, n))
lanes = InflightMatcher(model, depth); lanes.reserve(1, n, n)
def loop(k):
    pend = []
    for _ in range(k):
        pend.append(lanes.submit(data))
        if len(pend) == depth: pend.pop(0).result()
    for p in pend: p.result()
loop(100); torch.cuda.synchronize()
t0 = time.perf_counter(); loop(1000); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"N={n} depth={depth}: {1000 / dt:.0f} forwards/s, {dt:.3f} ms per forward")
pr = cProfile.Profile(); pr.enable(); loop(1000); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(45)
pstats.Stats(pr).sort_stats("cumulative").print_stats(30)
