#!/usr/bin/env python3
"""Host-bound case (B = 1, several lanes): is ONE hipGraphLaunch per forward cheaper on the HOST than the engine's ~46 launches?
Per lane: its own model + stream + a captured graph of forward_raw; the loop replays the lanes' graphs round robin (no result handling: pure issue rate),
against the same loop issuing forward_raw launch by launch on the same streams."""
import sys, time
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import gpu_util
from lightglue_amd import synthetic as synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
for depth in (1, 2, 4, 8):
    sd = synth.make_state_dict(0, recipe="A")
    models, streams, graphs, datas = [], [], [], []
    for k in range(depth):
        m = gpu_util.make_model(sd, "f16x3", depth_confidence=-1, width_confidence=-1); m.track_inplace_weight_edits = False
        d = gpu_util.to_torch(synth.make_batch(1, 1, n, n)); m.reserve(1, n, n)
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            m.forward_raw(d)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            out = m.forward_raw(d)
        models.append(m); streams.append(s); graphs.append((g, out)); datas.append(d)
    def plain(reps):
        for i in range(reps):
            k = i % depth
            with torch.cuda.stream(streams[k]):
                models[k].forward_raw(datas[k])
    def replay(reps):
        for i in range(reps):
            k = i % depth
            with torch.cuda.stream(streams[k]):
                graphs[k][0].replay()
    for name, fn in (("launches", plain), ("graph replay", replay)):
        fn(50); torch.cuda.synchronize()
        t0 = time.perf_counter(); fn(1000); t_issue = time.perf_counter() - t0
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"N={n} lanes={depth} {name:13s}: {1000 / dt:7.0f} forwards/s   host issue {t_issue * 1e3:.3f} us/forward   total {dt * 1e3:.3f} us/forward", flush=True)
