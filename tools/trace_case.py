#!/usr/bin/env python3
"""A few forwards of one named case, for `rocprofv3 --kernel-trace --stats -- python tools/trace_case.py <case>`."""
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import gpu_util
from lightglue_amd import synthetic as synth
case = sys.argv[1]
if case == "adaptive_b16_n2048":      # cfg #3', recipe C: mixed stop depths, pruning active
    sd = synth.make_state_dict(0, recipe="C"); model = gpu_util.make_model(sd, "f16x3"); data = synth.make_batch(1, 16, 2048, 2048)
elif case == "b1_n1024":              # single pair, non-adaptive: the small-grid kernel shapes
    sd = synth.make_state_dict(0, recipe="A"); model = gpu_util.make_model(sd, "f16x3", depth_confidence=-1, width_confidence=-1); data = synth.make_batch(1, 1, 1024, 1024)
elif case == "cfg4_b32_n4096":        # cfg #4: DISK 128-d, N = M = 4096, the 32-pair shard one of the 8 GPUs runs
    sd = synth.make_state_dict(3, recipe="A", input_dim=128); model = gpu_util.make_model(sd, "f16x3", depth_confidence=-1, width_confidence=-1, input_dim=128)
    data = synth.make_batch(301, 32, 4096, 4096, 128)
else:
    raise SystemExit("unknown case")
data = gpu_util.to_torch(data)
for _ in range(3 if case.startswith("cfg4") else 10): out = model(data)
torch.cuda.synchronize()
print(case, "stop", out["stop"])
