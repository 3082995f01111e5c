#!/usr/bin/env python3
"""Time the UNMODIFIED reference (CPU fp32, B = 1, pruning / early stop off) and the two forms of the port (oracle/: numpy = the
checker, torch-kernel backend = bench.py's timed cpu_baseline leg) side by side in the build container, at N = M = 512 and 1024 with
1 and 8 threads -> profiles/r03_cpu_reference.md.  The torch-backend ratios feed bench.py's PORT_OVER_REFERENCE_TIME (the GPU box has
no /root/reference, so bench.py times the port there and reports `reference_estimate_pairs_per_s = value x ratio` next to it).
usage: python tools/cpu_reference_table.py > profiles/r03_cpu_reference.md"""
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from lightglue_amd import synthetic  # noqa: E402
from oracle import lightglue_oracle as O  # noqa: E402
from threadpoolctl import threadpool_limits  # noqa: E402

def _time_reference(sd, n, threads, reps, warm=2):
    """The unmodified reference module (CPU fp32, /root/reference/benchmark.py:18-43's method: warm-up, then timed repetitions of forward)."""
    import importlib.util
    import torch
    spec = importlib.util.spec_from_file_location("lg_ref", "/root/reference/lightglue/lightglue.py")
    lg = importlib.util.module_from_spec(spec); spec.loader.exec_module(lg)
    torch.set_grad_enabled(False)
    model = lg.LightGlue(features=None, depth_confidence=-1, width_confidence=-1).eval()
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    td = {k: {kk: torch.from_numpy(vv) for kk, vv in v.items()} for k, v in synthetic.make_batch(1, 1, n, n).items()}
    old = torch.get_num_threads(); torch.set_num_threads(threads)
    try:
        for _ in range(warm):
            model(td)
        t0 = time.perf_counter()
        for _ in range(reps):
            model(td)
        return (time.perf_counter() - t0) / reps
    finally:
        torch.set_num_threads(old)


sd = synthetic.make_state_dict(0, recipe="A")
conf = O.make_conf(depth_confidence=-1, width_confidence=-1)
import torch  # noqa: E402

print(f"# r03 — reference CPU path vs the port (oracle/), build container ({bench._cpu_model()}, {os.cpu_count()} logical cores)\n")
print("`tools/cpu_reference_table.py`: unmodified `/root/reference/lightglue/lightglue.py` loaded standalone, CPU fp32, B = 1, recipe-A weights,")
print("pruning / early stop off, 2 warm-up + 5 timed forwards (best of 2 passes); port = `oracle/lightglue_oracle.py`: `numpy` = the checker")
print("(numpy / OpenBLAS under `threadpool_limits`), `torch` = the same restatement on torch's CPU kernels (`backend=\"torch\"`, `bench.py`'s")
print("timed cpu_baseline leg, `torch.set_num_threads`).\n")
print("| N = M | threads | reference ms | reference pairs/s | numpy port ms | numpy / reference time | torch-backend port ms | torch-backend pairs/s | torch-backend / reference time |")
print("|---|---|---|---|---|---|---|---|---|")
ratios = {}
for n in (512, 1024):
    for th in (1, 8):
        tr = min(_time_reference(sd, n, th, reps=5) for _ in range(2))
        data = synthetic.make_batch(1, 1, n, n)
        with threadpool_limits(limits=th):
            O.forward(sd, conf, data)
            t0 = time.perf_counter()
            for _ in range(3):
                O.forward(sd, conf, data)
            tp = (time.perf_counter() - t0) / 3
        old = torch.get_num_threads(); torch.set_num_threads(th)
        tt = float("inf")
        for _ in range(2):
            O.forward(sd, conf, data, backend="torch")
            t0 = time.perf_counter()
            for _ in range(5):
                O.forward(sd, conf, data, backend="torch")
            tt = min(tt, (time.perf_counter() - t0) / 5)
        torch.set_num_threads(old)
        ratios[(n, th)] = tt / tr
        print(f"| {n} | {th} | {tr * 1e3:.1f} | {1 / tr:.2f} | {tp * 1e3:.1f} | {tp / tr:.2f} | {tt * 1e3:.1f} | {1 / tt:.2f} | {tt / tr:.2f} |", flush=True)
print("\nPORT_OVER_REFERENCE_TIME =", {f"{th} thread{'s' if th > 1 else ''}": {f"N={n}": round(ratios[(n, th)], 2) for n in (512, 1024)} for th in (1, 8)})
