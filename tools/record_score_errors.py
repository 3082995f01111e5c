#!/usr/bin/env python3
"""Record max |matching_scores0 - reference| of every fixture for the default precision and the exact-fp32 mode on the CURRENT kernels ->
gpurun_out/recorded_score_errors.json (copy to tests/golden/: tests/test_gpu_parity.py::test_scores_stay_inside_the_recorded_envelope asserts 2 x these).
Re-run after any change that alters the arithmetic (not needed for bit-identical changes)."""
import json
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests")); sys.path.insert(0, str(ROOT / "tools"))
import gpu_util  # noqa: E402
import make_golden  # noqa: E402
from conftest import golden_names, load_golden  # noqa: E402

rows = {}
for name in golden_names():
    meta, gold = load_golden(name)
    case = meta["case"]
    sd, data = make_golden.case_inputs(case)
    kw = dict(case["conf"])
    if "prune_th" in case:
        kw["pruning_min_kpts"] = case["prune_th"]
    rows[name] = {}
    for prec in ("f16x3", "fp32"):
        model = gpu_util.make_model(sd, prec, **kw)
        if case.get("static_lengths"):
            model.static_lengths = list(case["static_lengths"])
        out = model(gpu_util.to_torch(data))
        torch.cuda.synchronize()
        same = out["matches0"].cpu().numpy() == gold["matches0"]
        d = np.abs(out["matching_scores0"].cpu().numpy() - gold["matching_scores0"])
        rows[name][prec] = float(d[same].max()) if same.any() else 0.0
        rows[name][prec + "_flips"] = int((~same).sum())
    print(name, rows[name], flush=True)
digest = __import__("subprocess").run([sys.executable, "-c", "import sys; sys.path.insert(0, '.'); import bench; print(bench.kernel_source_digest())"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
(ROOT / "gpurun_out").mkdir(exist_ok=True)
json.dump({"source": "tools/record_score_errors.py on the GPU: max |matching_scores0 - reference| over the keypoints whose index agrees, per fixture and GPU mode",
           "kernel_source_digest": digest, "max_abs_dscore": rows}, open(ROOT / "gpurun_out" / "recorded_score_errors.json", "w"), indent=1, sort_keys=True)
