#!/usr/bin/env python3
"""GPU diagnostic: per-stage error table of the HIP pipeline vs the oracle for every precision, then
end-to-end parity on the golden cases.  Never asserts; writes gpurun_out/lab.json.  (Test tooling.)"""
import json
import sys
import time
import traceback
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests")); sys.path.insert(0, str(ROOT / "tools"))
import gpu_util  # noqa: E402
import make_golden  # noqa: E402
from conftest import golden_names, load_golden, oracle_conf_for  # noqa: E402
from lightglue_amd import synthetic as synth  # noqa: E402


def main():
    out = {"stages": {}, "golden": {}}
    precisions = sys.argv[1:] or ["fp32", "f16x3", "f16x3/fp16", "bf16", "fp16"]
    sd = synth.make_state_dict(0, recipe="A")
    data = synth.make_batch(7, 2, 200, 160)
    for prec in precisions:
        try:
            t = time.time()
            res = gpu_util.stage_errors(sd, data, prec, dict(depth_confidence=-1, width_confidence=-1))
            out["stages"][prec] = res
            print(f"--- stage errors [{prec}] ({time.time() - t:.1f}s): max|err|, err/rms")
            for k, v in res.items():
                print(f"   {k:18s} {v[0]:.3e} {v[1]:.3e}")
        except Exception:
            traceback.print_exc()
            out["stages"][prec] = "EXC " + traceback.format_exc()[-400:]
    for prec in precisions:
        for name in golden_names():
            meta, gold = load_golden(name)
            case = meta["case"]
            try:
                sd2, data2 = make_golden.case_inputs(case)
                kw = dict(case["conf"])
                if "prune_th" in case:
                    kw["pruning_min_kpts"] = case["prune_th"]
                model = gpu_util.make_model(sd2, prec, **kw)
                o = model(gpu_util.to_torch(data2))
                torch.cuda.synchronize()
                m0 = o["matches0"].cpu().numpy(); s0 = o["matching_scores0"].cpu().numpy()
                stop = o["stop"] if not torch.is_tensor(o["stop"]) else o["stop"].cpu().tolist()
                rec = dict(idx_mismatch0=int((m0 != gold["matches0"]).sum()), idx_mismatch1=int((o["matches1"].cpu().numpy() != gold["matches1"]).sum()),
                           max_dscore=float(np.abs(s0 - gold["matching_scores0"]).max()) if s0.size else 0.0,
                           rms_dscore=float(np.sqrt(np.mean((s0 - gold["matching_scores0"]) ** 2))) if s0.size else 0.0,
                           n=int((gold["matches0"] > -1).sum()), got_n=int((m0 > -1).sum()), stop=stop, ref_stop=gold["stop"].tolist(),
                           prune_mismatch=int((o["prune0"].cpu().numpy().astype(np.float32) != gold["prune0"]).sum()))
            except Exception:
                traceback.print_exc()
                rec = "EXC " + traceback.format_exc()[-400:]
            out["golden"][f"{prec}/{name}"] = rec
            print(f"golden [{prec}] {name:32s} {rec}")
    Path(ROOT / "gpurun_out").mkdir(exist_ok=True)
    (ROOT / "gpurun_out" / "lab.json").write_text(json.dumps(out, indent=1, default=str))


if __name__ == "__main__":
    main()
