"""bench.py's CPU leg (oracle/cpu_leg.py, its own process): core selection, thread pinning, the JSON / npz contract with the parent."""
import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "oracle"))
import cpu_leg  # noqa: E402


def test_core_choice_stays_inside_the_mask_and_avoids_cpu0():
    allowed = sorted(os.sched_getaffinity(0))
    want = max(1, min(4, len(allowed) - 1))
    cores, where = cpu_leg.choose_cores(want)
    assert len(cores) == want and set(cores) <= set(allowed) and len(set(cores)) == want
    if len(allowed) > want:
        assert 0 not in cores
    assert "physical cores" in where
    assert cpu_leg._cpulist("0-3,8,10-11") == [0, 1, 2, 3, 8, 10, 11]


def test_cpu_leg_process_contract(tmp_path):
    out = tmp_path / "leg.npz"
    p = subprocess.run([sys.executable, str(ROOT / "oracle" / "cpu_leg.py"), "--n", "96", "--m", "80", "--threads", "2", "--budget", "2", "--max-pairs", "2", "--rounds", "3",
                        "--no-reference", "--out", str(out)], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads(p.stdout.strip().splitlines()[-1])
    assert d["kind"] == "port" and d["unit"] == "image-pairs/s" and d["cores"] == len(d["cpus"]) <= 2 and d["value"] > 0
    assert len(d["rounds_pairs_per_s"]) == 3 and d["round_spread"] >= 0 and len(d["round_median_forward_ms"]) == 3 and len(d["forward_ms_p10_p50_p90"]) == 3
    done, total = (int(x) for x in d["threads_pinned"].split(" ")[0::2][:2])
    assert done == total, d["threads_pinned"]                     # EVERY thread of the process sits inside the chosen cores
    assert set(d["cfg1_n512_b1"]) == {"1 thread(s)", f"{d['cores']} thread(s)"} or d["cores"] == 1
    z = np.load(out)
    k = int(z["pairs"])
    assert k == d["pairs_per_round"] and z["matches0_0"].shape[-1] == 96 and z["matches1_0"].shape[-1] == 80
