#!/usr/bin/env python3
"""HBM bytes per launch of the bench's kernel classes from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, collected in their
own runs with --kernel-trace only, as MI355X_MICROARCH.md prescribes) -> profiles/pmc_traffic.json, which bench.py reads for
`roofline.traffic` as long as the kernel sources still match the digest stored here.
    bytes = 2 x FETCH_SIZE (gfx950 correction: 128-B requests tallied at 64 B) + WRITE_SIZE, both reported in KB by rocprofv3
usage: pmc_traffic.py <fetch.db> <write.db> <key prefix, e.g. f16x3/f16x3/B32/N1024> [more triples ...]   (on the GPU box; tools/pmc_round.sh)"""
import json
import sqlite3
import sys
from collections import defaultdict
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402

CLASSES = {   # kernel-name substring -> class (first match wins)
    "tail_kernel": "fused_tail+next", "attn_split_kernel": "attention", "attn_dma_kernel": "attention", "attn_kernel": "attention",
    "lse_sweep": "assign_sweeps", "argmax_sweep": "assign_sweeps", "sim_kernel": "sim", "proj_kernel": "proj", "adapt_compact": "compact",
    "prep_kernel": "prep", "rowdot_kernel": "rowdot", "gemm_kernel": "gemm",
}


def per_kernel(db_path, counter):
    db = sqlite3.connect(db_path)
    rows = db.execute("select kernel_name, avg(value), count(*) from counters_collection where counter_name = ? group by kernel_name", (counter,)).fetchall()
    return {k: (v, n) for k, v, n in rows}


def classify(name):
    for sub, cls in CLASSES.items():
        if sub in name:
            return cls
    return None


def main():
    out = {"kernel_source_digest": bench.kernel_source_digest(), "bytes_per_launch": {}, "launches": {},
           "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `python bench.py` (tools/pmc_round.sh, tools/pmc_traffic.py): 2 x FETCH_SIZE + WRITE_SIZE, KB -> bytes, averaged over the launches of a class"}
    args = sys.argv[1:]
    for i in range(0, len(args), 3):
        fetch, write, prefix = per_kernel(args[i], "FETCH_SIZE"), per_kernel(args[i + 1], "WRITE_SIZE"), args[i + 2]
        tot, cnt = defaultdict(float), defaultdict(int)
        sweep_kernels = set()
        for k, (f, n) in fetch.items():
            cls = classify(k)
            if cls is None or k not in write:
                continue
            tot[cls] += n * (2.0 * f + write[k][0]) * 1024.0
            cnt[cls] += n
            if cls == "assign_sweeps":
                sweep_kernels.add(k)
        for cls in tot:
            # the two sweeps are ONE pass of the class each: bytes per forward = sum over both kernels, not their average
            per = tot[cls] / cnt[cls] * (len(sweep_kernels) if cls == "assign_sweeps" else 1)
            out["bytes_per_launch"][f"{prefix}/{cls}"] = per
            out["launches"][f"{prefix}/{cls}"] = cnt[cls]
    (ROOT / "profiles" / "pmc_traffic.json").write_text(json.dumps(out, indent=1) + "\n")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
