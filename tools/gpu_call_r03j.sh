#!/bin/bash
# Round-3 call j (tail tile interleave adopted): GPU suite + smoke, default bench, kernel trace, PMC passes (-> profiles/pmc_traffic.json),
# default bench again with the fresh traffic file, recipe-D bench.
O=gpurun_out/r03j; mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests -m gpu -q > $O/gputests.log 2>&1; grep -E 'passed|failed|error' $O/gputests.log | tail -4
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-calibration > $O/trace.log 2>&1
python tools/rocpd_stats.py $(find $O/trace -name "*.db" | head -1) $O/kernel_trace.md | head -16
bash tools/pmc_round.sh $O/pmc
find $O -name "*.db" -delete
python bench.py > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err
python bench.py --recipe D --no-cpu-baseline > $O/bench_recipe_d.json 2>> $O/bench.err
python - <<'PY'
import json
for f in ("bench.json", "bench_recipe_d.json"):
    d = json.loads(open("gpurun_out/r03j/" + f).read().strip().splitlines()[-1])
    print(f, round(d["value"]), round(d["ms_per_step"], 3), "sync", d["value_synchronous_forward"], "tail frac", round(d["roofline"]["frac"], 4), "traffic", d["roofline"]["traffic"], "attn frac", round(d["roofline_attention"]["frac"], 4),
          "hbm frac", round(d["roofline_hbm"]["frac"], 3), "cpu", (d.get("cpu_baseline") or {}).get("value"), d["kernel_ms_per_step"])
    print("   parity", d["parity"], d.get("parity_oracle"))
PY
