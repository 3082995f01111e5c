#!/bin/bash
# Round-3 call k (generalised tail interleave): GPU suite + smoke, kernel trace, PMC passes (-> profiles/pmc_traffic.json), default bench
# with the fresh traffic file, tail phase stamps + CU-slot check.
O=gpurun_out/r03r; mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests -m gpu -q > $O/gputests.log 2>&1; grep -E 'passed|failed|error' $O/gputests.log | tail -4
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-calibration > $O/trace.log 2>&1
python tools/rocpd_stats.py $(find $O/trace -name "*.db" | head -1) $O/kernel_trace.md | head -8
bash tools/pmc_round.sh $O/pmc > $O/pmc_round.log 2>&1; tail -4 $O/pmc_round.log
find $O -name "*.db" -delete
python bench.py > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err
( python tools/tail_timing.py f16x3 1; python tools/tail_wall.py ) 2>&1 | grep -v amdgpu.ids | tee $O/tail_timing.log
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03r/bench.json").read().strip().splitlines()[-1])
print(round(d["value"]), round(d["ms_per_step"], 3), "sync", d["value_synchronous_forward"], "tail frac", round(d["roofline"]["frac"], 4), "traffic", d["roofline"]["traffic"], "attn frac", round(d["roofline_attention"]["frac"], 4),
      "hbm frac", round(d["roofline_hbm"]["frac"], 3), "cpu", (d.get("cpu_baseline") or {}).get("value"), d["kernel_ms_per_step"])
print("   parity", d["parity"], d.get("parity_oracle"))
PY
