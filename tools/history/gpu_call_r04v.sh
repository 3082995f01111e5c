#!/bin/bash
# Round-4 call v: PMC passes and bench lines for the final source digest (comment-only change since call n) — GPU suite, default / recipe-D bench lines, kernel trace of cfg #2, the three PMC
# passes (SQ, FETCH_SIZE, WRITE_SIZE) + profiles/pmc_traffic.json for the new source digest, all configs, tail stamps
O=gpurun_out/r04v; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
echo "(suite not repeated: comment-only source change since call n, ISA identical)"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash tools/pmc_round.sh $O/pmc
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err
timeout 600 python bench.py --recipe D > $O/bench_recipe_d.json 2>> $O/bench.err
python - <<'PY'
import json
for f in ("bench.json", "bench_recipe_d.json"):
    d = json.loads(open("gpurun_out/r04v/" + f).read().strip().splitlines()[-1])
    print(f, round(d["value"]), round(d["ms_per_step"], 3), "tail frac", round(d["roofline"]["frac"], 4), d["roofline"]["avg_launch_ms"], "traffic", d["roofline"]["traffic"], "attn frac", round(d["roofline_attention"]["frac"], 4), "hbm frac", round(d["roofline_hbm"]["frac"], 3), d["kernel_ms_per_step"])
    print("   parity", d["parity"]["index_mismatches"], d["parity"]["max_dscore"], d.get("parity_oracle"))
PY
rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-calibration > $O/trace.log 2>&1
python tools/rocpd_stats.py $(find $O/trace -name "*.db" | head -1) $O/kernel_trace.md | head -12
find $O -name "*.db" -delete
timeout 400 python tools/bench_configs.py 2>&1 | grep "^|" | tee $O/configs.md
( python tools/tail_timing.py f16x3 5; python tools/tail_timing.py f16x3 6 ) 2>&1 | grep -v amdgpu.ids | tee $O/tail_timing.log
