#!/bin/bash
# Round-4 call aa (last): kernel trace of cfg #2, the config table and the tail stamps on the FINAL sources (digest 3e075215b93c1eb6), then the two opt-in bench lines
O=gpurun_out/r04aa; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-calibration > $O/trace.log 2>&1
python tools/rocpd_stats.py $(find $O/trace -name "*.db" | head -1) $O/kernel_trace.md | head -14
find $O -name "*.db" -delete
( python tools/tail_timing.py f16x3 1; python tools/tail_wall.py ) 2>&1 | grep -v amdgpu.ids | tee $O/tail_timing.log | tail -12
timeout 150 python tools/bench_configs.py 2>&1 | grep -v amdgpu.ids | tail -12; cp gpurun_out/configs.md $O/configs.md
timeout 60 python bench.py --attention fp16 --no-cpu-baseline > $O/bench_fast_attention.json 2>/dev/null
timeout 60 python bench.py --recipe D --no-cpu-baseline > $O/bench_recipe_d.json 2>/dev/null
python - <<'PY'
import json
for f in ("bench_fast_attention.json", "bench_recipe_d.json"):
    try:
        d = json.loads(open("gpurun_out/r04aa/" + f).read().strip().splitlines()[-1]); print(f, round(d["value"]), d["parity"])
    except Exception as e: print(f, "missing", e)
PY
