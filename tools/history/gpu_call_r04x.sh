#!/bin/bash
# Round-4 call x: smoke, kernel trace of cfg #2, all configs and the tail stamps on the FINAL sources (the PMC passes and the bench line of this digest are call w's)
O=gpurun_out/r04x; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-calibration > $O/trace.log 2>&1
python tools/rocpd_stats.py $(find $O/trace -name "*.db" | head -1) $O/kernel_trace.md | head -12
find $O -name "*.db" -delete
timeout 400 python tools/bench_configs.py 2>&1 | grep "^|" | tee $O/configs.md
( python tools/tail_timing.py f16x3 5; python tools/tail_timing.py f16x3 6 ) 2>&1 | grep -v amdgpu.ids > $O/tail_timing.log; grep -E "phaseA|total" $O/tail_timing.log
