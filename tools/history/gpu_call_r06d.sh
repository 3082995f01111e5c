#!/bin/bash
# Round 6, call d: gather projection (no compaction launch) — bit-identity against the in-place compaction, the adaptive fixtures, A/B on cfg #3' / #5'
O=gpurun_out/r06d; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round6.py -x -q -k "gather or verify" > $O/tests_new.log 2>&1; tail -15 $O/tests_new.log
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_properties.py -x -q -n 4 > $O/tests_parity.log 2>&1; tail -5 $O/tests_parity.log
timeout 900 python -m pytest tests/test_gpu_round5.py -x -q -n 4 -k "adaptive or extension" > $O/tests_r5.log 2>&1; tail -5 $O/tests_r5.log
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$1', round(d['value'],1), round(d['ms_per_step'],3), {x: round(k[x],3) for x in k})"; }
for round in 1 2 3; do for v in 0 1; do for c in 3 5; do
  LG_BENCH_OPTS="adapt_gather=$v" timeout 200 python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline --no-calibration --no-gather-probe 2>/dev/null | tail -1 | line cfg${c}_gather$v
done; done; done 2>&1 | tee $O/ab_adaptive.log
