#!/bin/bash
# Round 5, call k: persistent split attention (engine option attn_persist = workgroups per CU on dense launches) against the one-unit-per-workgroup kernel: bit identity, cfg #2, cfg #4.
O=gpurun_out/r05k; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_round5.py -q -x -k persistent > $O/tests.log 2>&1; tail -3 $O/tests.log
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$1', round(d['value'],1), round(d['ms_per_step'],3), {x: round(k[x],3) for x in ('attn_self','attn_cross','fused_tail') if x in k}, (d['parity'] or {}).get('index_mismatches'))"; }
for round in 1 2 3; do for v in 0 2 1; do
  LG_BENCH_OPTS="attn_persist=$v" timeout 90 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-calibration --no-gather-probe 2>/dev/null | tail -1 | line persist$v
done; done 2>&1 | tee $O/ab_cfg2.log
for v in 0 2; do LG_BENCH_OPTS="attn_persist=$v" timeout 200 python bench.py --config 4 --steps 4 --warmup 2 --no-cpu-baseline --no-calibration --no-gather-probe 2>/dev/null | tail -1 | line cfg4_persist$v; done 2>&1 | tee $O/ab_cfg4.log
