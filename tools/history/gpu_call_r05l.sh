#!/bin/bash
# Round 5, call l: persistent split attention with the CU's second workgroup started a fraction of a tile late (attn_persist_delay cycles), against the one-unit kernel.
O=gpurun_out/r05l; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$1', round(d['value'],1), round(d['ms_per_step'],3), {x: round(k[x],3) for x in ('attn_self','attn_cross','fused_tail') if x in k}, (d['parity'] or {}).get('index_mismatches'))"; }
for round in 1 2; do
  LG_BENCH_OPTS="attn_persist=0" timeout 90 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-calibration --no-gather-probe 2>/dev/null | tail -1 | line persist0
  for dly in 0 1000 2000 3000 5000; do
    LG_BENCH_OPTS="attn_persist=2 attn_persist_delay=$dly" timeout 90 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-calibration --no-gather-probe 2>/dev/null | tail -1 | line persist2_delay$dly
  done
done 2>&1 | tee $O/ab_cfg2.log
