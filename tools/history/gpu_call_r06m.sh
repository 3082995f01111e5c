#!/bin/bash
# Round 6, call m: the barrier-free ring attention (attn_ring = 1): correctness against the one-barrier kernel, then time / power
O=gpurun_out/r06m; rm -rf $O; mkdir -p $O
timeout 600 python tools/experiments/attn_ring_check.py > $O/ring_check.log 2>&1; tail -14 $O/ring_check.log
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$1', round(d['value'],1), round(d['ms_per_step'],3), {x: round(k[x],3) for x in ('attn_self','attn_cross','fused_tail') if x in k}, 'W', (d.get('power') or {}).get('board_power_w_median'), 'sclk', (d.get('power') or {}).get('sclk_mhz_median'), (d['parity'] or {}).get('index_mismatches'), (d['parity'] or {}).get('max_dscore'))"; }
for round in 1 2 3; do for v in 0 1; do
  LG_BENCH_OPTS="attn_ring=$v" timeout 120 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-gather-probe 2>/dev/null | tail -1 | line ring$v
done; done 2>&1 | tee $O/ab_cfg2.log
for v in 0 1; do LG_BENCH_OPTS="attn_ring=$v" timeout 200 python bench.py --config 4 --steps 4 --warmup 2 --no-cpu-baseline --no-calibration --no-gather-probe 2>/dev/null | tail -1 | line cfg4_ring$v; done 2>&1 | tee $O/ab_cfg4.log
