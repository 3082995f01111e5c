#!/bin/bash
# Round 6, call n: ring attention diagnostics: counters (1) vs the same 32-key ring with a barrier per tile (2) vs the one-barrier 64-key kernel (0)
O=gpurun_out/r06n; rm -rf $O; mkdir -p $O
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$1', round(d['value'],1), round(d['ms_per_step'],3), {x: round(k[x],3) for x in ('attn_self','attn_cross','fused_tail') if x in k}, 'W', (d.get('power') or {}).get('board_power_w_median'), 'sclk', (d.get('power') or {}).get('sclk_mhz_median'), (d['parity'] or {}).get('index_mismatches'), (d['parity'] or {}).get('max_dscore'))"; }
for round in 1 2; do for v in 0 1 2; do
  LG_BENCH_OPTS="attn_ring=$v" timeout 120 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-gather-probe 2>/dev/null | tail -1 | line ring$v
done; done 2>&1 | tee $O/ab_cfg2.log
