#!/bin/bash
# Round 6, call a: ping-pong split attention (attn_pp 1..4) and phase / no priority on the one-barrier kernel (5, 6) against HEAD's kernel (0), one library.
O=gpurun_out/r06a; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python tools/attn_variant_check.py > $O/variant_check.log 2>&1; tail -12 $O/variant_check.log
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$1', round(d['value'],1), round(d['ms_per_step'],3), {x: round(k[x],3) for x in ('attn_self','attn_cross','fused_tail') if x in k}, (d['parity'] or {}).get('index_mismatches'), (d['parity'] or {}).get('max_dscore'))"; }
for round in 1 2; do for v in 0 1 2 3 4 5 6; do
  LG_BENCH_OPTS="attn_pp=$v" timeout 120 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-calibration --no-gather-probe 2>/dev/null | tail -1 | line pp$v
done; done 2>&1 | tee $O/ab_cfg2.log
for v in 0 1 2 3 4; do
  LG_BENCH_OPTS="attn_pp=$v" timeout 200 python bench.py --config 4 --steps 4 --warmup 2 --no-cpu-baseline --no-calibration --no-gather-probe 2>/dev/null | tail -1 | line cfg4_pp$v
done 2>&1 | tee $O/ab_cfg4.log
