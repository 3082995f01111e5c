#!/bin/bash
# A/B of engine options in one process family:  tools/ab_opt.sh "key=val key=val ..."   (each setting = one bench run, two rounds)
for round in 1 2; do
  for kv in "$@"; do
    LG_BENCH_OPTS="$kv" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$kv', round(d['value']), round(d['ms_per_step'],3), d['parity'] and (d['parity']['index_mismatches'], round(d['parity']['max_dscore'],6)), {x: round(k[x],3) for x in ('attn_self','attn_cross','fused_tail','gemm_qkv_self','assign') if x in k})"
  done
done
