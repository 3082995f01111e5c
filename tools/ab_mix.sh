#!/bin/bash
# A/B of (library, engine options) pairs in one gpurun call:  tools/ab_mix.sh "lib|opts" "lib|opts" ...
for round in 1 2; do
  for spec in "$@"; do
    lib=${spec%%|*}; opts=${spec#*|}
    LIGHTGLUE_AMD_LIB=$PWD/$lib LG_BENCH_OPTS="$opts" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$spec', round(d['value']), round(d['ms_per_step'],3), d['parity'] and (d['parity']['index_mismatches'], round(d['parity']['max_dscore'],6)), {x: round(k[x],3) for x in ('attn_self','attn_cross','fused_tail','gemm_qkv_self','assign') if x in k})"
  done
done
