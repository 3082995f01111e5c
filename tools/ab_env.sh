#!/bin/bash
# A/B of engine environment switches in ONE gpurun call:   tools/ab_env.sh "VAR=a VAR=b ..." [bench args]
SETTINGS=$1; shift
for round in 1 2; do
  for kv in $SETTINGS; do
    env $kv timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$kv', round(d['value']), round(d['ms_per_step'],3), {x: round(k[x],3) for x in ('attn_self','attn_cross','fused_tail','assign') if x in k})"
  done
done
