#!/bin/bash
# experiment: start-stagger of the two co-resident 4-wave tail workgroups (LG_TAIL_VARIANT=1), one gpurun call
run() { env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline $EXTRA 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$*', '$EXTRA', round(d['value']), round(d['ms_per_step'],3), {x: round(k[x],3) for x in ('attn_self','attn_cross','fused_tail','gemm_qkv_self','gemm_qkv_cross','assign') if x in k})"; }
for round in 1 2; do
  EXTRA="" run LG_X=0
  EXTRA="--no-fuse-next"
  run LG_X=0
  for st in 0 1000 2000 3000 4000 6000; do run LG_TAIL_VARIANT=1 LG_TAIL_STAGGER=$st; done
done
