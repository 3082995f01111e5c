#!/bin/bash
# Time several builds of the library in ONE gpurun call (box-to-box variance is +-5-10 %):
#   tools/ab.sh "lib1.so lib2.so ..." [bench args]
LIBS=$1; shift
for round in 1 2; do
  for lib in $LIBS; do
    LIGHTGLUE_AMD_LIB=$PWD/$lib timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$lib', round(d['value']), round(d['ms_per_step'],3), {x: round(k[x],3) for x in ('attn_self','attn_cross','fused_tail','gemm_qkv_self','gemm_qkv_cross','assign') if x in k})"
  done
done
