#!/bin/bash
# Round-4 call o: fewer VALU instructions in exposed places of the tail — mix-form split (4 instead of 8 instructions per pair) in the activation prologue (NEW),
# + residual rows by buffer load / store (xbuf) — against HEAD
O=gpurun_out/r04o; mkdir -p $O
export TMPDIR=/tmp
BASE=build_variants/liblightglue_amd_base.so; NEW=lightglue_amd/liblightglue_amd.so; X=build_variants/liblightglue_amd_xbuf.so
for round in 1 2 3 4; do for lib in $BASE $NEW $X; do
  LIGHTGLUE_AMD_LIB=$PWD/$lib timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$lib', round(d['value']), round(d['ms_per_step'],3), {x: round(k[x],3) for x in ('attn_self','attn_cross','fused_tail','gemm_qkv_self') if x in k}, d['parity']['index_mismatches'], d['parity']['max_dscore'])"
done; done 2>&1 | tee $O/ab_cfg2.log
for lib in $BASE $X; do
  echo "== $lib" | tee -a $O/ab_configs.log
  LIGHTGLUE_AMD_LIB=$PWD/$lib timeout 600 python tools/bench_configs.py "#3' " "#5' " 2>&1 | grep "^|" | tee -a $O/ab_configs.log
done
