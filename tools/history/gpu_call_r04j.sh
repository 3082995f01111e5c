#!/bin/bash
# Round-4 call j: (1) timing ablations of the tail's phase A (wrong results by construction): no weight stream / no LDS fragment reads / neither, read off the
# phase stamps and the tail class time; (2) split attention as 4 waves x 32 rows (QT = 2) against 8 waves x 16 rows, after the first-tile wait fix.
O=gpurun_out/r04j; mkdir -p $O
export TMPDIR=/tmp
NEW=lightglue_amd/liblightglue_amd.so
for v in "" ablw abll ablwl; do
  lib=${v:+build_variants/liblightglue_amd_$v.so}; lib=${lib:-$NEW}
  echo "== $lib" | tee -a $O/ablation.log
  LIGHTGLUE_AMD_LIB=$PWD/$lib timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print(round(d['value']), round(d['ms_per_step'],3), {x: round(k[x],3) for x in ('attn_self','attn_cross','fused_tail') if x in k})" | tee -a $O/ablation.log
  LIGHTGLUE_AMD_LIB=$PWD/$lib timeout 300 python tools/tail_timing.py f16x3 6 2>&1 | grep -E "phaseA|LN|GELU0|phaseB|epilogue|total|proj" | tee -a $O/ablation.log
done
Q=build_variants/liblightglue_amd_qt2.so
for round in 1 2; do for lib in $NEW $Q; do
  LIGHTGLUE_AMD_LIB=$PWD/$lib timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$lib', round(d['value']), round(d['ms_per_step'],3), {x: round(k[x],3) for x in ('attn_self','attn_cross','fused_tail') if x in k}, d['parity']['index_mismatches'], d['parity']['max_dscore'])"
done; done 2>&1 | tee $O/ab_qt2.log
for lib in $NEW $Q; do
  echo "== $lib" | tee -a $O/ab_qt2_configs.log
  LIGHTGLUE_AMD_LIB=$PWD/$lib timeout 600 python tools/bench_configs.py "#3' " "#4 " "#5' " 2>&1 | grep "^|" | tee -a $O/ab_qt2_configs.log
done
