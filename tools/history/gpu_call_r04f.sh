#!/bin/bash
# Round-4 call f: second A/B of the ISA-level tail changes, one box.  V3 = early ctx + laundered ctx conversion + pinned phase-B ring + rotary L2 touch
# (best of call e); a = V3 without launder / early ctx (hipcc's hoisted conversion back); b = V3 + rotary rows fetched once, ahead of the projection's
# MFMA loops; c = a + the same.  All builds carry the 4-rows-in-flight assignment sweeps (class `assign`).
O=gpurun_out/r04f; mkdir -p $O
export TMPDIR=/tmp
BASE=build_variants/liblightglue_amd_base.so
V3=build_variants/liblightglue_amd_earlypinpf.so; A=build_variants/liblightglue_amd_a.so; B=build_variants/liblightglue_amd_b.so; C=build_variants/liblightglue_amd_c.so
LIGHTGLUE_AMD_LIB=$PWD/$C timeout 1500 python -m pytest tests -m gpu -q -x > $O/gputests.log 2>&1; grep -E 'passed|failed|error' $O/gputests.log | tail -4
for round in 1 2 3; do for lib in $BASE $V3 $A $B $C; do
  LIGHTGLUE_AMD_LIB=$PWD/$lib timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$lib', round(d['value']), round(d['ms_per_step'],3), {x: round(k[x],3) for x in ('attn_self','attn_cross','fused_tail','gemm_qkv_self','sim','assign') if x in k}, d['parity']['index_mismatches'], d['parity']['max_dscore'])"
done; done 2>&1 | tee $O/ab_cfg2.log
for lib in $V3 $A $B $C; do
  echo "== $lib" | tee -a $O/ab_configs.log
  LIGHTGLUE_AMD_LIB=$PWD/$lib timeout 600 python tools/bench_configs.py "#3' " "#3b " "#5' " 2>&1 | grep "^|" | tee -a $O/ab_configs.log
done
