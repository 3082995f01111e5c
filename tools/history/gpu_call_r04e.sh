#!/bin/bash
# Round-4 call e: waits found in the compiled ISA (tools/isa_wait_distance.py, tools/check_isa.py) —
#   attention: the tile-publishing wait as a BUILTIN (hipcc no longer waits for the next tile's DMAs in the peeled first tile),
#   tail: ctx rows laundered (hipcc had hoisted their conversion in front of the MFMA loop), head weights loaded without a branch,
#   variants: early = ctx rows requested right behind the x rows, earlypin = + phase-B ring pinned 3 chunks ahead with the GELU dealt to
#   the chunks by hand, earlypinpf = + rotary rows touched into L2 a GELU step ahead of the fused projection's epilogues.
# A/B against HEAD (base), one box; GPU suite on the most aggressive variant.
O=gpurun_out/r04e; mkdir -p $O
export TMPDIR=/tmp
BASE=build_variants/liblightglue_amd_base.so; NEW=lightglue_amd/liblightglue_amd.so
V1=build_variants/liblightglue_amd_early.so; V2=build_variants/liblightglue_amd_earlypin.so; V3=build_variants/liblightglue_amd_earlypinpf.so
LIGHTGLUE_AMD_LIB=$PWD/$V3 timeout 1500 python -m pytest tests -m gpu -q -x > $O/gputests.log 2>&1; grep -E 'passed|failed|error' $O/gputests.log | tail -4
for round in 1 2; do for lib in $BASE $NEW $V1 $V2 $V3; do
  LIGHTGLUE_AMD_LIB=$PWD/$lib timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$lib', round(d['value']), round(d['ms_per_step'],3), {x: round(k[x],3) for x in ('attn_self','attn_cross','fused_tail','gemm_qkv_self','sim','assign') if x in k}, d['parity']['index_mismatches'], d['parity']['max_dscore'])"
done; done 2>&1 | tee $O/ab_cfg2.log
for lib in $BASE $NEW $V2 $V3; do
  echo "== $lib" | tee -a $O/ab_configs.log
  LIGHTGLUE_AMD_LIB=$PWD/$lib timeout 600 python tools/bench_configs.py "#3' " "#3b " "#4 " "#5' " 2>&1 | grep "^|" | tee -a $O/ab_configs.log
done
( LIGHTGLUE_AMD_LIB=$PWD/$V3 timeout 300 python tools/tail_timing.py f16x3 5; LIGHTGLUE_AMD_LIB=$PWD/$V3 timeout 300 python tools/tail_timing.py f16x3 6 ) 2>&1 | grep -v amdgpu.ids | tee $O/tail_timing.log
