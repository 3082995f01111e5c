#!/bin/bash
# Round-4 call g: LDS fragment reads of phase B and of the projection one chunk ahead of their MFMAs + biases ahead of the projection loops (NEW, the tree)
# against HEAD (base); one2 / one3 = NEW + both halves of the activation tile loaded up front (one barrier), 16 chunks unrolled, weight ring 2 / 3 deep.
O=gpurun_out/r04g; mkdir -p $O
export TMPDIR=/tmp
BASE=build_variants/liblightglue_amd_base.so; NEW=lightglue_amd/liblightglue_amd.so; O2=build_variants/liblightglue_amd_one2.so; O3=build_variants/liblightglue_amd_one3.so
timeout 1500 python -m pytest tests -m gpu -q -x > $O/gputests.log 2>&1; grep -E 'passed|failed|error' $O/gputests.log | tail -4
for round in 1 2 3; do for lib in $BASE $NEW $O2 $O3; do
  LIGHTGLUE_AMD_LIB=$PWD/$lib timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$lib', round(d['value']), round(d['ms_per_step'],3), {x: round(k[x],3) for x in ('attn_self','attn_cross','fused_tail','gemm_qkv_self','sim','assign') if x in k}, d['parity']['index_mismatches'], d['parity']['max_dscore'])"
done; done 2>&1 | tee $O/ab_cfg2.log
for lib in $BASE $NEW $O3; do
  echo "== $lib" | tee -a $O/ab_configs.log
  LIGHTGLUE_AMD_LIB=$PWD/$lib timeout 600 python tools/bench_configs.py "#3' " "#3b " "#4 " "#5' " 2>&1 | grep "^|" | tee -a $O/ab_configs.log
done
