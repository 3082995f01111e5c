#!/bin/bash
# Round 5, call m: fused projection started per wave on its own columns (rotated k order + per-wave ready words in LDS instead of the workgroup barrier) against HEAD.
O=gpurun_out/r05m; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -n 4 > $O/tests_parity.log 2>&1; tail -3 $O/tests_parity.log
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$1', round(d['value'],1), round(d['ms_per_step'],3), {x: round(k[x],3) for x in ('attn_self','attn_cross','fused_tail','gemm_qkv_self') if x in k}, (d['parity'] or {}).get('index_mismatches'), (d['parity'] or {}).get('max_dscore'))"; }
lib() { if [ "$1" = tree ]; then echo $PWD/lightglue_amd/liblightglue_amd.so; else echo $PWD/build_variants/liblightglue_amd_$1.so; fi; }
for round in 1 2 3; do for v in head tree; do
  LIGHTGLUE_AMD_LIB=$(lib $v) timeout 90 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-calibration --no-gather-probe 2>/dev/null | tail -1 | line $v
done; done 2>&1 | tee $O/ab_cfg2.log
for v in head tree; do echo "== $v"; LIGHTGLUE_AMD_LIB=$(lib $v) timeout 120 python tools/tail_timing.py f16x3 5 2>&1 | grep -E "phaseA|LN|GELU0|phaseB|epilogue|total|proj"; done | tee $O/stamps.log
for v in head tree; do echo "== $v"; LIGHTGLUE_AMD_LIB=$(lib $v) timeout 120 python tools/bench_configs.py "#3' " "#4 " "#5' " 2>&1 | grep "^| #"; done | tee $O/ab_configs.log
