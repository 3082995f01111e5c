#!/bin/bash
# Round 5, call e: ctx half of the tail's activation tile by piecewise LDS-DMA + in-place conversion (tree) against the round-4 kernels (base).
O=gpurun_out/r05e; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$1', round(d['value']), round(d['ms_per_step'],3), {x: round(k[x],3) for x in ('attn_self','attn_cross','fused_tail') if x in k}, d['parity']['index_mismatches'], d['parity']['max_dscore'])"; }
lib() { if [ "$1" = tree ]; then echo $PWD/lightglue_amd/liblightglue_amd.so; else echo $PWD/build_variants/liblightglue_amd_$1.so; fi; }
VARS="${VARS:-base tree}"
for round in 1 2 3; do for v in $VARS; do
  LIGHTGLUE_AMD_LIB=$(lib $v) timeout 90 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-calibration 2>/dev/null | tail -1 | line $v
done; done 2>&1 | tee $O/ab_cfg2.log
N="x published,top half0,end half0 chunks,top half1 (ctx published),end half1 chunks"
for v in dbase ddma; do echo "== $v"; LIGHTGLUE_AMD_LIB=$(lib $v) timeout 120 python tools/tail_diag.py "$N" 0 0,1,3,2,4 2>&1 | tail -7; done | tee $O/phaseA_diag.log
for v in base tree; do echo "== $v"; LIGHTGLUE_AMD_LIB=$(lib $v) timeout 120 python tools/tail_timing.py f16x3 5 2>&1 | grep -E "phaseA|LN|GELU0|phaseB|epilogue|total"; done | tee $O/stamps.log
timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -n 4 > $O/tests_tree.log 2>&1; tail -3 $O/tests_tree.log
for v in base tree; do echo "== $v"; LIGHTGLUE_AMD_LIB=$(lib $v) timeout 90 python tools/bench_configs.py "#3' " "#5' " 2>&1 | grep "^| #"; done | tee $O/ab_configs.log
