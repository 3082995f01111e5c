#!/bin/bash
# Round 5, call f: next-tile L2 touch from the fused projection (pf1: pass 0, pf2: pass 1) and the scalar GELU on top of the ctx-DMA tail (tree); base = round-4 kernels.
O=gpurun_out/r05f; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$1', round(d['value']), round(d['ms_per_step'],3), {x: round(k[x],3) for x in ('attn_self','attn_cross','fused_tail') if x in k}, d['parity']['index_mismatches'], d['parity']['max_dscore'])"; }
lib() { if [ "$1" = tree ]; then echo $PWD/lightglue_amd/liblightglue_amd.so; else echo $PWD/build_variants/liblightglue_amd_$1.so; fi; }
VARS="${VARS:-base tree pf pfx}"
for round in 1 2; do for v in $VARS; do
  LIGHTGLUE_AMD_LIB=$(lib $v) timeout 90 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-calibration 2>/dev/null | tail -1 | line $v
done; done 2>&1 | tee $O/ab_cfg2.log
for v in tree pf pfx; do echo "== $v"; LIGHTGLUE_AMD_LIB=$(lib $v) timeout 120 python tools/tail_timing.py f16x3 5 2>&1 | grep -E "phaseA|LN|GELU0|phaseB|epilogue|total|proj"; done | tee $O/stamps.log
