#!/bin/bash
# Round 5, call d: does de-synchronising the CUs shorten the tail's activation prologue?  dstag = dbase + half of the first 256 workgroups (ids with bit 3 set)
# spin ~half a tile before they start; statistics over the workgroups of the LATER rounds (id >= 256) only, dbase the same way.
O=gpurun_out/r05d; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
N="x published,top half0 (ctx waited),end half0 chunks,top half1 (ctx published),end half1 chunks"
for rep in 1 2; do
for v in dbase dstag; do echo "== $v, workgroups >= 256"; LIGHTGLUE_AMD_LIB=$PWD/build_variants/liblightglue_amd_$v.so timeout 120 python tools/tail_diag.py "$N" 256 0,1,3,2,4 2>&1 | tail -7
  LIGHTGLUE_AMD_LIB=$PWD/build_variants/liblightglue_amd_$v.so timeout 120 python tools/tail_timing.py f16x3 5 2>&1 | grep -E "phaseA|LN|GELU0|phaseB|epilogue|total"; done
done | tee $O/stagger_diag.log
