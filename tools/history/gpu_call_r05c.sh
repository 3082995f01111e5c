#!/bin/bash
# Round 5, call c: where phase A of the tail spends its cycles.  Diagnostic builds (patched COPIES of csrc/, extra shader-clock stamps into TAILDBG2):
# dbase = the round-4 two-halves phase A, dstream = the streamed rounds of call b.
O=gpurun_out/r05c; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2; do
echo "== dbase (round-4 phase A): start -> x published -> top of half 0 (ctx requested, hoisted waits) -> end of half-0 chunks -> top of half 1 (ctx published) -> end"
LIGHTGLUE_AMD_LIB=$PWD/build_variants/liblightglue_amd_dbase.so timeout 120 python tools/tail_diag.py "x published,top half0,top half1 (ctx published),end half0 chunks,end half1 chunks" 2>&1 | tail -8
echo "== dstream: start -> round 0 published -> top of rounds 0,2,4,6 -> end"
LIGHTGLUE_AMD_LIB=$PWD/build_variants/liblightglue_amd_dstream.so timeout 120 python tools/tail_diag.py "round0 published,top r0,top r2,top r4,top r6" 2>&1 | tail -8
done | tee $O/phaseA_diag.log
