#!/bin/bash
# Round-4 call c: chunk-parallel compaction + 1024-thread decide kernel + tail / projection latency hiding (bias in the accumulators, epilogue
# operands fetched under the MFMA loops) against the round-3 build (build_variants/liblightglue_amd_base.so = 8516920's csrc), and the two prepared
# tile-map patches (rot4 = attention units of 4 query tiles round-robin over XCDs + projection residue rotation, rot8 = units of 8), one box.
O=gpurun_out/r04c; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x > $O/gputests.log 2>&1; grep -E 'passed|failed|error' $O/gputests.log | tail -4
BASE=build_variants/liblightglue_amd_base.so; NEW=lightglue_amd/liblightglue_amd.so
R4=build_variants/liblightglue_amd_rot4.so; R8=build_variants/liblightglue_amd_rot8.so
tools/ab.sh "$BASE $NEW $R4 $R8" 2>&1 | tee $O/ab_cfg2.log
for lib in $BASE $NEW $R4 $R8; do
  echo "== $lib" | tee -a $O/ab_configs.log
  LIGHTGLUE_AMD_LIB=$PWD/$lib timeout 600 python tools/bench_configs.py "#3' " "#3b " "#5' " "#4 " 2>&1 | grep "^|" | tee -a $O/ab_configs.log
done
( timeout 300 python tools/tail_timing.py f16x3 5; timeout 300 python tools/tail_timing.py f16x3 6 ) 2>&1 | grep -v amdgpu.ids | tee $O/tail_timing.log
