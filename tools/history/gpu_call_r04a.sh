#!/bin/bash
# Round-4 call a: the new self-launch test, baseline bench on this box, and the two prepared tile-map patches of round 3 (attention unit rotation, projection
# rotation) A/B'd on cfg #2 AND on the adaptive / ragged configs (the tile-map lesson of round 3).
O=gpurun_out/r04a; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_parallel_gpu.py -m gpu -q -x > $O/gputests.log 2>&1; tail -3 $O/gputests.log
BASE=lightglue_amd/liblightglue_amd.so
tools/ab.sh "$BASE build_variants/liblightglue_amd_attnrot4.so build_variants/liblightglue_amd_attnrot2.so" 2>&1 | tee $O/ab_cfg2.log
for lib in $BASE build_variants/liblightglue_amd_attnrot4.so build_variants/liblightglue_amd_attnrot2.so build_variants/liblightglue_amd_projrot.so; do
  echo "== $lib" | tee -a $O/ab_configs.log
  LIGHTGLUE_AMD_LIB=$PWD/$lib timeout 600 python tools/bench_configs.py "#3' " "#3b " "#5' " 2>&1 | grep "^|" | tee -a $O/ab_configs.log
done
for lib in $BASE build_variants/liblightglue_amd_attnrot4.so; do
  echo "== $lib" | tee -a $O/ab_configs.log
  LIGHTGLUE_AMD_LIB=$PWD/$lib timeout 600 python tools/bench_configs.py "#4 " 2>&1 | grep "^|" | tee -a $O/ab_configs.log
done
