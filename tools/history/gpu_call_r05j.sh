#!/bin/bash
# Round 5, call j: 64-row compaction chunks (tree) against 128-row ones (c128): adaptive configs, kernel trace of cfg #3' for the compaction's launch time, adaptive tests.
O=gpurun_out/r05j; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
lib() { if [ "$1" = tree ]; then echo $PWD/lightglue_amd/liblightglue_amd.so; else echo $PWD/build_variants/liblightglue_amd_$1.so; fi; }
for round in 1 2; do for v in c128 tree; do echo "== $v"; LIGHTGLUE_AMD_LIB=$(lib $v) timeout 120 python tools/bench_configs.py "#3' " "#3b" "#5' " 2>&1 | grep "^| #"; done; done | tee $O/ab_configs.log
for v in c128 tree; do
  LIGHTGLUE_AMD_LIB=$(lib $v) rocprofv3 --kernel-trace --stats -d $O/trace_$v -o t -- python tools/trace_case.py adaptive_b16_n2048 > $O/trace_$v.log 2>&1
  python tools/rocpd_stats.py $(find $O/trace_$v -name "*.db" | head -1) $O/kernel_trace_adaptive_$v.md | grep -i "compact\|decide\|proj_kernel\|Name" | head -6
done
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $ctr -d $O/pmc_adaptive_$ctr -o p -- python tools/trace_case.py adaptive_b16_n2048 > $O/pmc_adaptive_$ctr.log 2>&1
  python tools/rocpd_pmc.py $(find $O/pmc_adaptive_$ctr -name "*.db" | head -1) $O/pmc_adaptive_$ctr.md | grep -i "compact\|kernel \|---"
done
find $O -name "*.db" -delete
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_properties.py -q -x -n 4 -k "adaptive or pruned or prune or ragged or compact" > $O/tests_adaptive.log 2>&1; tail -3 $O/tests_adaptive.log
