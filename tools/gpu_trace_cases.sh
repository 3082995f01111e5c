#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/round
for c in adaptive_b16_n2048 b1_n1024; do
  rocprofv3 --kernel-trace --stats -d gpurun_out/round/trace_$c -o t -- python tools/trace_case.py $c > gpurun_out/round/trace_$c.log 2>&1
  python tools/rocpd_stats.py $(find gpurun_out/round/trace_$c -name "*.db" | head -1) gpurun_out/round/kernel_trace_$c.md | head -16
  find gpurun_out/round/trace_$c -name "*.db" -delete
done
