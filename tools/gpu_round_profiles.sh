#!/bin/bash
# one gpurun call: the profile set of a round (kernel trace, PMC passes, all BASELINE configs, B = 1 latency, extractor)
mkdir -p gpurun_out/round
python -m pytest tests -m gpu -q -x > gpurun_out/round/gputests.log 2>&1; grep -E 'passed|failed|error' gpurun_out/round/gputests.log | tail -4   # (the full log is kept: a failure must be readable afterwards)
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/round/trace -o t -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-calibration > gpurun_out/round/trace.log 2>&1
python tools/rocpd_stats.py $(find gpurun_out/round/trace -name "*.db" | head -1) gpurun_out/round/kernel_trace.md | head -24
find gpurun_out/round -name "*.db" -delete
bash tools/pmc_round.sh gpurun_out/round/pmc
timeout 300 python tools/bench_configs.py 2>&1 | grep -v amdgpu.ids | tail -12
timeout 200 python tools/latency_b1.py 2>&1 | grep -v amdgpu.ids > gpurun_out/round/latency_b1.log; cat gpurun_out/round/latency_b1.log
timeout 200 python tools/bench_superpoint.py 2>&1 | grep -v amdgpu.ids | tail -8
