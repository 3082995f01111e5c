#!/bin/bash
# Round-4 call b: GPU suite with the recipe-E fixtures, per-fixture error table (-> profiles/r04_fixture_errors.md), phase stamps of the tail WITH its
# fused next projection (both forms) and of the standalone projection, default bench on this box.
O=gpurun_out/r04b; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q > $O/gputests.log 2>&1; grep -E 'passed|failed|error' $O/gputests.log | tail -4
timeout 900 python tools/gpu_lab.py f16x3 f16x3/fp16 fp32 > $O/lab.log 2>&1; cp gpurun_out/lab.json $O/lab.json; grep -c "^golden" $O/lab.log
( timeout 300 python tools/tail_timing.py f16x3 5; timeout 300 python tools/tail_timing.py f16x3 6; timeout 300 python tools/tail_timing.py f16x3 2; timeout 300 python tools/tail_timing.py f16x3 1 ) 2>&1 | grep -v amdgpu.ids | tee $O/tail_timing.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err; tail -c 1500 $O/bench.json
