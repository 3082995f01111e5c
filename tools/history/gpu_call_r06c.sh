#!/bin/bash
# Round 6, call c: envelope / default range guard / N = 8192 tests, the new wire row (full output dict at every N, status raised on every rank), 8-rank rehearsal.
O=gpurun_out/r06c; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_round6.py tests/test_parallel_gpu.py "tests/test_gpu_round5.py::test_extension_outputs_equal_the_int32_ones" tests/test_gpu_round5.py::test_range_guard_flags_the_overflowing_pair_only tests/test_gpu_round5.py::test_nan_input_is_flagged -x -q --durations=8 > $O/tests.log 2>&1; tail -30 $O/tests.log
