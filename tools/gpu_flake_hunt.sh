#!/bin/bash
# repeat the parity file in fresh processes until something fails (full log of the failing run is kept)
for i in 1 2 3 4 5 6; do
  timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -m gpu > gpurun_out/flake_$i.log 2>&1
  grep -E "passed|failed" gpurun_out/flake_$i.log | tail -1
  if grep -q "failed" gpurun_out/flake_$i.log; then break; fi
done
