#!/bin/bash
# deferred output assembly: the new test, then the default bench with and without it (same box)
python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "deferred or match_batch or ragged" 2>&1 | grep -E "passed|failed" | tail -2
for round in 1 2; do
  for flag in "" "--no-pipeline"; do
    timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-calibration $flag 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pipeline' if '$flag' == '' else 'synchronous', round(d['value']), round(d['ms_per_step'],3), d['gpu_ms_per_step_sum'], d['parity'] and (d['parity']['index_mismatches'], d['parity']['max_dscore']))"
  done
done
