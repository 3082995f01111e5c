#!/bin/bash
# does the box hold its clock under the bench workload?  samples rocm-smi (sclk, power) while bench.py runs a long timed region
rocm-smi --showclocks --showpower --showperflevel 2>&1 | grep -v "^$" | head -30
( for i in $(seq 1 60); do rocm-smi --showclocks --showpower --json 2>/dev/null | python -c "
import sys,json
try:
    d=json.load(sys.stdin); c=d[list(d)[0]]
    print({k:v for k,v in c.items() if 'sclk' in k.lower() or 'power' in k.lower() or 'mclk' in k.lower()})
except Exception as e: print('err', e)
"; sleep 0.2; done ) > gpurun_out/clock_samples.txt 2>&1 &
SAMPLER=$!
sleep 1
python bench.py --steps 1500 --warmup 20 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), d['ms_per_step'], d.get('effective_mfma_clock_mhz'))"
wait $SAMPLER
cat gpurun_out/clock_samples.txt | head -70
