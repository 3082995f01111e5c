#!/bin/bash
# Round 6, call e: the whole GPU suite on the current tree + the default bench (new CPU leg on the box's host, clock probe) twice
O=gpurun_out/r06e; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q -n 4 > $O/gputests.log 2>&1; grep -E 'passed|failed|error' $O/gputests.log | tail -4
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
for i in 1 2; do python bench.py > $O/bench_$i.json 2> $O/bench_$i.err; tail -1 $O/bench_$i.json | python -c "
import sys, json
d = json.loads(sys.stdin.read())
c = d['cpu_baseline']
print(round(d['value'],1), round(d['ms_per_step'],3), 'sync', round(d['value_synchronous_forward'] or 0,1), 'tail frac', round(d['roofline']['frac'],4), 'traffic', d['roofline']['traffic'], 'clock in tail', d['shader_clock_mhz_inside_tail_kernel'], 'sustained', d['sustained_dense_bf16_tflops'], d['effective_mfma_clock_mhz'])
print('  cpu', c.get('kind'), round(c.get('value',0),3), c.get('rounds_pairs_per_s'), 'spread', c.get('round_spread'), 'cores', c.get('cores'), c.get('threads_pinned'), c.get('one_thread_pairs_per_s'), c.get('cfg1_n512_b1'))
print('  cpu sample:', c.get('sample'))
print('  parity', d['parity'], d.get('parity_oracle'))
print('  gather probe', {k: v for k, v in (d['gather_probe_one_gpu'] or {}).items() if k != 'what'})
print('  step_output', d['step_output'])
"; done
