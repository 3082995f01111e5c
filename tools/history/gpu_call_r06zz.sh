#!/bin/bash
# Round 6, final: PMC traffic passes + default bench on the FINAL kernel-source digest (after the explicit vmcnt(0) in adapt_decide_kernel), whole GPU suite serially as the driver runs it
O=gpurun_out/r06zz; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
bash tools/pmc_round.sh $O/pmc > $O/pmc_round.log 2>&1; tail -12 $O/pmc_round.log
cp profiles/pmc_traffic.json $O/pmc_traffic.json
python bench.py > $O/bench_default.json 2> $O/bench.err; tail -1 $O/bench_default.json | python -c "
import sys, json
d = json.loads(sys.stdin.read()); c = d['cpu_baseline']
print(round(d['value'],1), round(d['ms_per_step'],3), 'tail frac', round(d['roofline']['frac'],4), 'traffic', d['roofline']['traffic'], 'clock', d['shader_clock_mhz_inside_tail_kernel'], 'sustained', d['sustained_dense_bf16_tflops'])
print(' cpu', round(c['value'],3), c.get('rounds_pairs_per_s'), c.get('round_median_forward_ms'), c.get('forward_ms_p10_p50_p90'))
print(' parity', d['parity']['index_mismatches'], d['parity_oracle']['index_mismatches'], d['parity_oracle']['pairs'])"
python -m pytest tests -m gpu -q > $O/gputests.log 2>&1; grep -E 'passed|failed|error' $O/gputests.log | tail -3
