#!/bin/bash
# Round-4 call h: GPU suite + default bench + configs on the tree after calls e - g (phase-B ring pinned, LDS fragments one chunk ahead in phase B and in the
# fused projection, rotary rows once, attention wait as a builtin, final projection fused, matchability terms from the tail heads)
O=gpurun_out/r04h; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > $O/gputests.log 2>&1; grep -E 'passed|failed|error' $O/gputests.log | tail -4
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err; python -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(round(d['value']), d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['roofline_attention']['frac'], d['roofline_hbm']['frac'], d['kernel_ms_per_step'], d['parity'], d['parity_oracle'], d['cpu_baseline']['value'])"
timeout 600 python tools/bench_configs.py 2>&1 | grep "^|" | tee $O/configs.log
