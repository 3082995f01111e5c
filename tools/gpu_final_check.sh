#!/bin/bash
# end-of-round check in one gpurun call: GPU suite (full log kept), smoke, default bench, 2 ranks on one GPU
mkdir -p gpurun_out/final
python -m pytest tests -m gpu -q -x > gpurun_out/final/gputests.log 2>&1; grep -E 'passed|failed|error' gpurun_out/final/gputests.log | tail -3
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err; tail -2 gpurun_out/final/bench.err
python -c "
import json; d=json.loads(open('gpurun_out/final/bench.json').read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],3), d['roofline']['frac'], d['roofline']['frac_of_sustained'], d['roofline_hbm']['frac'], d['effective_mfma_clock_mhz'], d['sustained_dense_bf16_tflops']); print('parity', d['parity']['index_mismatches'], d['parity']['unexplained'], d['parity']['max_dscore'])"
LG_BENCH_BACKEND=gloo LG_BENCH_ONE_GPU=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 2>&1 | tail -1 | cut -c1-200
