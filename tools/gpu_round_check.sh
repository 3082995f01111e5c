set -x
mkdir -p gpurun_out/round
python -m pytest tests -m gpu -q -x > gpurun_out/round/gputests.log 2>&1; grep -E 'passed|failed|error' gpurun_out/round/gputests.log | tail -4   # (the full log is kept: a failure must be readable afterwards)
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
mkdir -p gpurun_out/round
python bench.py > gpurun_out/round/bench.json 2> gpurun_out/round/bench.err; tail -3 gpurun_out/round/bench.err; python -c "
import json; d=json.loads(open('gpurun_out/round/bench.json').read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],3), d['roofline']['frac'], d['roofline_hbm']['frac'], d['cpu_baseline']['value'], d['kernel_ms_per_step']); print('parity', d['parity']); print('parity_oracle', d['parity_oracle'])"
LG_BENCH_BACKEND=gloo LG_BENCH_ONE_GPU=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 2>&1 | tail -1 | cut -c1-400
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/round/trace -o t -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-calibration > gpurun_out/round/trace.log 2>&1
python tools/rocpd_stats.py $(find gpurun_out/round/trace -name "*.db" | head -1) gpurun_out/round/kernel_trace.md | head -14
find gpurun_out/round -name "*.db" -delete
bash tools/pmc_round.sh gpurun_out/round/pmc
