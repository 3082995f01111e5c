#!/bin/bash
# ONE gpurun call: the profile set of a round.  Everything lands in gpurun_out/round/ (copy what is to be judged into profiles/).
#   GPU tests + smoke, bench (default, fast opt-in, recipe D), 2-rank flow test, kernel traces (cfg #2, cfg #4 shard, adaptive cfg #3', B = 1),
#   PMC passes (SQ / FETCH_SIZE / WRITE_SIZE in their own runs, kernel trace only) for cfg #2 and for the adaptive case, all BASELINE
#   configs on one GPU, B = 1 latency, SuperPoint extractor
O=gpurun_out/round; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests -m gpu -q > $O/gputests.log 2>&1; grep -E 'passed|failed|error' $O/gputests.log | tail -4   # (the full log is kept: a failure must be readable afterwards)
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err
python bench.py --attention fp16 --no-cpu-baseline > $O/bench_fast_attention.json 2>> $O/bench.err
python bench.py --recipe D > $O/bench_recipe_d.json 2>> $O/bench.err    # the same workload on trained-model statistics (parity: tests/golden/trained_stats_1024_b8)
python - <<'PY'
import json
for f in ("bench.json", "bench_fast_attention.json", "bench_recipe_d.json"):
    d = json.loads(open("gpurun_out/round/" + f).read().strip().splitlines()[-1])
    print(f, round(d["value"]), round(d["ms_per_step"], 3), "sync", d["value_synchronous_forward"], "tail frac", round(d["roofline"]["frac"], 4), "attn frac", round(d["roofline_attention"]["frac"], 4),
          "hbm frac", round(d["roofline_hbm"]["frac"], 3), "cpu", (d.get("cpu_baseline") or {}).get("value"), (d.get("cpu_baseline") or {}).get("reference_estimate_pairs_per_s"), d["kernel_ms_per_step"])
    print("   parity", d["parity"], d.get("parity_oracle"))
PY
for c in 3 4 5; do python bench.py --config $c --steps 5 --warmup 3 --no-cpu-baseline > $O/bench_cfg$c.json 2>> $O/bench.err; python - $c <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/round/bench_cfg{sys.argv[1]}.json"))
print("bench --config", sys.argv[1], round(d["value"], 1), "pairs/s", round(d["ms_per_step"], 2), "ms/step; roofline:", d["roofline"]["kernel"], round(d["roofline"]["frac"], 4), "; gather probe:", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in (d["gather_probe_one_gpu"] or {}).items() if k != "what"})
PY
done
LG_BENCH_BACKEND=gloo LG_BENCH_ONE_GPU=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 2>&1 | tail -1 > $O/bench_2rank_gloo_one_gpu.json; python -c "
import json; d = json.loads(open('$O/bench_2rank_gloo_one_gpu.json').read()); print('2 ranks on one GPU (gloo):', round(d['value']), d['rccl'])"
rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-calibration --no-gather-probe > $O/trace.log 2>&1
python tools/rocpd_stats.py $(find $O/trace -name "*.db" | head -1) $O/kernel_trace.md | head -16
for c in cfg4_b32_n4096 adaptive_b16_n2048 b1_n1024; do
  rocprofv3 --kernel-trace --stats -d $O/trace_$c -o t -- python tools/trace_case.py $c > $O/trace_$c.log 2>&1
  python tools/rocpd_stats.py $(find $O/trace_$c -name "*.db" | head -1) $O/kernel_trace_$c.md | head -12
done
bash tools/pmc_round.sh $O/pmc
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $ctr -d $O/pmc_adaptive_$ctr -o p -- python tools/trace_case.py adaptive_b16_n2048 > $O/pmc_adaptive_$ctr.log 2>&1
  python tools/rocpd_pmc.py $(find $O/pmc_adaptive_$ctr -name "*.db" | head -1) $O/pmc_adaptive_$ctr.md | grep -i "compact\|kernel \|---"
done
find $O -name "*.db" -delete
timeout 400 python tools/bench_configs.py 2>&1 | grep -v amdgpu.ids | tail -13; cp gpurun_out/configs.md $O/configs.md
timeout 200 python tools/latency_b1.py 2>&1 | grep -v amdgpu.ids > $O/latency_b1.log; cat $O/latency_b1.log
timeout 200 python tools/bench_superpoint.py 2>&1 | grep -v amdgpu.ids | tail -8 | tee $O/superpoint.log
( python tools/tail_timing.py f16x3 1; python tools/tail_timing.py f16x3 5; python tools/tail_wall.py ) 2>&1 | grep -v amdgpu.ids | tee $O/tail_timing.log
# round 6: the clock the two big kernels get inside a forward (power management) and the board's power under the bench loop
if [ -f build_variants/liblightglue_amd_attn_wall.so ]; then LIGHTGLUE_AMD_LIB=$PWD/build_variants/liblightglue_amd_attn_wall.so timeout 200 python tools/attn_wall.py 2>&1 | grep -v "^live\|amdgpu.ids" | tee $O/attn_wall_clock.log; fi
( python bench.py --steps 2500 --warmup 5 --no-cpu-baseline --no-calibration --no-gather-probe > $O/bench_long.json 2>/dev/null & BP=$!; sleep 14; for i in 1 2 3 4 5; do rocm-smi --showpower --showclocks 2>&1 | grep -E "Power|sclk"; sleep 1.5; done; wait $BP ) > $O/rocm_smi_under_bench.log 2>&1; grep -E "Power|sclk" $O/rocm_smi_under_bench.log | tail -4
