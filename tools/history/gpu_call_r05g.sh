#!/bin/bash
# Round 5, call g: validation of the ABI extension (reference dtypes / wire row / status written by the engine), range guard, ticket-dealt compaction,
# bench.py --config; the whole GPU suite; same-box A/B against the round-4 kernels.
O=gpurun_out/r05g; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -n 4 > $O/tests_gpu.log 2>&1; tail -5 $O/tests_gpu.log
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$1', round(d['value']), round(d['ms_per_step'],3), {x: round(k[x],3) for x in ('attn_self','attn_cross','fused_tail') if x in k}, d['parity']['index_mismatches'], d['parity']['max_dscore'])"; }
lib() { if [ "$1" = tree ]; then echo $PWD/lightglue_amd/liblightglue_amd.so; else echo $PWD/build_variants/liblightglue_amd_$1.so; fi; }
for round in 1 2; do for v in base tree; do
  LIGHTGLUE_AMD_LIB=$(lib $v) timeout 90 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-calibration --no-gather-probe 2>/dev/null | tail -1 | line $v
done; done 2>&1 | tee $O/ab_cfg2.log
timeout 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err; python - <<'PY'
import json
d = json.load(open("gpurun_out/r05g/bench_default.json"))
print("default:", round(d["value"]), d["roofline"]["frac"], d["cpu_baseline"]["value"], d["cpu_baseline"]["rounds_pairs_per_s"], d["cpu_baseline"]["one_thread_pairs_per_s"], d["cpu_baseline"]["cores"], d["parity"], d["parity_oracle"], d["gather_probe_one_gpu"])
PY
for c in 3 4 5; do timeout 300 python bench.py --config $c --steps 5 --warmup 3 --no-cpu-baseline > $O/bench_cfg$c.json 2> $O/bench_cfg$c.err; python - $c <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/r05g/bench_cfg{sys.argv[1]}.json"))
print("cfg", sys.argv[1], round(d["value"], 1), d["ms_per_step"], d["roofline"]["kernel"], round(d["roofline"]["frac"], 4), d["config"]["baseline_config"][:60], d["gather_probe_one_gpu"])
PY
done 2>&1 | tee $O/configs.log
for v in base tree; do echo "== $v"; LIGHTGLUE_AMD_LIB=$(lib $v) timeout 120 python tools/bench_configs.py "#3' " "#3b" "#5' " 2>&1 | grep "^| #"; done | tee $O/ab_configs.log
