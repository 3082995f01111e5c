#!/bin/bash
# Round-4 call w: the per-keypoint preparation fused into the first projection launch (proj_first_kernel) — GPU suite, A/B against HEAD, then the PMC passes
# and bench lines for the new source digest
O=gpurun_out/r04w; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
BASE=build_variants/liblightglue_amd_base.so; NEW=lightglue_amd/liblightglue_amd.so
timeout 1500 python -m pytest tests -m gpu -q > $O/gputests.log 2>&1; grep -E 'passed|failed|error' $O/gputests.log | tail -4; grep -E "^FAILED|^E  " $O/gputests.log | head -8
for round in 1 2 3; do for lib in $BASE $NEW; do
  LIGHTGLUE_AMD_LIB=$PWD/$lib timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$lib', round(d['value']), round(d['ms_per_step'],3), {x: round(k[x],3) for x in ('prep','gemm_qkv_self','attn_self','attn_cross','fused_tail','sim','assign') if x in k}, d['parity']['index_mismatches'], d['parity']['max_dscore'])"
done; done 2>&1 | tee $O/ab_cfg2.log
timeout 200 python tools/latency_b1.py 2>&1 | grep "tiles=0" | tee $O/latency_b1.log
bash tools/pmc_round.sh $O/pmc > $O/pmc.log 2>&1; tail -3 $O/pmc.log | cut -c1-200
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04w/bench.json").read().strip().splitlines()[-1])
print("bench", round(d["value"]), round(d["ms_per_step"], 3), "tail frac", round(d["roofline"]["frac"], 4), d["roofline"]["avg_launch_ms"], "traffic", d["roofline"]["traffic"], d["kernel_ms_per_step"], d["parity"]["index_mismatches"], d["parity_oracle"]["index_mismatches"], d["parity_oracle"]["unexplained"])
PY
