#!/bin/bash
# Round-4 call z: adoption of the exp2-form GELU (call y: +2.6 % cfg #2, tail -4.8 %) — PMC passes for the new source digest, the bench line, the full GPU suite
O=gpurun_out/r04z2; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
bash tools/pmc_round.sh $O/pmc > $O/pmc.log 2>&1; tail -6 $O/pmc.log
python bench.py > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04z2/bench.json").read().strip().splitlines()[-1])
print(round(d["value"]), round(d["ms_per_step"], 3), "tail frac", round(d["roofline"]["frac"], 4), "avg launch ms", round(d["roofline"]["avg_launch_ms"], 4), "traffic", d["roofline"]["traffic"], d["roofline"]["traffic_source"])
print("parity", d["parity"], d.get("parity_oracle")); print(d["kernel_ms_per_step"])
PY
timeout 330 python -m pytest tests -m gpu -q -n 4 > $O/gputests.log 2>&1; tail -4 $O/gputests.log
