#!/bin/bash
# Round-4 call ab (the round's last GPU seconds): attention with scalar-base LDS-DMA addresses + stepping LDS fragment address registers
# (tools/experiments/attn_scalar_dma_lds_step.patch) as a variant library against the tree's; if it wins the same-box A/B with clean parity, the SAME call
# swaps the patched sources + library in ON THE BOX and produces what adoption needs (PMC passes for the new digest, bench line, full GPU suite).
O=gpurun_out/r04ab; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
VAR=build_variants/liblightglue_amd_attnsb.so
for round in 1 2; do for lib in lightglue_amd/liblightglue_amd.so $VAR; do
  LIGHTGLUE_AMD_LIB=$PWD/$lib timeout 90 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-calibration 2>/dev/null | tail -1 > $O/line.json
  python - "$lib" <<'PY' | tee -a $O/ab_cfg2.log
import sys, json
d = json.loads(open("gpurun_out/r04ab/line.json").read()); k = d["kernel_ms_per_step"]
print(json.dumps({"lib": sys.argv[1], "value": round(d["value"], 1), "ms": round(d["ms_per_step"], 3), "attn_self": k.get("attn_self"), "attn_cross": k.get("attn_cross"), "tail": k.get("fused_tail"),
                  "mismatches": d["parity"]["index_mismatches"], "dscore": d["parity"]["max_dscore"]}))
PY
done; done
python - <<'PY' > $O/decision.txt
import json
rows = [json.loads(l) for l in open("gpurun_out/r04ab/ab_cfg2.log")]
base = [r for r in rows if "build_variants" not in r["lib"]]; var = [r for r in rows if "build_variants" in r["lib"]]
mb, mv = sum(r["value"] for r in base) / len(base), sum(r["value"] for r in var) / len(var)
ok = len(var) == 2 and all(r["mismatches"] == 0 and r["dscore"] < 2e-4 for r in var) and mv > 1.005 * mb and min(r["value"] for r in var) > max(r["value"] for r in base)
print("ADOPT" if ok else "KEEP", round(mb, 1), round(mv, 1), round(mv / mb, 4))
PY
cat $O/decision.txt
if grep -q ADOPT $O/decision.txt; then
  cp build_variants/src_attnsb/lg_* lightglue_amd/csrc/; cp $VAR lightglue_amd/liblightglue_amd.so
  python -c "import bench; print(bench.kernel_source_digest())" 2>/dev/null | tail -1 > $O/digest.txt; cat $O/digest.txt
  bash tools/pmc_round.sh $O/pmc > $O/pmc.log 2>&1; tail -3 $O/pmc.log
  python bench.py > $O/bench.json 2> $O/bench.err
  python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04ab/bench.json").read().strip().splitlines()[-1])
print(round(d["value"]), round(d["ms_per_step"], 3), "tail frac", round(d["roofline"]["frac"], 4), "traffic", d["roofline"]["traffic"], "parity", d["parity"]["index_mismatches"], d.get("parity_oracle", {}).get("index_mismatches"), d["kernel_ms_per_step"])
PY
  timeout 200 python -m pytest tests -m gpu -q -n 4 > $O/gputests.log 2>&1; tail -3 $O/gputests.log
fi
