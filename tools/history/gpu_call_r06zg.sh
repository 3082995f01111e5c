O=gpurun_out/final; rm -rf $O; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; tail -2 $O/gputests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
( time python bench.py > $O/bench.json 2> $O/bench.err ) 2>&1 | grep real
python bench.py --config 3 --no-cpu-baseline > $O/bench_cfg3.json 2>> $O/bench.err
python bench.py --config 5 --no-cpu-baseline > $O/bench_cfg5.json 2>> $O/bench.err
python - <<'PY'
import json
for f in ("bench.json","bench_cfg3.json","bench_cfg5.json"):
    d = json.loads(open("gpurun_out/final/"+f).read().strip().splitlines()[-1])
    print(f, round(d["value"],1), d["unit"], round(d["ms_per_step"],3), "roofline", d["roofline"].get("kernel"), round(d["roofline"]["frac"],4), "traffic", d["roofline"].get("traffic"), "cpu", (d.get("cpu_baseline") or {}).get("value"), (d.get("cpu_baseline") or {}).get("kind"), "parity", (d.get("parity") or {}).get("index_mismatches"), (d.get("parity_oracle") or {}).get("index_mismatches"), "W", (d.get("power") or {}).get("watts_median"))
PY
