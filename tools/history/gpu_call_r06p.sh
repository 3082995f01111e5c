export TMPDIR=/tmp
python -m pytest tests -m gpu -q -n 4 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python bench.py 2>/dev/null | tail -1 > gpurun_out/final_bench.json; python -c "
import json; d=json.load(open('gpurun_out/final_bench.json')); c=d['cpu_baseline']
print(round(d['value'],1), 'tail frac', round(d['roofline']['frac'],4), 'traffic', d['roofline']['traffic'], 'power', d['power'])
print(' cpu', round(c['value'],2), c['rounds_pairs_per_s'], c['round_median_forward_ms'], 'pairs checked', d['parity_oracle']['pairs'], d['parity_oracle']['index_mismatches'])"
