#!/bin/bash
# Round 5, first call: same-box A/B of the GELU placement candidates (tools/prepare_r05a.sh builds them) against the tree's library, cfg #2, two rounds, parity of every run;
# then the tail stamps of the best two and cfg #3' / #5' for the winner.  The tree is untouched; adoption = apply the patch with the winning flags as defaults, full suite, PMC refresh.
O=gpurun_out/r05a; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$1', round(d['value']), round(d['ms_per_step'],3), {x: round(k[x],3) for x in ('attn_self','attn_cross','fused_tail') if x in k}, d['parity']['index_mismatches'], d['parity']['max_dscore'])"; }
lib() { if [ "$1" = base ]; then echo $PWD/lightglue_amd/liblightglue_amd.so; else echo $PWD/build_variants/liblightglue_amd_$1.so; fi; }
for round in 1 2; do for v in base gs gsi3 gi2 go gso; do
  LIGHTGLUE_AMD_LIB=$(lib $v) timeout 90 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-calibration 2>/dev/null | tail -1 | line $v
done; done 2>&1 | tee $O/ab_cfg2.log
BEST=$(python - <<'PY'
import re, collections
acc = collections.defaultdict(list)
for l in open("gpurun_out/r05a/ab_cfg2.log"):
    p = l.split()
    if len(p) > 2 and p[1].isdigit(): acc[p[0]].append(int(p[1]))
print(max((k for k in acc if k != "base"), key=lambda k: sum(acc[k]) / len(acc[k])))
PY
)
echo "best variant: $BEST" | tee -a $O/ab_cfg2.log
for v in base $BEST; do echo "== $v"; LIGHTGLUE_AMD_LIB=$(lib $v) timeout 120 python tools/tail_timing.py f16x3 5 2>&1 | grep -E "phaseA|LN|GELU0|phaseB|epilogue|total"; done | tee $O/stamps.log
LIGHTGLUE_AMD_LIB=$(lib $BEST) timeout 150 python -m pytest tests/test_gpu_parity.py -q -x -n 4 -k "default_precision_parity or pipeline_stages_layer0 or fused_next_projection or tail_row_tile" > $O/tests_best.log 2>&1; tail -2 $O/tests_best.log
for v in base $BEST; do echo "== $v"; LIGHTGLUE_AMD_LIB=$(lib $v) timeout 90 python tools/bench_configs.py "#3' " "#5' " 2>&1 | grep "^| #"; done | tee $O/ab_configs.log
# compaction with two workgroups per CU (tools/experiments/compact_two_per_cu.patch): adaptive configs, two rounds, then the adaptive fixtures with it
for round in 1 2; do for v in base c2; do echo "== $v"; LIGHTGLUE_AMD_LIB=$(lib $v) timeout 90 python tools/bench_configs.py "#3' " "#3b" "#5' " 2>&1 | grep "^| #"; done; done | tee $O/ab_compaction.log
LIGHTGLUE_AMD_LIB=$(lib c2) timeout 150 python -m pytest tests/test_gpu_parity.py -q -x -n 4 -k "adaptive or pruned or ragged" > $O/tests_c2.log 2>&1; tail -2 $O/tests_c2.log
