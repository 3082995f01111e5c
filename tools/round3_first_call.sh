#!/bin/bash
# First GPU call of the next round: everything below was BUILT at the end of round 2 without GPU minutes left.
# In the build container first:   tools/tail_sched_variants.sh build (≈10 min: the sched_group_barrier builds compile slowly) ; tools/build_variant.sh ctx6 -DLG_EXPERIMENTS -DLG_TAIL_CTX_FP6=1 ; tools/build_variant.sh fold -DLG_ATTN_FOLD=1 ; tools/build_variant.sh ablw -DLG_PROJ_ABLATE_W=1
# then:   gpurun --timeout 1800 -- 'bash tools/round3_first_call.sh'        (≈20-25 min of box time; results in gpurun_out/round3/)
mkdir -p gpurun_out/round3; O=gpurun_out/round3
export TMPDIR=/tmp
# 1. precision "f16x3" (split scheme on f16 planes): the gated parity tests + per-golden margins next to the default's
LG_TEST_UNVALIDATED=1 timeout 600 python -m pytest tests/test_gpu_unvalidated.py -m gpu -q > $O/f16x3_tests.log 2>&1; tail -3 $O/f16x3_tests.log
timeout 600 python tools/gpu_lab.py bf16x3 f16x3 > $O/lab_f16x3.log 2>&1; grep -c "golden \[f16x3\]" $O/lab_f16x3.log
python - <<'PY'
import json
d = json.load(open("gpurun_out/lab.json"))["golden"]
for prec in ("bf16x3", "f16x3"):
    r = [v for k, v in d.items() if k.startswith(prec + "/") and isinstance(v, dict)]
    if r:
        print(prec, "cases", len(r), "index mismatches", sum(x["idx_mismatch0"] + x["idx_mismatch1"] for x in r), "worst max_dscore %.2e" % max(x["max_dscore"] for x in r),
              "mean rms_dscore %.2e" % (sum(x["rms_dscore"] for x in r) / len(r)))
PY
# 2. whole-step A/B inside this one box: default, f16x3, the scheduling builds, the ctx-half fp6 build (needs f16x3)
ab() { LIGHTGLUE_AMD_LIB=$PWD/$1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline ${2:+--precision $2} 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['kernel_ms_per_step']; p = d.get('parity') or {}
print('$1 ${2:-bf16x3}', round(d['value']), round(d['ms_per_step'], 3), 'tail', round(k.get('fused_tail', 0), 3), 'parity', p.get('index_mismatches'), p.get('unexplained'), p.get('max_dscore'))"; }
for round in 1 2; do
  ab lightglue_amd/liblightglue_amd.so
  ab lightglue_amd/liblightglue_amd.so f16x3
  for v in o1 o2 gs o1gs o2gs pe pp o1gspp; do [ -f lightglue_amd/liblightglue_amd_sched_$v.so ] && ab lightglue_amd/liblightglue_amd_sched_$v.so; done
  [ -f lightglue_amd/liblightglue_amd_fold.so ] && ab lightglue_amd/liblightglue_amd_fold.so
  [ -f lightglue_amd/liblightglue_amd_ablw.so ] && ab lightglue_amd/liblightglue_amd_ablw.so   # TIMING ABLATION (wrong results, parity column meaningless): projection without weight loads
done 2>&1 | tee $O/ab.log
# 3. the attention fold build (q, k pre-scaled; scores accumulate from -m_run): NOT bit-identical to the default -> the regular parity suite
#    on the reference's fixtures, both precisions (stage-level tests compare q / k with the unscaled oracle tensors and do not apply)
[ -f lightglue_amd/liblightglue_amd_fold.so ] && { LIGHTGLUE_AMD_LIB=$PWD/lightglue_amd/liblightglue_amd_fold.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "golden_exactly or default_precision_parity" > $O/fold_tests.log 2>&1; tail -5 $O/fold_tests.log; }
# 4. LAST (a kernel that has never run: if it faults, everything above is already on disk): the ctx-half fp6 build (needs f16x3) —
#    whole-step A/B against f16x3 of the product library, then the gated parity tests with this library
if [ -f lightglue_amd/liblightglue_amd_ctx6.so ]; then
  for round in 1 2; do ab lightglue_amd/liblightglue_amd.so f16x3; ab lightglue_amd/liblightglue_amd_ctx6.so f16x3; done 2>&1 | tee $O/ab_ctx6.log
  LIGHTGLUE_AMD_LIB=$PWD/lightglue_amd/liblightglue_amd_ctx6.so LG_TEST_UNVALIDATED=1 timeout 600 python -m pytest tests/test_gpu_unvalidated.py -m gpu -q > $O/ctx6_tests.log 2>&1; tail -3 $O/ctx6_tests.log
fi
