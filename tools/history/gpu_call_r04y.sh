#!/bin/bash
# Round-4 call y (the round's last GPU minutes): instruction-count patches of tools/experiments/ measured as VARIANT libraries built from patched
# copies of csrc (tools/make_variant_src.sh) — the tree and its PMC file stay as they are; adoption (full suite + PMC refresh) is next round's first call.
#   gelu = GELU by exp2(-a Q(a)) (15 instead of 23 instructions per pair, one transcendental per value instead of two)
#   ln   = LayerNorm statistics + affine on packed f32 instructions
#   pp   = fused projection: LDS fragment addresses in four stepping registers (4 instead of 23 VALU per two chunks, none in front of the reads)
#   all  = the three together
O=gpurun_out/r04y; mkdir -p $O
export TMPDIR=/tmp
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$1', round(d['value']), round(d['ms_per_step'],3), {x: round(k[x],3) for x in ('attn_self','attn_cross','fused_tail') if x in k}, d['parity']['index_mismatches'], d['parity']['max_dscore'])"; }
ab() { for v in base gelu ln pp all; do LIGHTGLUE_AMD_LIB=$PWD/build_variants/liblightglue_amd_$v.so timeout 90 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-calibration 2>/dev/null | tail -1 | line $v; done; }
ab 2>&1 | tee $O/ab_cfg2.log
LIGHTGLUE_AMD_LIB=$PWD/build_variants/liblightglue_amd_all.so timeout 90 python bench.py --recipe D --steps 5 --warmup 2 --no-cpu-baseline --no-calibration 2>/dev/null | tail -1 | line all_recipeD 2>&1 | tee -a $O/ab_cfg2.log
LIGHTGLUE_AMD_LIB=$PWD/build_variants/liblightglue_amd_all.so timeout 110 python -m pytest tests/test_gpu_parity.py -q -x -k "default_precision_parity or pipeline_stages_layer0 or fused_next_projection or tail_row_tile" > $O/tests_all.log 2>&1; tail -3 $O/tests_all.log
ab 2>&1 | tee -a $O/ab_cfg2.log
for v in base all; do echo "== $v"; LIGHTGLUE_AMD_LIB=$PWD/build_variants/liblightglue_amd_$v.so timeout 60 python tools/bench_configs.py "#3' " "#5' " 2>&1 | grep "^| #"; done | tee $O/ab_configs.log
