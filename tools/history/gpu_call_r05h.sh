#!/bin/bash
# Round 5, call h: the new round-5 tests + parallel tests after the fixes; A/B of the ctx-DMA tail against the same tree built with -DLG_TAIL_CTX_DMA=0; adaptive configs (ticket compaction).
O=gpurun_out/r05h; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_round5.py tests/test_parallel_gpu.py -q -x -n 4 --durations=8 > $O/tests_new.log 2>&1; tail -14 $O/tests_new.log
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$1', round(d['value']), round(d['ms_per_step'],3), {x: round(k[x],3) for x in ('attn_self','attn_cross','fused_tail') if x in k}, d['parity']['index_mismatches'], d['parity']['max_dscore'])"; }
lib() { if [ "$1" = tree ]; then echo $PWD/lightglue_amd/liblightglue_amd.so; else echo $PWD/build_variants/liblightglue_amd_$1.so; fi; }
for round in 1 2; do for v in nodma tree; do
  LIGHTGLUE_AMD_LIB=$(lib $v) timeout 90 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-calibration --no-gather-probe 2>/dev/null | tail -1 | line $v
done; done 2>&1 | tee $O/ab_cfg2.log
for v in nodma tree; do echo "== $v"; LIGHTGLUE_AMD_LIB=$(lib $v) timeout 120 python tools/bench_configs.py "#3' " "#3b" "#5' " 2>&1 | grep "^| #"; done | tee $O/ab_configs.log
