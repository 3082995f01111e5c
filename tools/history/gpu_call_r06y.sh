#!/bin/bash
# round 6 call y: similarity on f16 planes (lg_sim.hip): bit-identity test, then same-box timing of sim_planes = 0 / 1 and of the store placement variant
python -m pytest tests/test_gpu_round6.py -x -q -k "similarity_on_planes" 2>&1 | tail -5
ab() {  # name, env...
  for round in 1 2; do
    for v in "tree0:LG_BENCH_OPTS=sim_planes=0" "tree1:LG_BENCH_OPTS=sim_planes=1" "pend0:LIGHTGLUE_AMD_LIB=$PWD/build_variants/liblightglue_amd_simpend0.so"; do
      name=${v%%:*}; kv=${v#*:}
      env "$kv" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-gather-probe "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; p=d.get('parity') or {}; print('$name', '$*', round(d['value'],1), round(d['ms_per_step'],3), {x: round(k[x],4) for x in ('sim','assign','gemm_final_proj','fused_tail') if x in k}, p.get('index_mismatches'), p.get('max_dscore'))"
    done
  done
}
ab
ab --config 3 --inflight 1
ab --config 4 --steps 6 --warmup 2
