#!/bin/bash
# Round-4 call d: matchability terms from the tail heads (no rowdot launch), final projection as its own 64-row kernel / fused into the last
# tail at fixed depth, similarity kernel with 16-byte stores — GPU suite, A/B against HEAD~ (build_variants/..._base.so), adaptive configs,
# B = 1 latency, PMC passes of the adaptive case (compaction bytes) and L2 hit counters of the default bench.
O=gpurun_out/r04d; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x > $O/gputests.log 2>&1; grep -E 'passed|failed|error' $O/gputests.log | tail -4
BASE=build_variants/liblightglue_amd_base.so; NEW=lightglue_amd/liblightglue_amd.so
for round in 1 2; do for lib in $BASE $NEW; do
  LIGHTGLUE_AMD_LIB=$PWD/$lib timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', round(d['value']), round(d['ms_per_step'],3), d['kernel_ms_per_step'], d['parity']['index_mismatches'], d['parity']['max_dscore'])"
done; done 2>&1 | tee $O/ab_cfg2.log
for lib in $BASE $NEW; do
  echo "== $lib" | tee -a $O/ab_configs.log
  LIGHTGLUE_AMD_LIB=$PWD/$lib timeout 600 python tools/bench_configs.py "#3' " "#3b " "#5' " 2>&1 | grep "^|" | tee -a $O/ab_configs.log
done
timeout 200 python tools/latency_b1.py 2>&1 | grep -v amdgpu.ids > $O/latency_b1.log; cat $O/latency_b1.log
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $ctr -d $O/pmc_adaptive_$ctr -o p -- python tools/trace_case.py adaptive_b16_n2048 > $O/pmc_adaptive_$ctr.log 2>&1
  python tools/rocpd_pmc.py $(find $O/pmc_adaptive_$ctr -name "*.db" | head -1) $O/pmc_adaptive_$ctr.md | grep -i "compact\|decide\|kernel \|---"
done
timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d $O/pmc_tcc -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-calibration > $O/pmc_tcc.log 2>&1
python tools/rocpd_pmc.py $(find $O/pmc_tcc -name "*.db" | head -1) $O/pmc_tcc.md | head -14
timeout 300 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum -d $O/pmc_tcp -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-calibration > $O/pmc_tcp.log 2>&1
python tools/rocpd_pmc.py $(find $O/pmc_tcp -name "*.db" | head -1) $O/pmc_tcp.md | head -14
find $O -name "*.db" -delete
