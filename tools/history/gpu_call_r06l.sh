#!/bin/bash
# Round 6, call l: SQ counters of the attention with and without its barrier (periodic inputs): same matrix work? where do the cycles go?
O=gpurun_out/r06l; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
for v in tree nobarrier; do
  if [ $v = tree ]; then L=$PWD/lightglue_amd/liblightglue_amd.so; else L=$PWD/build_variants/liblightglue_amd_$v.so; fi
  LG_BENCH_PERIODIC=64 LG_BENCH_ABLATION=1 LIGHTGLUE_AMD_LIB=$L rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS -d $O/sq_$v -o s -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-calibration --no-gather-probe > $O/sq_$v.log 2>&1
  python tools/rocpd_pmc.py $(find $O/sq_$v -name "*.db" | head -1) $O/pmc_sq_$v.md | grep -i "attn\|kernel \|---\|tail_kernelILi4ELi1"
done
find $O -name "*.db" -delete
