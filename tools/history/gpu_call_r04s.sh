#!/bin/bash
# Round-4 call s: instruction-mix counters of the final kernels (separate PMC passes, kernel trace only): VALU / MFMA / SALU / LDS / VMEM instruction counts,
# LDS bank conflicts, wait cycles
O=gpurun_out/r04s; mkdir -p $O
export TMPDIR=/tmp
run() { timeout 300 rocprofv3 --kernel-trace --pmc $2 -d $O/$1 -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-calibration > $O/$1.log 2>&1; python tools/rocpd_pmc.py $(find $O/$1 -name "*.db" | head -1) $O/pmc_$1.md | head -7; }
run insts "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS"
run mem "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
run waits "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC"
run busy "SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES"
find $O -name "*.db" -delete
