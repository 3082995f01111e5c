#!/bin/bash
# PMC passes of the default bench (separate passes per counter group, kernel trace only):  tools/pmc_round.sh <outdir>
OUT=${1:-gpurun_out/pmc}
export TMPDIR=/tmp
mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES -d $OUT/sq -o s -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-calibration --no-gather-probe > $OUT/sq.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o f -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-calibration --no-gather-probe > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o w -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-calibration --no-gather-probe > $OUT/write.log 2>&1
python tools/rocpd_pmc.py $(find $OUT/sq -name "*.db" | head -1) $OUT/pmc_sq.md | head -12
python tools/rocpd_pmc.py $(find $OUT/fetch -name "*.db" | head -1) $OUT/pmc_fetch.md | grep -i "sweep\|tail_kernel\|attn\|sim_kernel"
python tools/rocpd_pmc.py $(find $OUT/write -name "*.db" | head -1) $OUT/pmc_write.md | grep -i "sweep\|tail_kernel\|attn\|sim_kernel"
python tools/pmc_traffic.py $(find $OUT/fetch -name "*.db" | head -1) $(find $OUT/write -name "*.db" | head -1) f16x3/f16x3/B32/N1024 | tail -20; cp profiles/pmc_traffic.json $OUT/
find $OUT -name "*.db" -delete
