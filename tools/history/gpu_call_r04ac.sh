#!/bin/bash
# Round-4 call ac: kernel trace and SQ counter pass of cfg #2 on the final sources, on whatever box the pool hands out (the two classes differ by ~6 % on the tail)
O=gpurun_out/r04ac; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 60 rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/trace.log 2>&1
python tools/rocpd_stats.py $(find $O/trace -name "*.db" | head -1) $O/kernel_trace.md | head -8
tail -1 $O/trace.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('under trace:', round(d['value']), 'sustained', d['sustained_dense_bf16_tflops'], d['effective_mfma_clock_mhz'])" 2>/dev/null
timeout 60 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES -d $O/sq -o s -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-calibration > $O/sq.log 2>&1
python tools/rocpd_pmc.py $(find $O/sq -name "*.db" | head -1) $O/pmc_sq.md | head -7
find $O -name "*.db" -delete
