set -x
python -m pytest tests -m gpu -q -x 2>&1 | tail -3
mkdir -p gpurun_out/r01e
python bench.py > gpurun_out/r01e/bench.json 2> gpurun_out/r01e/bench.err; tail -c 600 gpurun_out/r01e/bench.json
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/r01e/trace -o t -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r01e/trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/r01e/fetch -o f -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r01e/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/r01e/write -o w -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r01e/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES -d gpurun_out/r01e/sq -o s -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r01e/sq.log 2>&1
find gpurun_out/r01e -name "*.db" | while read f; do echo $f; done
python tools/rocpd_stats.py $(find gpurun_out/r01e/trace -name "*.db" | head -1) gpurun_out/r01e/kernel_trace.md | head -12
python tools/rocpd_pmc.py $(find gpurun_out/r01e/fetch -name "*.db" | head -1) gpurun_out/r01e/pmc_fetch.md | head -6
python tools/rocpd_pmc.py $(find gpurun_out/r01e/write -name "*.db" | head -1) gpurun_out/r01e/pmc_write.md | head -6
python tools/rocpd_pmc.py $(find gpurun_out/r01e/sq -name "*.db" | head -1) gpurun_out/r01e/pmc_sq.md | head -8
find gpurun_out/r01e -name "*.db" -delete
