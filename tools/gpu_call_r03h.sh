#!/bin/bash
# Round-3 re-entry call: default bench with the torch-backend cpu_baseline, sub-batch streams A/B, three timing ablations, GPU suite.
mkdir -p gpurun_out/r03h
O=gpurun_out/r03h
( timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?" ) 
tail -c 1500 $O/bench_default.json
( timeout 300 python tools/ab_streams.py --streams 1,2,4 --rounds 2 > $O/ab_streams.log 2>&1; echo "streams rc $?" )
( timeout 200 python tools/ab_streams.py --pairs 64 --streams 1,2 --rounds 1 >> $O/ab_streams.log 2>&1; echo "streams64 rc $?" )
cat $O/ab_streams.log
( tools/ab.sh "lightglue_amd/liblightglue_amd.so build_variants/liblightglue_amd_abl_attn_lds.so build_variants/liblightglue_amd_abl_gelu.so build_variants/liblightglue_amd_abl_wlo.so" > $O/ablations.log 2>&1 )
cat $O/ablations.log
( timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" )
tail -5 $O/pytest_gpu.log
