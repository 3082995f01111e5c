#!/bin/bash
# Round-3 call i: new soak test, then A/B of two experiment builds (tail tile interleave, non-temporal q/k/v stores) against the product.
mkdir -p gpurun_out/r03i
O=gpurun_out/r03i
( timeout 300 python -m pytest tests/test_gpu_properties.py -x -q -k "repeated_forwards" > $O/soak.log 2>&1; echo "soak rc $?" )
tail -3 $O/soak.log
( tools/ab.sh "lightglue_amd/liblightglue_amd.so build_variants/liblightglue_amd_ilv.so build_variants/liblightglue_amd_ntq.so build_variants/liblightglue_amd_ilvntq.so" > $O/ab.log 2>&1 )
cat $O/ab.log
