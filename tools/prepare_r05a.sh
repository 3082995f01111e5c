#!/bin/bash
# Builds the variant libraries of round 5's first A/B call from the prepared patch (tools/experiments/gelu_scalar_interleave.patch); run in the build container, then
#   gpurun --timeout 400 -- 'bash tools/gpu_call_r05a.sh'
set -e
cd "$(dirname "$0")/.."
tools/make_variant_src.sh r05 tools/experiments/gelu_scalar_interleave.patch | tail -1
export LG_VARIANT_SRC=$PWD/build_variants/src_r05
tools/build_variant.sh gs   -DLG_GELU_SCALAR=1 | tail -1
tools/build_variant.sh gsi3 -DLG_GELU_SCALAR=1 -DLG_GELU_INTERLEAVE=3 | tail -1
tools/build_variant.sh gi2  -DLG_GELU_INTERLEAVE=2 | tail -1
tools/build_variant.sh go   -DLG_GELU_OFFSET=1 | tail -1
tools/build_variant.sh gso  -DLG_GELU_SCALAR=1 -DLG_GELU_OFFSET=1 | tail -1
unset LG_VARIANT_SRC
tools/make_variant_src.sh r05c tools/experiments/compact_two_per_cu.patch | tail -1
LG_VARIANT_SRC=$PWD/build_variants/src_r05c tools/build_variant.sh c2 | tail -1
rm -rf build_variants/obj_*
