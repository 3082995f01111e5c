#!/bin/bash
# Build a variant of the library with extra compiler flags into lightglue_amd/liblightglue_amd_<name>.so
# (travels to the GPU box; select it with LIGHTGLUE_AMD_LIB):   tools/build_variant.sh <name> <flags...>
set -e
NAME=$1; shift
cd "$(dirname "$0")/../lightglue_amd/csrc"
mkdir -p build_$NAME
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form -fno-slp-vectorize"
for f in lg_gemm lg_tail $( [[ "$*" == *LG_EXPERIMENTS* ]] && echo lg_tail4 lg_tail128 ) lg_proj lg_attention lg_pointwise lg_adaptive lg_assign lg_superpoint lg_sp_encoder lg_engine; do
  hipcc $FLAGS "$@" -c $f.hip -o build_$NAME/$f.o &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o ../liblightglue_amd_$NAME.so build_$NAME/*.o
echo built lightglue_amd/liblightglue_amd_$NAME.so
