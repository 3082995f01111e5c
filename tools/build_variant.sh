#!/bin/bash
# Build a variant of the library with extra compiler flags into build_variants/liblightglue_amd_<name>.so (git-ignored, travels to
# the GPU box; select it with LIGHTGLUE_AMD_LIB=$PWD/build_variants/liblightglue_amd_<name>.so):   tools/build_variant.sh <name> <flags...>
# Variant libraries never live in lightglue_amd/ (the driver pushed 48 MB of them).
# LG_VARIANT_SRC=<dir> builds from a patched COPY of lightglue_amd/csrc (tools/make_variant_src.sh) instead of the tree.
set -e
NAME=$1; shift
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
mkdir -p "$ROOT/build_variants/obj_$NAME"
cd "${LG_VARIANT_SRC:-$ROOT/lightglue_amd/csrc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form -fno-slp-vectorize"
for f in lg_gemm lg_sim lg_tail lg_proj lg_attention lg_pointwise lg_adaptive lg_assign lg_superpoint lg_sp_encoder lg_engine; do
  hipcc $FLAGS "$@" -c $f.hip -o "$ROOT/build_variants/obj_$NAME/$f.o" &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/build_variants/liblightglue_amd_$NAME.so" "$ROOT"/build_variants/obj_$NAME/*.o
echo built build_variants/liblightglue_amd_$NAME.so
