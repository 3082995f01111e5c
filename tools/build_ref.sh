#!/bin/bash
# Build the kernels of a git revision into build_variants/liblightglue_amd_<name>.so (the A/B partner of the working tree):  tools/build_ref.sh <rev> <name>
set -e
REV=$1; NAME=$2
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
T=$(mktemp -d); git -C "$ROOT" archive "$REV" lightglue_amd/csrc include | tar -x -C "$T"
mkdir -p "$ROOT/build_variants/obj_$NAME"; cd "$T/lightglue_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form -fno-slp-vectorize"
for f in lg_*.hip; do hipcc $FLAGS -c $f -o "$ROOT/build_variants/obj_$NAME/${f%.hip}.o" 2>/dev/null & done; wait
hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/build_variants/liblightglue_amd_$NAME.so" "$ROOT"/build_variants/obj_$NAME/*.o
rm -rf "$T"; echo built build_variants/liblightglue_amd_$NAME.so from $REV
