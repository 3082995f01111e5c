#!/bin/bash
# Scheduling experiments on the fused tail (lg_tail.hip: LG_TAIL_STEP_ORDER / LG_TAIL_GELU_SCALAR; lg_proj_body.h: LG_PROJ_EPI_PREFETCH / LG_PROJ_PINGPONG, fused projection only; the two are alternatives, not to be combined).  Every variant computes the
# same values in the same per-element order as the product build, so outputs must be bit-identical to it.
#   tools/tail_sched_variants.sh build          (here, in the build container: the .so files travel with gpurun)
#   tools/tail_sched_variants.sh ab             (on the GPU box, ONE gpurun call: whole-step A/B, two rounds — DESIGN.md §5: only
#                                                whole-step throughput inside one box is a valid A/B metric)
#   tools/tail_sched_variants.sh parity         (on the GPU box: the golden parity tests against each variant)
set -e
cd "$(dirname "$0")/.."
VARIANTS="o1:-DLG_TAIL_STEP_ORDER=1 o2:-DLG_TAIL_STEP_ORDER=2 gs:-DLG_TAIL_GELU_SCALAR=1 o1gs:-DLG_TAIL_STEP_ORDER=1,-DLG_TAIL_GELU_SCALAR=1 o2gs:-DLG_TAIL_STEP_ORDER=2,-DLG_TAIL_GELU_SCALAR=1 pe:-DLG_PROJ_EPI_PREFETCH=1 pp:-DLG_PROJ_PINGPONG=1 o1gspp:-DLG_TAIL_STEP_ORDER=1,-DLG_TAIL_GELU_SCALAR=1,-DLG_PROJ_PINGPONG=1"
case "$1" in
  build)
    # only lg_tail.o differs: compile that file per variant and link it with the product build's other objects
    make -C lightglue_amd/csrc -j8 >/dev/null
    FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form -fno-slp-vectorize"
    for v in $VARIANTS; do
      name=${v%%:*}; flags=${v#*:}; d=lightglue_amd/csrc/build_sched_$name; mkdir -p $d
      ( hipcc $FLAGS ${flags//,/ } -c lightglue_amd/csrc/lg_tail.hip -o $d/lg_tail.o &&
        hipcc --offload-arch=gfx950 -shared -fPIC -o lightglue_amd/liblightglue_amd_sched_$name.so $d/lg_tail.o $(ls lightglue_amd/csrc/build/*.o | grep -v lg_tail.o) &&
        echo built lightglue_amd/liblightglue_amd_sched_$name.so ) &
    done
    wait ;;
  ab)
    LIBS="lightglue_amd/liblightglue_amd.so"
    for v in $VARIANTS; do LIBS="$LIBS lightglue_amd/liblightglue_amd_sched_${v%%:*}.so"; done
    tools/ab.sh "$LIBS" ;;
  parity)
    for v in $VARIANTS; do
      lib=$PWD/lightglue_amd/liblightglue_amd_sched_${v%%:*}.so
      echo "== $lib"; LIGHTGLUE_AMD_LIB=$lib python -m pytest tests/test_gpu_parity.py tests/test_gpu_properties.py -m gpu -q -x 2>&1 | tail -2
    done ;;
  *) echo "usage: $0 build|ab|parity"; exit 2 ;;
esac
