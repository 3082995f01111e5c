#!/bin/bash
# phase clocks of the fused tail for several library builds in one gpurun call:  tools/phase_ab.sh "lib1 lib2"
for lib in $1; do echo "== $lib"; LIGHTGLUE_AMD_LIB=$PWD/$lib python tools/tail_timing.py bf16x3 1 2>&1 | grep -v amdgpu.ids | tail -7 | head -6; done
