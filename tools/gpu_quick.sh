#!/bin/bash
# one gpurun call: parity tests, phase clocks, A/B bench against the round-1 library
LIBS=${LIBS:-"lightglue_amd/_ab/liblightglue_amd_r1.so lightglue_amd/liblightglue_amd.so"}
python -m pytest tests/test_gpu_parity.py tests/test_gpu_properties.py -x -q -m gpu 2>&1 | tail -5
python tools/tail_timing.py bf16x3 1 2>&1 | grep -v amdgpu.ids | tail -8
python tools/tail_timing.py bf16x3 2 2>&1 | grep -v amdgpu.ids | tail -8
bash tools/ab.sh "$LIBS"
bash tools/ab.sh "$LIBS" --no-fuse-next
