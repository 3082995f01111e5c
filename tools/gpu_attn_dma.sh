#!/bin/bash
# one gpurun call: LDS-DMA attention kernel — bit identity against the register-staged kernel, then A/B on the bench workload
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lds_dma" 2>&1 | tail -3
bash tools/ab_opt.sh "attn_dma=1" "attn_dma=0" "attn_dma=2"
