export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "pipeline_stages or fused_next or default_precision_parity or fp32_mode or tail_row_tile" 2>&1 | tail -4
ab() { LIGHTGLUE_AMD_LIB=$PWD/$1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['kernel_ms_per_step']; print('$1', round(d['value']), round(d['ms_per_step'], 3), 'attn', k['attn_self'], 'tail', k['fused_tail'], 'proj0', k['gemm_qkv_self'], d['parity']['index_mismatches'], d['parity']['max_dscore'])"; }
for r in 1 2; do ab build_variants/liblightglue_amd_base.so; ab lightglue_amd/liblightglue_amd.so; done
