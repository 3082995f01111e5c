#!/bin/bash
# Round 6, call k: does staggering the waves behind the attention's barrier (s_sleep by wave index) change time / power?  (burst hypothesis behind call j)
O=gpurun_out/r06k; rm -rf $O; mkdir -p $O
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$1', round(d['value'],1), round(d['ms_per_step'],3), {x: round(k[x],3) for x in ('attn_self','attn_cross','fused_tail') if x in k}, 'W', (d.get('power') or {}).get('board_power_w_median'), 'sclk', (d.get('power') or {}).get('sclk_mhz_median'), (d['parity'] or {}).get('index_mismatches'))"; }
for round in 1 2; do for v in tree stag1 stag2 stag3; do
  if [ $v = tree ]; then L=$PWD/lightglue_amd/liblightglue_amd.so; else L=$PWD/build_variants/liblightglue_amd_$v.so; fi
  LIGHTGLUE_AMD_LIB=$L timeout 120 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-gather-probe 2>/dev/null | tail -1 | line $v
done; done 2>&1 | tee $O/ab_attention_stagger.log
