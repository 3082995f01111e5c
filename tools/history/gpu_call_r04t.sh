#!/bin/bash
# Round-4 call t: phase B — waves 0-3 GELU block then MFMA run, waves 4-7 the opposite order (bmixed), against GELU-first for all
O=gpurun_out/r04t; mkdir -p $O
export TMPDIR=/tmp
NEW=lightglue_amd/liblightglue_amd.so
for round in 1 2 3 4; do for lib in $NEW build_variants/liblightglue_amd_bmixed.so; do
  LIGHTGLUE_AMD_LIB=$PWD/$lib timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$lib', round(d['value']), round(d['ms_per_step'],3), {x: round(k[x],3) for x in ('attn_self','attn_cross','fused_tail') if x in k}, d['parity']['index_mismatches'], d['parity']['max_dscore'])"
done; done 2>&1 | tee $O/ab_cfg2.log
for lib in $NEW build_variants/liblightglue_amd_bmixed.so; do echo "== $lib"; LIGHTGLUE_AMD_LIB=$PWD/$lib timeout 300 python tools/tail_timing.py f16x3 6 2>&1 | grep -E "phaseA|phaseB|total"; done | tee $O/stamps.log
