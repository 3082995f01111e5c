#!/bin/bash
# round 6 call y2: sim_planes_kernel, image-1 rows per workgroup 128 / 256 / 512 (stores right behind their MFMAs)
ab() {
  for round in 1 2; do
    for name in p0c128 p0c256 p0c512; do
      LIGHTGLUE_AMD_LIB=$PWD/build_variants/liblightglue_amd_$name.so timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-gather-probe "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; p=d.get('parity') or {}; print('$name', '$*', round(d['value'],1), round(d['ms_per_step'],3), {x: round(k[x],4) for x in ('sim','assign','gemm_final_proj') if x in k}, p.get('index_mismatches'))"
    done
  done
}
ab
ab --config 3 --inflight 1
ab --config 5 --inflight 1
ab --config 4 --steps 6 --warmup 2
