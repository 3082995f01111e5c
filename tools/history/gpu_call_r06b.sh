#!/bin/bash
# Round 6, call b: the shader clock the two big kernels run at inside a forward (power management), one-barrier vs ping-pong attention.
O=gpurun_out/r06b; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
for v in 0 1 4; do echo "== attn_pp=$v"; LG_BENCH_OPTS="attn_pp=$v" LIGHTGLUE_AMD_LIB=$PWD/build_variants/liblightglue_amd_attn_wall.so timeout 200 python tools/attn_wall.py 2>&1 | grep -v "^live\|amdgpu.ids"; done | tee $O/attn_wall.log
timeout 200 python tools/tail_wall.py 2>&1 | grep -v "^live\|amdgpu.ids" | tee $O/tail_wall.log
( rocm-smi --showpower --showclocks --showmaxpower 2>&1 | head -40 ) > $O/rocm_smi_idle.log
( python bench.py --steps 3000 --warmup 5 --no-cpu-baseline --no-calibration --no-gather-probe > $O/bench_long.json 2>/dev/null & BP=$!; sleep 12; for i in 1 2 3 4 5 6; do rocm-smi --showpower --showclocks 2>&1 | grep -E "Power|sclk|mclk|fclk" ; sleep 1.5; done; wait $BP ) > $O/rocm_smi_load.log 2>&1
tail -1 $O/bench_long.json | cut -c1-300
cat $O/rocm_smi_load.log | head -40
