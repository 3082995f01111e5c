for round in 1 2; do
for opt in "" "attn_rows=32"; do
LG_BENCH_OPTS="$opt" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-gather-probe 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; p=d['power']
print('opts=[$opt]', round(d['value'],1), {x: round(k[x],3) for x in ('attn_self','attn_cross','fused_tail')}, 'W', p['board_power_w_median'], 'sclk', p['sclk_mhz_median'], 'tail clk', d['shader_clock_mhz_inside_tail_kernel'])"
done; done
