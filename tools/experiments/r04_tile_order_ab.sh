#!/bin/bash
# r04: the backward's tiles heaviest first (tile_order_kernel) against the plain XCD-run order (GSPLAT_NO_TILE_ORDER=1),
# same box, alternating; per-stage times of the headline scene and of the garden-shaped workload
export GSPLAT_BENCH_REFERENCE_HOST=0 GSPLAT_BENCH_EXCHANGE_HOST_COST=0 GSPLAT_BENCH_ALTERNATING=0 GSPLAT_BENCH_TRAIN_STEP=0
for round in 1 2; do
for off in 0 1; do
  GSPLAT_NO_TILE_ORDER=$off timeout -k 10 400 python bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms']; g=d['extra_workloads']['garden1200k']; h=d['extra_workloads']['config3_halfculled']
print('order_off=$off', round(d['value'],1), 'bwd_evt', round(d['roofline']['avg_launch_ms'],4), 'fwd', s['render_forward'], 'bwd', s['render_backward'], '| garden step', g['ms_per_step'], 'fwd', g['stage_ms']['render_forward'], 'bwd', g['stage_ms']['render_backward'], '| halfculled', h['ms_per_step'])" || exit 1
done; done
