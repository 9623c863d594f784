#!/bin/bash
# r04: the forward's stable coarse tile order on skewed scenes (GSPLAT_FWD_TILE_ORDER=0 switches it off), caps of 0.5 / 1 / 2
# average list lengths; same box, alternating; headline + the extra workloads
export GSPLAT_BENCH_REFERENCE_HOST=0 GSPLAT_BENCH_EXCHANGE_HOST_COST=0 GSPLAT_BENCH_ALTERNATING=0 GSPLAT_BENCH_TRAIN_STEP=0 GSPLAT_NO_BUILD=1
for round in 1 2; do
for cfg in "0 1.0" "1 0.5" "1 1.0" "1 2.0"; do
  set -- $cfg
  GSPLAT_FWD_TILE_ORDER=$1 GSPLAT_FWD_ORDER_CAP=$2 timeout -k 10 400 python bench.py --steps 100 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms']; e=d['extra_workloads']
print('fwd_order=$1 cap=$2', round(d['value'],1), 'fwd', s['render_forward'], 'bwd', s['render_backward'], '|', ' '.join('%s %.4f f %.4f b %.4f' % (k, v['ms_per_step'], v['stage_ms']['render_forward'], v['stage_ms']['render_backward']) for k, v in e.items() if k in ('garden1200k','dense4m','bigsplats')))" || exit 1
done; done
