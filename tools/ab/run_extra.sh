#!/bin/bash
# usage: tools/ab/run_extra.sh <steps> <lib tags...>: like run.sh, but with the extra workloads (dense / skewed / large-splat
# scenes): a change that helps the uniform benchmark scene must not cost the scenes that saturate early
steps=$1; shift
for round in 1 2; do
for tag in "$@"; do
  GSPLAT_LIB=tools/ab/lib$tag.so timeout -k 10 400 python bench.py --steps $steps --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms']; e=d['extra_workloads']
print('$tag', round(d['value'],1), 'fwd', s['render_forward'], 'bwd', s['render_backward'], '|', ' '.join('%s %.4f f %.4f b %.4f' % (k, v['ms_per_step'], v['stage_ms']['render_forward'], v['stage_ms']['render_backward']) for k, v in e.items() if 'ms_per_step' in v))" || exit 1
done; done
