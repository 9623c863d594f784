#!/bin/bash
# usage: tools/ab/run.sh <steps> <lib tags...>: same-box A/B of prebuilt library variants (tools/ab/lib<tag>.so, loaded
# through GSPLAT_LIB), two alternating rounds
steps=$1; shift
for round in 1 2; do
for tag in "$@"; do
  GSPLAT_LIB=tools/ab/lib$tag.so timeout -k 10 300 python bench.py --steps $steps --warmup 30 --no-cpu-baseline --no-extra-workloads 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms']; print('$tag', round(d['value'],1), 'bwd_evt', round(d['roofline']['avg_launch_ms'],4), 'fwd', s['render_forward'], 'bwd', s['render_backward'], 'pre', s['preprocess'], 'sort', s['bin_sort'], 'pbwd', s['preprocess_backward'], 'cull', s['project_cull'])" || exit 1
done; done
