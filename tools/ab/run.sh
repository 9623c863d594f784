#!/bin/bash
# usage: tools/ab/run.sh <steps> <lib tags...>: same-box A/B of prebuilt library variants (tools/ab/lib<tag>.so)
steps=$1; shift
cp 3dgs_amd/libgsplat_hip.so /tmp/lib_keep.so
for round in 1 2; do
for tag in "$@"; do
  cp tools/ab/lib$tag.so 3dgs_amd/libgsplat_hip.so
  timeout -k 10 200 python bench.py --steps $steps --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms']; print('$tag', round(d['value'],1), 'fwd', s['render_forward'], 'bwd', s['render_backward'], 'pre', s['preprocess'], 'sort', s['bin_sort'], 'pbwd', s['preprocess_backward'], 'cull', s['project_cull'])" || exit 1
done; done
cp /tmp/lib_keep.so 3dgs_amd/libgsplat_hip.so
