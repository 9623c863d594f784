#!/bin/bash
# r05: the forward's long-list segments (gs_render.h: FwdSegments), same box, same library: GSPLAT_NO_FWD_SEGMENTS=1 against
# the default on (a) the from-disk schedule with 1.2 M SfM points, (b) bench.py's workloads
cd $GRAFT_REPO_ROOT
export GSPLAT_NO_BUILD=1 GSPLAT_NO_RENDER_DUMPS=1
python tools/make_colmap_dataset.py /tmp/ds --points 1200000 > /tmp/dataset.log 2>&1 && python tools/write_config.py /tmp/garden.yaml > /dev/null 2>&1 || exit 1
for rep in 1 2; do
for mode in 1 0; do
  GSPLAT_NO_FWD_SEGMENTS=$mode GSPLAT_DEBUG_STAGES=1 python train.py /tmp/garden.yaml /tmp/ds > /tmp/train_$mode.log 2>&1 || { tail -20 /tmp/train_$mode.log; exit 1; }
  echo "== GSPLAT_NO_FWD_SEGMENTS=$mode (rep $rep)"
  grep -E "stages|training done" /tmp/train_$mode.log | tail -2 | cut -c1-300
done
done
for rep in 1 2; do
for mode in 1 0; do
  GSPLAT_NO_FWD_SEGMENTS=$mode timeout -k 10 400 python bench.py --steps ${STEPS:-100} --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms']; e=d['extra_workloads']
print('NO_FWD_SEGMENTS=$mode', round(d['value'],1), 'fwd', s['render_forward'], 'bwd', s['render_backward'], '|', ' '.join('%s %.4f f %.4f b %.4f' % (k, v['ms_per_step'], v['stage_ms']['render_forward'], v['stage_ms']['render_backward']) for k, v in e.items()))" || exit 1
done
done
