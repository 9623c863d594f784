#!/bin/bash
# r05: the forward's long-list segments behind their gate (GSPLAT_FWD_SEGMENTS_GATE: longest chain against the work per
# resident workgroup; 0 = always split, 1e9 = never), same box, same library
cd $GRAFT_REPO_ROOT
export GSPLAT_NO_BUILD=1 GSPLAT_NO_RENDER_DUMPS=1
python tools/make_colmap_dataset.py /tmp/ds --points 1200000 > /tmp/dataset.log 2>&1 && python tools/write_config.py /tmp/garden.yaml > /dev/null 2>&1 || exit 1
for gate in ${GATES:-1e9 0 3}; do
  GSPLAT_FWD_SEGMENTS_GATE=$gate GSPLAT_DEBUG_STAGES=1 python train.py /tmp/garden.yaml /tmp/ds > /tmp/train_$gate.log 2>&1 || { tail -20 /tmp/train_$gate.log; exit 1; }
  echo "== gate $gate"
  grep -E "stages|training done" /tmp/train_$gate.log | tail -2 | cut -c1-300
done
for gate in ${GATES:-1e9 0 3}; do
  GSPLAT_FWD_SEGMENTS_GATE=$gate timeout -k 10 400 python bench.py --steps ${STEPS:-60} --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms']; e=d['extra_workloads']
print('gate $gate', round(d['value'],1), 'fwd', s['render_forward'], 'bwd', s['render_backward'], '|', ' '.join('%s %.4f f %.4f b %.4f' % (k, v['ms_per_step'], v['stage_ms']['render_forward'], v['stage_ms']['render_backward']) for k, v in e.items()))" || exit 1
done
