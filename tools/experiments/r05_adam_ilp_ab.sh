#!/bin/bash
# r05: optimizer_step_kernel with 1 / 4 / 8 elements per thread (tools/ab/libai{1,4,8}.so: gs_loss.hip with -DGS_ADAM_ILP=n)
cd $GRAFT_REPO_ROOT
export GSPLAT_NO_BUILD=1 GSPLAT_NO_RENDER_DUMPS=1
for rep in 1 2; do
for tag in ${TAGS:-ai1 ai4 ai8}; do
  GSPLAT_LIB=tools/ab/lib$tag.so timeout -k 10 300 python bench.py --steps 100 --warmup 30 --no-cpu-baseline --no-extra-workloads 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', round(d['value'],1), 'train step with loss and adam', round(d['train_step_ms_with_loss_and_adam'],4))" || exit 1
done
done
python tools/make_colmap_dataset.py /tmp/ds --points 1200000 > /tmp/dataset.log 2>&1 && python tools/write_config.py /tmp/garden.yaml > /dev/null 2>&1 || exit 1
for tag in ${TAGS:-ai1 ai4 ai8}; do
  GSPLAT_LIB=tools/ab/lib$tag.so GSPLAT_DEBUG_STAGES=1 python train.py /tmp/garden.yaml /tmp/ds > /tmp/train_$tag.log 2>&1 || { tail -20 /tmp/train_$tag.log; exit 1; }
  echo "== $tag"
  grep -E "stages|training done" /tmp/train_$tag.log | tail -2 | cut -c1-200
done
