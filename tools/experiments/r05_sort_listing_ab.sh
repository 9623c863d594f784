#!/bin/bash
# r05: the workgroup sort classes list their tiles with parallel loads (sl1) instead of walking their candidates (sl0)
cd $GRAFT_REPO_ROOT
export GSPLAT_NO_BUILD=1 GSPLAT_NO_RENDER_DUMPS=1
python tools/make_colmap_dataset.py /tmp/ds --points 1200000 > /tmp/dataset.log 2>&1 && python tools/write_config.py /tmp/garden.yaml > /dev/null 2>&1 || exit 1
for rep in 1 2; do
for tag in sl0 sl1; do
  GSPLAT_LIB=tools/ab/lib$tag.so GSPLAT_DEBUG_STAGES=1 python train.py /tmp/garden.yaml /tmp/ds > /tmp/train_$tag.log 2>&1 || { tail -20 /tmp/train_$tag.log; exit 1; }
  echo "== $tag (rep $rep)"
  grep -E "stages|training done" /tmp/train_$tag.log | tail -2 | cut -c1-300
done
done
