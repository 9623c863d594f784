#!/bin/bash
# r05: segment length of the backward's long-list split: tools/ab/libseg{496,992,1488}.so (gs_render + gs_fused built with
# -DGS_SEG_ENTRIES=...), from-disk schedule with 1.2 M SfM points, stage times of the trained state
cd $GRAFT_REPO_ROOT
export GSPLAT_NO_BUILD=1 GSPLAT_NO_RENDER_DUMPS=1
python tools/make_colmap_dataset.py /tmp/ds --points 1200000 > /tmp/dataset.log 2>&1 && python tools/write_config.py /tmp/garden.yaml > /dev/null 2>&1 || exit 1
for rep in 1; do
for sz in ${SEG_TAGS:-992m1984 496m992 496m1488 248m992 248m744}; do
  GSPLAT_LIB=tools/ab/libseg$sz.so GSPLAT_DEBUG_STAGES=1 python train.py /tmp/garden.yaml /tmp/ds > /tmp/train_$sz.log 2>&1 || { tail -20 /tmp/train_$sz.log; exit 1; }
  echo "== segments of $sz (rep $rep)"
  grep -E "stages|training done" /tmp/train_$sz.log | tail -2 | cut -c1-300
done
done
