#!/bin/bash
# r05: long-list segment split (gs_render.h: TileSegments), same box A/B on the from-disk schedule with 1.2 M SfM points:
# GSPLAT_NO_SEGMENTS=1 (whole lists) against the default, stage times of the trained state + the run's throughput.
cd $GRAFT_REPO_ROOT
export GSPLAT_NO_BUILD=1 GSPLAT_NO_RENDER_DUMPS=1
python tools/make_colmap_dataset.py /tmp/ds --points 1200000 > /tmp/dataset.log 2>&1 && python tools/write_config.py /tmp/garden.yaml > /dev/null 2>&1 || exit 1
for rep in 1 2; do
for mode in 1 0; do
  GSPLAT_NO_SEGMENTS=$mode GSPLAT_DEBUG_STAGES=1 python train.py /tmp/garden.yaml /tmp/ds > /tmp/train_$mode.log 2>&1 || { tail -20 /tmp/train_$mode.log; exit 1; }
  echo "== GSPLAT_NO_SEGMENTS=$mode (rep $rep)"
  grep -E "stages|roofline|training done" /tmp/train_$mode.log | tail -3 | cut -c1-420
done
done
cp /tmp/train_0.log gpurun_out/r05_garden_1200k_points_train_segments.log
