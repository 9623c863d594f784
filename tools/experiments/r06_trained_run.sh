#!/bin/bash
# r06: the 7 000-iteration schedule from disk with 1.2 M SfM points, the Trainer's optimizer choreographies side by side
# (GSPLAT_FUSED_ADAM=0: optimizer kernels behind the backward, r01-r05; 1: the default since r06)
cd $GRAFT_REPO_ROOT
python tools/make_colmap_dataset.py /tmp/ds --points 1200000 > /dev/null 2>&1 && python tools/write_config.py /tmp/garden.yaml > /dev/null 2>&1 || exit 1
for mode in 0 1; do
  GSPLAT_FUSED_ADAM=$mode GSPLAT_NO_RENDER_DUMPS=1 GSPLAT_DEBUG_STAGES=1 timeout -k 10 500 python train.py /tmp/garden.yaml /tmp/ds > gpurun_out/r06_garden_1200k_points_train_adam$mode.log 2>&1 || { tail -5 gpurun_out/r06_garden_1200k_points_train_adam$mode.log; exit 1; }
  echo "== GSPLAT_FUSED_ADAM=$mode"; tail -6 gpurun_out/r06_garden_1200k_points_train_adam$mode.log | cut -c1-260
done
