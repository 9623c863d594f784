#!/bin/bash
# r06: BASELINE configs[1] (1e5 gaussians, 800x800, SH 0, forward only): the fused preprocess_kernel<0> (GSPLAT_PRE_SPLIT=0)
# against preprocess_geom_kernel<0> (=1: the same work, the chunk loop's requests a trip early; no colour kernel at degree 0)
cd $GRAFT_REPO_ROOT
export GSPLAT_NO_BUILD=1
for round in 1 2 3; do for v in 0 1; do
GSPLAT_PRE_SPLIT=$v python - <<PY
import importlib, sys, torch
sys.path.insert(0, ".")
import bench
scene = importlib.import_module("3dgs_amd.scene"); raster = importlib.import_module("3dgs_amd.raster")
o = bench.config2_workload(torch, scene, raster, torch.device("cuda", 0), reps=300)
print("split=$v", round(o["render_fps_render_only_context"]), round(o["render_fps_training_context"]), o["stage_ms"])
PY
done; done
