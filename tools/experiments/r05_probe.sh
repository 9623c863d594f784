#!/bin/bash
# r05: per-kernel times of (a) the dense4m forward on the counting route, (b) the C++ reference host's iteration at config 3,
# (c) the garden1200k workload -- rocprofv3 kernel trace only (no counters), each program started directly by the profiler.
set -e
cd $GRAFT_REPO_ROOT
export GSPLAT_NO_BUILD=1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_probe
mkdir -p $OUT
python - <<PY
import importlib, sys
sys.path.insert(0, ".")
scene = importlib.import_module("3dgs_amd.scene"); sio = importlib.import_module("3dgs_amd.scene_io")
N, W, H, L, _ = scene.WORKLOADS["config3"]
sio.write_host_scene("/tmp/config3_scene.bin", scene.make_workload_gaussians("config3"), scene.make_camera(W, H, 0), scene.make_grad_image(W, H), scene.CONFIG, L)
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/dense -- python3 $GRAFT_REPO_ROOT/tools/dense_profile.py 1 > $OUT/dense.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/host -- $GRAFT_REPO_ROOT/tests/cpp/reference_host /tmp/config3_scene.bin - 20 > $OUT/host.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/garden -- python3 $GRAFT_REPO_ROOT/tools/workload_stats.py garden1200k 30 > $OUT/garden.log 2>&1
cd $GRAFT_REPO_ROOT
for k in dense host garden; do
  f=$(ls -t $OUT/$k/*/*kernel_stats.csv | head -1)
  cp "$f" gpurun_out/r05_probe_${k}_kernel_stats.csv
  echo "== $k"; head -14 "$f" | cut -c1-200
done
tail -3 $OUT/host.log | cut -c1-1500
