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
WHAT=${1:-"dense host garden"}
cd /tmp && export TMPDIR=/tmp
for k in $WHAT; do
  case $k in
    dense) rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/dense -- python3 $GRAFT_REPO_ROOT/tools/dense_profile.py 1 > $OUT/dense.log 2>&1 ;;
    host) rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/host -- $GRAFT_REPO_ROOT/tests/cpp/reference_host /tmp/config3_scene.bin - 20 > $OUT/host.log 2>&1 ;;
    garden) rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/garden -- python3 $GRAFT_REPO_ROOT/tools/workload_stats.py garden1200k 30 > $OUT/garden.log 2>&1 ;;
  esac
done
cd $GRAFT_REPO_ROOT
for k in $WHAT; do
  f=$(ls -t $OUT/$k/*/*kernel_stats.csv | head -1)
  cp "$f" gpurun_out/r05_probe_${k}_kernel_stats.csv
  echo "== $k"; head -14 "$f" | cut -c1-200
done
[ -f $OUT/host.log ] && tail -1 $OUT/host.log | cut -c1-1500
# the same host without the profiler (wall-clock per iteration)
$GRAFT_REPO_ROOT/tests/cpp/reference_host /tmp/config3_scene.bin - 20 | tail -1 | cut -c1-1500
