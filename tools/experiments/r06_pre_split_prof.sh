#!/bin/bash
# r06: per-kernel times (rocprofv3 --kernel-trace --stats) of the per-gaussian forward, fused against split, lean and full
cd $GRAFT_REPO_ROOT
python -c "import importlib; importlib.import_module('3dgs_amd._lib').build()"
export GSPLAT_NO_BUILD=1
for split in 0 1 3; do for full in 0 1; do
  tag=split${split}_full${full}
  (cd /tmp && export TMPDIR=/tmp && GSPLAT_PRE_SPLIT=$split GSPLAT_STATS_FULL=$full rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r06_pre_$tag -- python3 $GRAFT_REPO_ROOT/tools/workload_stats.py config3 30 > $GRAFT_REPO_ROOT/gpurun_out/r06_pre_$tag.log 2>&1) || { tail -5 gpurun_out/r06_pre_$tag.log; exit 1; }
  f=$(ls -t gpurun_out/r06_pre_$tag/*/*kernel_stats.csv | head -1)
  echo "== $tag"; grep -E "preprocess|sh_colour|project_cull|bin_scatter|bin_offsets" $f | cut -d, -f1-4 | cut -c1-220
done; done
