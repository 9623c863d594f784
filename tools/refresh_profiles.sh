#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/refresh_profiles.sh <tag>   e.g. r04
# rocprofv3 kernel-trace summary of the bench command, the PMC passes (counters in their own runs, never combined with
# tracing), traffic.json from them -- and THEN the bench line, so that it carries the counters of this very build;
# everything lands under gpurun_out/<tag>_*, from where the summaries are copied to profiles/.
set -e
TAG=${1:-r05}
cd $GRAFT_REPO_ROOT
python -c "import importlib; importlib.import_module('3dgs_amd._lib').build()"   # once, before anything touches the GPU
export GSPLAT_NO_BUILD=1
(
  export GSPLAT_BENCH_TRAIN_STEP=0   # the profiled runs: the headline workload's kernels only,
  # no child processes under the profiler (the C++ reference host, the one-rank RCCL rehearsal), and only the headline
  # view's launches in the per-kernel averages (no alternating views)
  export GSPLAT_BENCH_REFERENCE_HOST=0 GSPLAT_BENCH_EXCHANGE_HOST_COST=0 GSPLAT_BENCH_ALTERNATING=0
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_stats -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra-workloads > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_stats.log 2>&1
  cd $GRAFT_REPO_ROOT
  cp "$(ls -t gpurun_out/${TAG}_stats/*/*kernel_stats.csv | head -1)" gpurun_out/${TAG}_kernel_stats.csv
  bash profiles/run_pmc.sh gpurun_out/${TAG}_pmc > gpurun_out/${TAG}_pmc.log 2>&1
  python3 profiles/summarize_pmc.py gpurun_out/${TAG}_pmc > gpurun_out/${TAG}_pmc_summary.json
  # r05: the same counters and kernel summary on the garden-shaped workload (a trained capture's density: large splats,
  # lists of ~1150 entries, 40 % culled), whose roofline entries the bench line carries next to the uniform scene's
  bash profiles/run_pmc.sh gpurun_out/${TAG}_pmc_garden1200k garden1200k > gpurun_out/${TAG}_pmc_garden1200k.log 2>&1
  python3 profiles/summarize_pmc.py gpurun_out/${TAG}_pmc_garden1200k > gpurun_out/${TAG}_pmc_summary_garden1200k.json
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_stats_garden1200k -- python3 $GRAFT_REPO_ROOT/tools/workload_stats.py garden1200k 30 > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_stats_garden1200k.log 2>&1)
  cp "$(ls -t gpurun_out/${TAG}_stats_garden1200k/*/*kernel_stats.csv | head -1)" gpurun_out/${TAG}_garden1200k_kernel_stats.csv
  python3 profiles/make_traffic.py gpurun_out/${TAG}_pmc_summary.json garden1200k=gpurun_out/${TAG}_pmc_summary_garden1200k.json > gpurun_out/${TAG}_traffic.json
)
cp gpurun_out/${TAG}_traffic.json profiles/traffic.json   # (on the box's copy of the tree: bench.py reads it from there)
python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
head -12 gpurun_out/${TAG}_kernel_stats.csv
