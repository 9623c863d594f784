set -e
cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/r01_bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/stats -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/stats.log 2>&1
cd $GRAFT_REPO_ROOT
bash profiles/run_pmc.sh gpurun_out/pmc_final > gpurun_out/pmc_final.log 2>&1
python3 profiles/summarize_pmc.py gpurun_out/pmc_final > gpurun_out/r01_pmc_summary.json
ls -t gpurun_out/stats/runc/*kernel_stats.csv | head -1
