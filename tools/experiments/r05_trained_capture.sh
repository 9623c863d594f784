#!/bin/bash
# r05: the compositing kernels on a TRAINED capture (the 7 000-iteration schedule from disk with 1.2 M SfM points):
# stage times + S_eff roofline (GSPLAT_DEBUG_STAGES), rocprofv3 kernel summary of the whole run, and the PMC passes
# (VALU instructions, FETCH_SIZE, WRITE_SIZE: separate runs, never combined with tracing) averaged over the run's launches.
cd $GRAFT_REPO_ROOT
export GSPLAT_NO_BUILD=1 GSPLAT_NO_RENDER_DUMPS=1
OUT=/tmp/r05_trained
mkdir -p $OUT gpurun_out
python tools/make_colmap_dataset.py /tmp/ds --points 1200000 > $OUT/dataset.log 2>&1 && python tools/write_config.py /tmp/garden.yaml > /dev/null 2>&1 || exit 1
GSPLAT_DEBUG_STAGES=1 python train.py /tmp/garden.yaml /tmp/ds > gpurun_out/r05_garden_1200k_points_train.log 2>&1
grep -E "stages|roofline|training done" gpurun_out/r05_garden_1200k_points_train.log | cut -c1-600
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $GRAFT_REPO_ROOT/train.py /tmp/garden.yaml /tmp/ds > $OUT/stats.log 2>&1
for pass in "sq SQ_INSTS_VALU SQ_WAVES" "fetch FETCH_SIZE" "write WRITE_SIZE"; do
  set -- $pass; name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $GRAFT_REPO_ROOT/train.py /tmp/garden.yaml /tmp/ds > $OUT/$name.log 2>&1 || echo "pass $name failed"
  echo "pass $name done"
done
cd $GRAFT_REPO_ROOT
cp "$(ls -t $OUT/stats/*/*kernel_stats.csv | head -1)" gpurun_out/r05_garden_1200k_points_train_kernel_stats.csv
python3 profiles/summarize_pmc.py $OUT > gpurun_out/r05_pmc_summary_garden_1200k_points_train.json
rm -rf $OUT  # (the per-dispatch traces of 7 000 iterations are hundreds of MB: only the summaries travel back)
head -8 gpurun_out/r05_garden_1200k_points_train_kernel_stats.csv | cut -c1-160
python3 - <<PY
import json
d=json.load(open("gpurun_out/r05_pmc_summary_garden_1200k_points_train.json"))
for k in ("render_bwd_kernel","render_fwd_kernel"):
    v=d.get(k,{}); print(k, {c: round(v[c],1) for c in v})
PY
