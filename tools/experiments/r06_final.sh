#!/bin/bash
# r06 final record: the GPU suite, the bench line, and the 7 000-iteration schedule from disk with 1.2 M SfM points
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q > gpurun_out/r06_gputest_final.log 2>&1; rc=$?
tail -3 gpurun_out/r06_gputest_final.log
[ $rc -ne 0 ] && exit $rc
python bench.py > gpurun_out/r06_bench_final.json 2> gpurun_out/r06_bench_final.err || exit 1
python - <<PY
import json
d=json.load(open("gpurun_out/r06_bench_final.json"))
print("bench", d["value"], d["ms_per_step"], "host", d["reference_host_path"].get("ms_per_iteration"), "train", d["train_step_ms_with_loss_and_adam"])
x=d["exchange_host_cost_one_rank"]; print({k:(v["ms_per_step_with_exchange"] if isinstance(v,dict) else v) for k,v in x.items() if k not in("workload","backend")})
PY
python tools/make_colmap_dataset.py /tmp/ds --points 1200000 > /dev/null 2>&1 && python tools/write_config.py /tmp/garden.yaml > /dev/null 2>&1 && \
  GSPLAT_NO_RENDER_DUMPS=1 GSPLAT_DEBUG_STAGES=1 python train.py /tmp/garden.yaml /tmp/ds > gpurun_out/r06_garden_1200k_points_train.log 2>&1
tail -6 gpurun_out/r06_garden_1200k_points_train.log | cut -c1-300
