#!/bin/bash
# r06: per-kernel times of the training iteration, optimizer step inside the backward against behind it
cd $GRAFT_REPO_ROOT
export GSPLAT_NO_BUILD=1 GSPLAT_BENCH_REFERENCE_HOST=0 GSPLAT_BENCH_EXCHANGE_HOST_COST=0 GSPLAT_BENCH_ALTERNATING=0
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r06_adam_prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extra-workloads > $GRAFT_REPO_ROOT/gpurun_out/r06_adam_prof.log 2>&1)
f=$(ls -t gpurun_out/r06_adam_prof/*/*kernel_stats.csv | head -1)
python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name']
    if any(k in n for k in ('preprocess_bwd','optimizer_','loss_','render_bwd','sh_adam_dir')):
        print(f"{n[:110]:110s} calls {r['Calls']:>4s} avg_us {float(r['AverageNs'])/1e3:8.1f}")
PY
