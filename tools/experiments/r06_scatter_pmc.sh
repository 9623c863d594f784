#!/bin/bash
# r06: L2 -> fabric write requests of bin_scatter_kernel, direct stores against LDS-staged band-by-band stores
cd $GRAFT_REPO_ROOT
export GSPLAT_NO_BUILD=1 GSPLAT_PRE_SPLIT=0
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  OUT=gpurun_out/r06_scatter_pmc_staged$v; mkdir -p $R/$OUT
  run() { local name=$1; shift
    GSPLAT_SCATTER_STAGED=$v timeout -k 10 150 rocprofv3 --pmc "$@" --output-format csv -d $R/$OUT/$name -- python3 $R/tools/workload_stats.py config3 6 > $R/$OUT/$name.log 2>&1 || echo "pass $name failed"; }
  run w1 TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
  run w2 TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
  run w3 TCC_HIT_sum TCC_MISS_sum
  run w4 TCC_WRITE_sum TCC_WRITEBACK_sum
  (cd $R && python3 profiles/summarize_pmc.py $OUT > $OUT/summary.json 2>/dev/null; python3 - $OUT/summary.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
for k,v in d.items():
    if any(s in k for s in ("bin_scatter","tile_depth_sort_wave")):
        print(sys.argv[1].split('/')[-2], k[:40], {a:round(b) for a,b in v.items()})
PY
)
done
