#!/bin/bash
# r05: kernel summary of workload_stats.py on a workload with the forward's segments on / off
cd $GRAFT_REPO_ROOT
export GSPLAT_NO_BUILD=1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for wl in ${WORKLOADS:-garden1200k dense4m}; do
for mode in 1 0; do
  export GSPLAT_NO_FWD_SEGMENTS=$mode
  rm -rf /tmp/trace_$wl_$mode
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trace_${wl}_$mode -- python3 $R/tools/workload_stats.py $wl 12 > /tmp/trace_${wl}_$mode.log 2>&1 || { tail -5 /tmp/trace_${wl}_$mode.log; exit 1; }
  echo "== $wl GSPLAT_NO_FWD_SEGMENTS=$mode"
  python3 - /tmp/trace_${wl}_$mode <<'PY'
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True))[-1]
for r in list(csv.DictReader(open(f)))[:14]:
    print("  %-70s calls %5s avg %9.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
done
