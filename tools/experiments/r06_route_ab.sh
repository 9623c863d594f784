#!/bin/bash
# r06: binning route 1 (LDS counting sort + per-tile sorts) against route 2 (rocPRIM Onesweep on depth bits, then tile bits) on the
# trained-capture shapes
cd $GRAFT_REPO_ROOT
export GSPLAT_NO_BUILD=1
for w in veiled1200k garden1200k dense4m config3; do for r in 1 2; do
  GSPLAT_STATS_ROUTE=$r python tools/workload_stats.py $w 20 2>&1 | grep -E "stage|ms per|bin_sort" | sed "s/^/$w route $r: /" | cut -c1-260
done; done
