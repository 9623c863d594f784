#!/bin/bash
# r03: compositing kernels launched with k workgroups per CU (unused dynamic LDS), by workload: is a launch of 2.12 rounds
# at k = 8 faster as 2.83 rounds at k = 6?  usage (GPU box): bash tools/experiments/r03_workgroups_per_cu.sh
for w in garden1200k config3 bigsplats; do
  for k in 8 7 6 5; do
    echo -n "$w bwd_k=$k fwd_k=7: "; GSPLAT_BWD_WG_PER_CU=$k GSPLAT_FWD_WG_PER_CU=7 python tools/workload_stats.py $w 30 2>&1 | tail -1
  done
  for k in 6 5; do
    echo -n "$w bwd_k=8 fwd_k=$k: "; GSPLAT_BWD_WG_PER_CU=8 GSPLAT_FWD_WG_PER_CU=$k python tools/workload_stats.py $w 30 2>&1 | tail -1
  done
done
