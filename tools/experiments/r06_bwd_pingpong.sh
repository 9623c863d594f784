#!/bin/bash
# r06: the backward with two half-batches in flight (render_bwd_pp_kernel, GSPLAT_BWD_PINGPONG=1):
# its parity rows first, then the headline both ways on one box (two alternating rounds), then the trained-capture shape
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_bwd_pingpong_gpu.py -m gpu -q -x > gpurun_out/r06_pp_tests.log 2>&1 || { tail -40 gpurun_out/r06_pp_tests.log; exit 1; }
tail -2 gpurun_out/r06_pp_tests.log
export GSPLAT_BENCH_REFERENCE_HOST=0 GSPLAT_BENCH_EXCHANGE_HOST_COST=0 GSPLAT_BENCH_ALTERNATING=0
for round in 1 2; do
for v in 0 1; do
  GSPLAT_BWD_PINGPONG=$v timeout -k 10 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-extra-workloads 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms']; print('pingpong=$v', round(d['value'],1), 'bwd_evt', round(d['roofline']['avg_launch_ms'],4), 'fwd', s['render_forward'], 'bwd', s['render_backward'], 'pre', s['preprocess'], 'sort', s['bin_sort'], 'pbwd', s['preprocess_backward'])" || exit 1
done; done
# where the wave cycles go, both kernels (-DGS_STAMP=1 build of the same sources)
for v in 0 1; do
  echo "== stamps, GSPLAT_BWD_PINGPONG=$v"
  GSPLAT_BWD_PINGPONG=$v GSPLAT_LIB=tools/ab/libstamp.so timeout -k 10 300 python tools/bwd_timeline.py config3 2>&1 | grep -v "^waves alive\|XCC" || exit 1
done
