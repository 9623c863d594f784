#!/bin/bash
# r06: bin_scatter with its stores staged in LDS and written band by band (GSPLAT_SCATTER_STAGED=1) against direct stores (=0)
cd $GRAFT_REPO_ROOT
export GSPLAT_PRE_SPLIT=${GSPLAT_PRE_SPLIT:-0}
python -m pytest tests/test_fused_gpu.py tests/test_ops_gpu.py -q -x -k "sorted or list or binning or config3 or small_scenes or instance_buffers or lean_forward or halfculled or long_lists" > gpurun_out/r06_scatter_tests.log 2>&1 || { tail -30 gpurun_out/r06_scatter_tests.log; exit 1; }
tail -2 gpurun_out/r06_scatter_tests.log
export GSPLAT_BENCH_REFERENCE_HOST=0 GSPLAT_BENCH_EXCHANGE_HOST_COST=0 GSPLAT_BENCH_ALTERNATING=0 GSPLAT_BENCH_TRAIN_STEP=0
for round in 1 2; do
for v in 0 1; do
  GSPLAT_SCATTER_STAGED=$v timeout -k 10 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms']; print('staged=$v', round(d['value'],1), 'ms', round(d['ms_per_step'],4), 'pre', s['preprocess'], 'sort', s['bin_sort'], 'fwd', s['render_forward'], 'bwd', s['render_backward'], 'pbwd', s['preprocess_backward'], '|', ' '.join(k+' '+str(v['stage_ms']['bin_sort']) for k,v in d['extra_workloads'].items()))" || exit 1
done; done
