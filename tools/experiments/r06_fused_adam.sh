#!/bin/bash
# r06: the optimizer step inside the per-gaussian backward: parity tests, then the training iteration both ways
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_optimizer_gpu.py tests/test_trainer_gpu.py -q -x > gpurun_out/r06_fused_adam_tests.log 2>&1 || { tail -30 gpurun_out/r06_fused_adam_tests.log; exit 1; }
tail -2 gpurun_out/r06_fused_adam_tests.log
export GSPLAT_BENCH_REFERENCE_HOST=0 GSPLAT_BENCH_EXCHANGE_HOST_COST=0 GSPLAT_BENCH_ALTERNATING=0
for round in 1 2; do
  timeout -k 10 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-extra-workloads 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), 'ms', round(d['ms_per_step'],4), 'train unfused', round(d['train_step_ms_optimizer_kernels_behind_the_backward'],4), 'partial', round(d['train_step_ms_small_groups_inside_the_backward'],4), 'all inside', round(d['train_step_ms_adam_inside_the_backward'],4))" || exit 1
done
