#!/bin/bash
# r06: the per-gaussian forward -- GSPLAT_PRE_SPLIT=0 the fused preprocess_kernel, 1 sh_colour_kernel then
# preprocess_geom_kernel on one stream, 2 the two side by side on two streams (2l: joined only in front of render_fwd);
# same library, same box, two alternating rounds
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_fused_gpu.py -q -x -k "preprocess_split or lean_forward_and_compacted" > gpurun_out/r06_pre_split_tests.log 2>&1 || { tail -30 gpurun_out/r06_pre_split_tests.log; exit 1; }
tail -2 gpurun_out/r06_pre_split_tests.log
export GSPLAT_BENCH_REFERENCE_HOST=0 GSPLAT_BENCH_EXCHANGE_HOST_COST=0 GSPLAT_BENCH_ALTERNATING=0
for round in 1 2; do
for v in 0 1 2 3; do
  GSPLAT_PRE_JOIN=$([ $v == 2l ] && echo late || echo early) GSPLAT_PRE_SPLIT=${v:0:1} timeout -k 10 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-extra-workloads 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms']; print('split=$v', round(d['value'],1), 'ms', round(d['ms_per_step'],4), 'lean_step', round(d['ms_per_step_lean_forward'],4), 'pre_full', d['preprocess_ms_all_forward_outputs'], 'pre_lean', d['preprocess_ms_lean_forward'], 'sort', s['bin_sort'], 'fwd', s['render_forward'], 'bwd', s['render_backward'], 'pbwd', s['preprocess_backward'], 'cull', s['project_cull'], 'train', round(d['train_step_ms_with_loss_and_adam'],4), 'fps', round(d['render_fps_forward_only']))" || exit 1
done; done
