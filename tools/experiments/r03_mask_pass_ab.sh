for round in 1 2; do for mp in 0 1; do
GSPLAT_MASK_PASS=$mp timeout -k 10 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-extra-workloads 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms']; print('mask_pass=$mp', round(d['value'],1), 'fwd_stage', s['render_forward'], 'bwd', s['render_backward'], 'fps', round(d['render_fps_forward_only'],1), round(d['render_fps_render_only_context'],1))"
done; done
