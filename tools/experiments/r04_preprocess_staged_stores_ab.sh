for round in 1 2; do for tag in ps0 ps1 ps2 ps3; do
GSPLAT_LIB=tools/ab/lib$tag.so timeout -k 10 300 python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-extra-workloads 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms']; print('$tag', round(d['value'],1), 'pre_lean', s['preprocess'], 'pre_full', d['preprocess_ms_all_forward_outputs'], 'step_full', round(d['ms_per_step_full_forward_outputs'],4))"
done; done
