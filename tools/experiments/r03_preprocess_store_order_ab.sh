for round in 1 2; do for tag in early late; do
GSPLAT_LIB=tools/ab/lib$tag.so timeout -k 10 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-extra-workloads 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', round(d['value'],1), 'pre_lean', d['stage_ms']['preprocess'], 'pre_full', d['preprocess_ms_all_forward_outputs'])"
done; done
