#!/bin/bash
# usage: tools/ablate.sh <tag> <EXTRA flags...>: rebuilds gs_render.o with the flags on the GPU box and runs the bench
tag=$1; shift
touch 3dgs_amd/csrc/gs_render.hip 3dgs_amd/csrc/gs_fused.hip
make -C 3dgs_amd/csrc -s EXTRA="$*" > gpurun_out/ablate_build_$tag.log 2>&1 || exit 1
timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', round(d['value'],1), d['stage_ms'])"
