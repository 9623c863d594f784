#!/bin/bash
# r06: preprocess_bwd_kernel<3, 3> (the per-gaussian backward behind sh_adam_dir_kernel) compiled for 3 (110 VGPRs: 4), 5
# (96) and 6 (80, 26 spilled) waves per SIMD: kernel times under rocprofv3
cd $GRAFT_REPO_ROOT
for tag in bw5 bw6; do
  echo "== $tag"
  rm -rf gpurun_out/r06_adam_prof
  if [ $tag != default ]; then export GSPLAT_LIB=$GRAFT_REPO_ROOT/tools/ab/lib$tag.so; fi
  bash tools/experiments/r06_fused_adam_prof.sh 2>&1 | grep "sh_adam_dir\|preprocess_bwd_kernel<3, 3>\|preprocess_bwd_kernel<3, 0>" || exit 1
done
