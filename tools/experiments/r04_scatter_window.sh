#!/bin/bash
# r04: what would bin_scatter gain if its isolated 8-byte stores cost nothing at HBM?  The variant
#   tools/ab/build_variant.sh scw 3dgs_amd/csrc/gs_binning.hip -DGS_SCATTER_WINDOW=1
# sends every placement into a 128 KB window (L2-resident; the lists are garbage, so only bin_scatter_kernel's own duration
# means anything).  Per-kernel times of both libraries from rocprofv3 (kernel trace only), same box, back to back.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export GSPLAT_NO_BUILD=1 GSPLAT_BENCH_TRAIN_STEP=0 GSPLAT_BENCH_REFERENCE_HOST=0 GSPLAT_BENCH_EXCHANGE_HOST_COST=0 GSPLAT_BENCH_ALTERNATING=0
for tag in base scw base scw; do
  lib=$R/3dgs_amd/libgsplat_hip.so; [ $tag == scw ] && lib=$R/tools/ab/libscw.so
  rm -rf $R/gpurun_out/scw_$tag
  GSPLAT_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/scw_$tag -- python3 $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-extra-workloads > $R/gpurun_out/scw_$tag.log 2>&1
  f=$(ls -t $R/gpurun_out/scw_$tag/*/*kernel_stats.csv | head -1)
  echo "$tag $(grep -E 'bin_scatter_kernel' $f | head -1 | cut -d, -f1-4)"
done
