#!/bin/bash
# usage: tools/ab/build_seg_variants.sh "<entries> <split_min>" ...: tools/ab/libseg<entries>m<split_min>.so, the library with
# gs_render and gs_fused built for that segment length / split threshold (gs_render.h: TileSegments)
cd "$(dirname "$0")/../../3dgs_amd/csrc" || exit 1
make -s || exit 1
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -munsafe-fp-atomics -mllvm -amdgpu-atomic-optimizer-strategy=None -Wno-unused-function"
for v in "$@"; do
  set -- $v; tag=seg$1m$2
  (hipcc $FLAGS -DGS_SEG_ENTRIES=$1 -DGS_SEG_SPLIT_MIN=$2 -c -o /tmp/seg_render_$tag.o gs_render.hip &
   hipcc $FLAGS -DGS_SEG_ENTRIES=$1 -DGS_SEG_SPLIT_MIN=$2 -c -o /tmp/seg_fused_$tag.o gs_fused.hip & wait)
  g++ -O2 -std=c++17 -fPIC -DGS_BUILD_FLAGS="\"variant:$tag\"" -c -o /tmp/seg_version_$tag.o gs_version.cpp &&
  hipcc --offload-arch=gfx950 -shared -o ../../tools/ab/lib$tag.so /tmp/seg_version_$tag.o gs_common.o gs_pergaussian.o \
    gs_binning.o /tmp/seg_render_$tag.o /tmp/seg_fused_$tag.o gs_loss.o gs_init.o gs_density.o || { echo "FAILED: $tag"; exit 1; }
done
