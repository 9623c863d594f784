#!/bin/bash
# usage: tools/ab/build_variant.sh <tag> <gs_render.hip variant> [extra flags]: links tools/ab/lib<tag>.so from the
# current objects with gs_render.o replaced by the given source (same flags as the Makefile)
tag=$1; src=$2; shift 2
cd "$(dirname "$0")/../../3dgs_amd/csrc" || exit 1
cp "$src" ./_variant_render.hip
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -munsafe-fp-atomics -mllvm -amdgpu-atomic-optimizer-strategy=None \
  -Wno-unused-function "$@" -c -o /tmp/_variant_render_$tag.o ./_variant_render.hip || { rm -f ./_variant_render.hip; exit 1; }
rm -f ./_variant_render.hip
hipcc --offload-arch=gfx950 -shared -o ../../tools/ab/lib$tag.so gs_common.o gs_pergaussian.o gs_binning.o /tmp/_variant_render_$tag.o gs_fused.o gs_loss.o gs_init.o gs_density.o
