#!/bin/bash
# usage: tools/ab/build_variant.sh <tag> <source.hip> [extra flags]: links tools/ab/lib<tag>.so from the current objects with
# the object of the given source (its basename decides which: gs_render.hip, gs_binning.hip, ...) rebuilt from that file
# with the Makefile's flags plus the extra ones
tag=$1; src=$(readlink -f "$2"); shift 2
name=$(basename "$src" .hip)
case " gs_common gs_pergaussian gs_binning gs_render gs_fused gs_loss gs_init gs_density " in
  *" $name "*) ;;
  *) echo "build_variant.sh: $src is not named after one of the library's objects (the basename decides which one it replaces)"; exit 1 ;;
esac
cd "$(dirname "$0")/../../3dgs_amd/csrc" || exit 1
cp "$src" ./_variant_$name.hip
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -munsafe-fp-atomics -mllvm -amdgpu-atomic-optimizer-strategy=None \
  -Wno-unused-function "$@" -c -o /tmp/_variant_${name}_$tag.o ./_variant_$name.hip || { rm -f ./_variant_$name.hip; exit 1; }
rm -f ./_variant_$name.hip
# the variant says so itself (gsplat_build_flags): stored profiles are never attached to it (3dgs_amd/_lib.py)
make -s gs_source_hash.inc || exit 1
g++ -O2 -std=c++17 -fPIC -DGS_BUILD_FLAGS="\"variant:$tag $*\"" -c -o /tmp/_variant_version_$tag.o gs_version.cpp || exit 1
objs="/tmp/_variant_version_$tag.o"
for o in gs_common gs_pergaussian gs_binning gs_render gs_fused gs_loss gs_init gs_density; do
  if [ "$o" == "$name" ]; then objs="$objs /tmp/_variant_${name}_$tag.o"; else objs="$objs $o.o"; fi
done
hipcc --offload-arch=gfx950 -shared -o ../../tools/ab/lib$tag.so $objs
