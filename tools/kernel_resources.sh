#!/bin/bash
# usage: tools/kernel_resources.sh <file.hip> [name filter] [extra flags]: VGPRs / SGPRs / LDS / scratch of every kernel
# of one source file, compiled device-only with the Makefile's flags
src=${1:-3dgs_amd/csrc/gs_render.hip}; pat=${2:-.}; shift 2
tmp=$(mktemp -d)
hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -munsafe-fp-atomics -mllvm -amdgpu-atomic-optimizer-strategy=None \
  -I$(dirname $src) "$@" --cuda-device-only -c -o $tmp/dev.co $src 2>/dev/null || { echo "compile failed"; exit 1; }
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$tmp/dev.co --output=$tmp/dev.elf
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $tmp/dev.elf | awk '/\.name:/{n=$2} /\.vgpr_count:/{v=$2} /\.sgpr_count:/{s=$2} /\.group_segment_fixed_size:/{l=$2} /\.private_segment_fixed_size:/{p=$2} /\.vgpr_spill_count:/{print n, "vgpr", v, "sgpr", s, "lds", l, "scratch", p, "spill", $2}' | grep -E "$pat" | c++filt | sed 's/(.*)//'
rm -rf $tmp
