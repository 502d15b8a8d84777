#!/bin/bash
# An experimental build of the library with extra compiler flags: stenos_amd/lib/exp/libstenos_<name>.so (git-ignored, travels
# to the GPU box; tools/exp_variants.sh and STENOS_LIB_PATH select it).  usage: tools/build_variant.sh <name> [-DSTENOS_...=.. ...]
set -e
name="$1"; shift
here="$(cd "$(dirname "$0")/.." && pwd)"
src="$here/stenos_amd/csrc"
mkdir -p "$here/stenos_amd/lib/exp"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off "$@" -shared -Wl,-Bsymbolic \
  "$src/kernels.hip" "$src/kernels_wide.hip" "$src/byte_kernels.hip" "$src/capi.cpp" "$src/strategy.cpp" -o "$here/stenos_amd/lib/exp/libstenos_$name.so" -ldl 2>/dev/null
echo "built libstenos_$name.so"
