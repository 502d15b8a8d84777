#!/bin/bash
# An experimental build of the library with extra compiler flags (the product sources hold no experiment switch any more -- tests/test_build_properties.py --: a
# variant is a patch to the sources plus, maybe, generic flags such as -mllvm options): stenos_amd/lib/exp/libstenos_<name>.so (git-ignored, travels
# to the GPU box; tools/exp_variants.sh and STENOS_LIB_PATH select it).  usage: tools/build_variant.sh <name> [-DSTENOS_...=.. ...]
set -e
name="$1"; shift
here="$(cd "$(dirname "$0")/.." && pwd)"
src="$here/stenos_amd/csrc"
mkdir -p "$here/stenos_amd/lib/exp"
flags="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off"
obj="$here/stenos_amd/lib/exp/decode_kernels_$name.o"
hipcc $flags "$@" -mllvm -structurizecfg-skip-uniform-regions=1 -c "$src/decode_kernels.hip" -o "$obj" 2>/dev/null
hipcc $flags ${ENCODE_FLAGS--DWV_PREDICATE_BRANCHES} "$@" -shared -Wl,-Bsymbolic \
  "$obj" "$src/kernels.hip" "$src/kernels_wide.hip" "$src/byte_kernels.hip" "$src/walk_kernels.hip" "$src/capi.cpp" "$src/strategy.cpp" -o "$here/stenos_amd/lib/exp/libstenos_$name.so" -ldl 2>/dev/null
rm -f "$obj"
echo "built libstenos_$name.so"
