#!/bin/bash
# An experimental build of the library that only recompiles decode_kernels.hip (the decoders): stenos_amd/lib/exp/libstenos_<name>.so.
# The other translation units come from the tree's build (stenos_amd/lib/*.o are not kept, so they are compiled once into /tmp/w/dobjs;
# delete that directory after changing them).
# usage: tools/variant_decode.sh <name> [-DSTENOS_...=.. | -mllvm ... ...]
set -e
name="$1"; shift
here="$(cd "$(dirname "$0")/.." && pwd)"
src="$here/stenos_amd/csrc"
objs=/tmp/w/dobjs
mkdir -p "$here/stenos_amd/lib/exp" $objs
flags="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off"
if [ ! -f $objs/capi.o ]; then
  for f in kernels kernels_wide byte_kernels walk_kernels; do hipcc $flags -DWV_PREDICATE_BRANCHES -c "$src/$f.hip" -o $objs/$f.o 2>/dev/null & done
  for f in capi strategy; do hipcc $flags -DWV_PREDICATE_BRANCHES -c "$src/$f.cpp" -o $objs/$f.o 2>/dev/null & done
  wait
fi
hipcc $flags -mllvm -structurizecfg-skip-uniform-regions=1 "$@" -c "$src/decode_kernels.hip" -o $objs/decode_$name.o 2>/dev/null
hipcc $flags -shared -Wl,-Bsymbolic $objs/decode_$name.o $objs/kernels.o $objs/kernels_wide.o $objs/byte_kernels.o $objs/walk_kernels.o $objs/capi.o $objs/strategy.o -o "$here/stenos_amd/lib/exp/libstenos_$name.so" -ldl 2>/dev/null
echo "built libstenos_$name.so"
