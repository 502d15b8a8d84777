#!/bin/bash
# An experimental build of the library that only recompiles kernels.hip (the encoders): stenos_amd/lib/exp/libstenos_<name>.so.
# The other translation units are compiled once into /tmp/w/objs (delete that directory after changing them).
# usage: tools/variant_fast.sh <name> [-DSTENOS_...=.. | -mllvm ... ...]
set -e
name="$1"; shift
here="$(cd "$(dirname "$0")/.." && pwd)"
src="$here/stenos_amd/csrc"
objs=/tmp/w/objs
mkdir -p "$here/stenos_amd/lib/exp" $objs
flags="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off"
# the cached objects go when one of their sources is newer
for f in "$src"/capi.cpp "$src"/strategy.cpp "$src"/decode_kernels.hip "$src"/kernels_wide.hip "$src"/byte_kernels.hip "$src"/walk_kernels.hip "$src"/*.h "$here"/include/*.h; do
  if [ -f $objs/capi.o ] && [ "$f" -nt $objs/capi.o ] && [ "${f##*/}" != "slot_codec.h" ] && [ "${f##*/}" != "superblock_codec.h" ] && [ -z "$KEEP_OBJS" ]; then rm -f $objs/*.o; fi
done
if [ ! -f $objs/capi.o ]; then
  hipcc $flags -mllvm -structurizecfg-skip-uniform-regions=1 -c "$src/decode_kernels.hip" -o $objs/decode_kernels.o 2>/dev/null &
  for f in kernels_wide byte_kernels walk_kernels; do hipcc $flags -DWV_PREDICATE_BRANCHES -c "$src/$f.hip" -o $objs/$f.o 2>/dev/null & done
  for f in capi strategy; do hipcc $flags -DWV_PREDICATE_BRANCHES -c "$src/$f.cpp" -o $objs/$f.o 2>/dev/null & done
  wait
fi
hipcc $flags ${ENCODE_FLAGS--DWV_PREDICATE_BRANCHES} "$@" -c "$src/kernels.hip" -o $objs/kernels_$name.o 2>/dev/null
hipcc $flags -shared -Wl,-Bsymbolic $objs/decode_kernels.o $objs/kernels_$name.o $objs/kernels_wide.o $objs/byte_kernels.o $objs/walk_kernels.o $objs/capi.o $objs/strategy.o -o "$here/stenos_amd/lib/exp/libstenos_$name.so" -ldl 2>/dev/null
echo "built libstenos_$name.so"
