#!/bin/bash
# The divergent branches of a kernel, mapped to source lines (with the chain of inlined calls): the device code of one
# .hip file is compiled to LLVM IR with line tables, LLVM's uniformity analysis is printed for it, and every terminator it
# marks DIVERGENT is looked up in the IR's debug locations.  One such branch inside a loop makes the compiler rebuild the
# control flow of the whole loop around lane masks; a kernel that prints nothing here can be compiled with
# -mllvm -structurizecfg-skip-uniform-regions.
# usage: tools/divergent_branches.sh <file.hip> <kernel name fragment>... [-- extra compiler flags]
set -e
here="$(cd "$(dirname "$0")/.." && pwd)"
src="$1"; shift
keys=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do keys+=("$1"); shift; done
[ "$1" == "--" ] && shift
mkdir -p "$here/build"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -gline-tables-only -S -emit-llvm --cuda-device-only "$@" \
  -o "$here/build/uniformity.ll" "$here/stenos_amd/csrc/$src" 2>/dev/null
/opt/rocm/lib/llvm/bin/opt -mtriple=amdgcn-amd-amdhsa -mcpu=gfx950 -passes='print<uniformity>' -disable-output "$here/build/uniformity.ll" 2> "$here/build/uniformity.txt"
for k in "${keys[@]}"; do
  echo "== $k"
  python3 "$here/tools/divergent_branches.py" "$here/build/uniformity.ll" "$here/build/uniformity.txt" "$k" | sort | uniq -c | sort -rn
done
rm -f "$here/build/uniformity.ll" "$here/build/uniformity.txt"  # (build/ travels to the GPU box: no listings left behind)
