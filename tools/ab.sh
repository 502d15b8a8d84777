#!/bin/bash
# GPU box: kernel times of the variants in stenos_amd/lib/exp, twice round robin (drift shows), for a few workloads.
# usage: bash tools/ab.sh "<kind:T ...>" variant...
work="$1"; shift
for round in 1 2; do
  for v in "$@"; do
    for w in $work; do
      k=${w%%:*}; T=${w##*:}
      STENOS_LIB_PATH=$PWD/stenos_amd/lib/exp/libstenos_$v.so timeout -k 10 120 python tools/time_codec.py 8 3 $k $T 2>&1 | tail -1
    done
  done
done
