#!/bin/bash
# GPU box: like ab.sh with more repetitions per run and three rounds (noisy boxes).  usage: bash tools/ab_long.sh "<kind:T ...>" variant...
work="$1"; shift
for round in 1 2 3; do
  for v in "$@"; do
    for w in $work; do
      k=${w%%:*}; T=${w##*:}
      STENOS_LIB_PATH=$PWD/stenos_amd/lib/exp/libstenos_$v.so timeout -k 10 120 python tools/time_codec.py 8 10 $k $T 2>&1 | tail -1
    done
  done
done
