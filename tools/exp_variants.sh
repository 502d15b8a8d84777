#!/bin/bash
# kernel time of the headline encode for experimental builds in stenos_amd/lib/exp (frames may be wrong: timing only)
for v in "$@"; do
  echo -n "$v: "
  STENOS_LIB_PATH=$PWD/stenos_amd/lib/exp/libstenos_$v.so timeout -k 10 120 python tools/time_encode.py 8 3 2>&1 | tail -1
done
