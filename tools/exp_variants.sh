#!/bin/bash
# encode / decode kernel times for builds in stenos_amd/lib/exp, one after the other on the same box (box-to-box variance is
# about 10 %).  usage: tools/exp_variants.sh "<time_codec.py arguments>" variant...
args="$1"; shift
for v in "$@"; do
  STENOS_LIB_PATH=$PWD/stenos_amd/lib/exp/libstenos_$v.so timeout -k 10 180 python tools/time_codec.py $args 2>&1 | tail -1
done
