#!/bin/bash
# GPU box: a fast parity subset, then encode kernel times of the three level-1 workloads, for the tree's library and,
# when build/libstenos_base.so exists, for that baseline build (A/B in the same call)
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2 || exit 1
for cfg in "8 9 rand12 4" "4 9 walk 2" "4 9 sine 8"; do
  [ -f build/libstenos_base.so ] && { STENOS_LIB_PATH=$PWD/build/libstenos_base.so timeout -k 10 120 python tools/time_encode.py $cfg 2>/dev/null | grep kernel_ms || exit 1; }
  timeout -k 10 120 python tools/time_encode.py $cfg 2>/dev/null | grep kernel_ms || exit 1
done
for cfg in "8 9 rand12 4" "4 9 walk 2"; do
  [ -f build/libstenos_base.so ] && { STENOS_LIB_PATH=$PWD/build/libstenos_base.so timeout -k 10 120 python tools/time_decode.py $cfg 2>/dev/null | grep kernel_ms || exit 1; }
  timeout -k 10 120 python tools/time_decode.py $cfg 2>/dev/null | grep kernel_ms || exit 1
done
