#!/bin/bash
# GPU box: the randomised differential tests (tests/test_gpu_fuzz.py) with fresh seeds, a few minutes each, one summary line per
# run appended to gpurun_out/fuzz_soak.txt.  usage: bash tools/fuzz_soak.sh <runs> [seconds level 1] [seconds all levels]
RUNS=${1:-3}; S1=${2:-150}; S2=${3:-100}
for i in $(seq 1 $RUNS); do
  seed=$((20261004 + 7919 * i + $(date +%s) % 100000))
  STENOS_FUZZ_SEED=$seed STENOS_FUZZ_SECONDS=$S1 STENOS_FUZZ_LEVEL_SECONDS=$S2 timeout -k 10 $((S1 + S2 + 120)) python -m pytest tests/test_gpu_fuzz.py -x -q -s 2>&1 | grep -E "random cases|passed|failed|Error|assert" | tr "\n" " " >> gpurun_out/fuzz_soak.txt
  echo " seed=$seed" >> gpurun_out/fuzz_soak.txt
done
