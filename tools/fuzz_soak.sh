#!/bin/bash
# GPU box: the randomised differential tests (tests/test_gpu_fuzz.py) with fresh seeds, a few minutes each, one summary line per
# run appended to gpurun_out/fuzz_soak.txt: the seed, pytest's exit code and its counts -- a run that crashed or did not
# collect shows as rc != 0 / "NO RESULT", never as an empty line that reads like a clean run.  Exits non-zero if any run failed.
# usage: bash tools/fuzz_soak.sh <runs> [seconds level 1] [seconds all levels]
RUNS=${1:-3}; S1=${2:-150}; S2=${3:-100}
mkdir -p gpurun_out
bad=0
for i in $(seq 1 $RUNS); do
  seed=$((20261004 + 7919 * i + $(date +%s) % 100000))
  out=$(STENOS_FUZZ_SEED=$seed STENOS_FUZZ_SECONDS=$S1 STENOS_FUZZ_LEVEL_SECONDS=$S2 timeout -k 10 $((S1 + S2 + 120)) python -m pytest tests/test_gpu_fuzz.py -x -q -s 2>&1)
  rc=$?
  line=$(echo "$out" | grep -E "random cases|passed|failed|Error|assert|fault" | tr "\n" " ")
  if [ $rc -ne 0 ] || ! echo "$line" | grep -q "passed"; then
    bad=1
    line="NO RESULT / FAILED: $line"
  fi
  echo "seed=$seed rc=$rc $line" >> gpurun_out/fuzz_soak.txt
done
exit $bad
