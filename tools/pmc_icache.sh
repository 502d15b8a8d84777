#!/bin/bash
# Instruction-cache counters of the headline kernels in several separate processes (the encoder runs 3.3 or 3.9 ms depending on
# the process: do the misses differ?).  usage: bash tools/pmc_icache.sh [runs]  -> gpurun_out/pmc_icache/r<i>/.../*_counter_collection.csv
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_icache
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for i in $(seq 1 ${1:-4}); do
  rm -rf $OUT/r$i
  timeout -k 10 200 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT/r$i -- python3 $R/bench.py --gib 8 --steps 2 --warmup 1 --no-cpu-baseline --no-full-entropy --no-other-configs > $OUT/r$i.log 2>&1
  grep -o '"kernel_ms": [0-9.]*' $OUT/r$i.log | head -1
done
