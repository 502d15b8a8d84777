#!/bin/bash
# Collect PMC counters for the bench workload (separate passes, no tracing domains mixed in).
# usage (on the GPU box): bash tools/pmc_run.sh [GiB]   -> gpurun_out/pmc/p*/.../*_counter_collection.csv
R=$GRAFT_REPO_ROOT
G=${1:-8}
mkdir -p $R/gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU"
P2="SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS"
i=1
for P in "$P1" "$P2" "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf $R/gpurun_out/pmc/p$i
  timeout -k 10 200 rocprofv3 --pmc $P --output-format csv -d $R/gpurun_out/pmc/p$i -- python3 $R/bench.py --gib $G --steps 1 --warmup 0 --no-cpu-baseline --no-full-entropy --no-other-configs > $R/gpurun_out/pmc/p$i.log 2>&1
  i=$((i+1))
done
