#!/bin/bash
# PMC counters for one bench configuration (GPU box): instruction mix, LDS, and the L2-fabric traffic (FETCH_SIZE / WRITE_SIZE),
# one pass per counter group (no tracing domains mixed in).
# usage: bash tools/pmc_config.sh <int32|int16|double> [kind]   -> gpurun_out/pmc_<config>[_kind]/p{1..4}/.../*_counter_collection.csv
# then, in the build container: python tools/pmc_summary.py <tag> <input bytes> pmc_<config>[_kind] <config>[_kind]
R=$GRAFT_REPO_ROOT
CFG=$1
KIND=${2:+--kind $2}
OUT=$R/gpurun_out/pmc_$CFG${2:+_$2}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU"
P2="SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS"
i=1
for P in "$P1" "$P2" "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf $OUT/p$i
  timeout -k 10 200 rocprofv3 --pmc $P --output-format csv -d $OUT/p$i -- python3 $R/bench.py --config $CFG $KIND --gib 8 --steps 1 --warmup 0 --no-cpu-baseline --no-full-entropy --no-other-configs > $OUT/p$i.log 2>&1
  i=$((i+1))
done
