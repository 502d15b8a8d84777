#!/bin/bash
# PMC counters of the codec kernels on one data kind of tools/kind_sweep.py (2 GiB, three launches each way), one pass per group.
# usage: bash tools/pmc_kind.sh <kind> <T>   -> gpurun_out/pmc_kind_<kind>_<T>/p{1,2}/.../*_counter_collection.csv
# then, in the build container: python tools/pmc_summary.py <tag> 2147483648 pmc_kind_<kind>_<T> kind_<kind>_<T>
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_kind_$1_$2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export KINDS=$1
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU"
P2="SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS"
i=1
for P in "$P1" "$P2"; do
  rm -rf $OUT/p$i
  timeout -k 10 200 rocprofv3 --pmc $P --output-format csv -d $OUT/p$i -- python3 $R/tools/kind_sweep.py $2 > $OUT/p$i.log 2>&1
  i=$((i+1))
done
