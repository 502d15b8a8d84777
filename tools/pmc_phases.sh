#!/bin/bash
# VALU/SALU/LDS instructions per block of the fused encoder with phases switched off (diagnostics; frames are wrong)
# STENOS_DEBUG_PHASES bits: 1 = no LZ attempt, 2 = no emission, 4 = no analysis
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for dbg in 0 1 3 7; do
  rm -rf $R/gpurun_out/pmcph$dbg
  STENOS_DEBUG_PHASES=$dbg rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $R/gpurun_out/pmcph$dbg -- python3 $R/tools/one_encode.py > /dev/null 2>&1
done
