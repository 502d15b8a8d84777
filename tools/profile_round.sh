#!/bin/bash
# GPU box: everything profiles/ needs for one checkpoint: bench line, kernel-trace statistics of the same command,
# and the PMC passes (HBM traffic, instruction mix).  usage: bash tools/profile_round.sh <tag>
R=$GRAFT_REPO_ROOT
TAG=${1:-r01_vX}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 python3 $R/bench.py > $R/gpurun_out/${TAG}_bench.log 2>&1 || exit 1
rm -rf $R/gpurun_out/kt
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kt -- python3 $R/bench.py --no-cpu-baseline --no-full-entropy --no-other-configs > $R/gpurun_out/kt.log 2>&1 || exit 1
find $R/gpurun_out/kt -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/${TAG}_kernel_stats_raw.csv \;
bash $R/tools/pmc_run.sh 8
