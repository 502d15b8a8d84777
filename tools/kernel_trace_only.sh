R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/kt
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kt -- python3 $R/bench.py --no-cpu-baseline --no-full-entropy --no-other-configs > $R/gpurun_out/kt.log 2>&1 || exit 1
find $R/gpurun_out/kt -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/r05_v3_kernel_stats_raw.csv \;
grep "encode_superblocks" $R/gpurun_out/r05_v3_kernel_stats_raw.csv | cut -d, -f9-12
grep -o '"kernel_ms": [0-9.]*' $R/gpurun_out/kt.log | head -2
