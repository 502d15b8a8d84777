#!/usr/bin/env python3
"""Copy one checkpoint's measurements from gpurun_out/ (tools/profile_round.sh <tag>) into profiles/:
<tag>_bench.json (the bench line), <tag>_kernel_stats.csv (this repo's kernels and the runtime's copy/fill kernels out of
rocprofv3 --kernel-trace --stats); tools/pmc_summary.py <tag> writes <tag>_pmc_counters.json."""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
line = [l for l in open(os.path.join(ROOT, "gpurun_out", f"{tag}_bench.log")) if l.startswith("{")][-1]
with open(os.path.join(ROOT, "profiles", f"{tag}_bench.json"), "w") as f:
    json.dump(json.loads(line), f, indent=1)
rows = list(csv.reader(open(os.path.join(ROOT, "gpurun_out", f"{tag}_kernel_stats_raw.csv"))))
with open(os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.csv"), "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-full-entropy --no-other-configs  (MI355X, 8 GiB int32 rand12, 2 warmup + 10 timed steps; bench.py runs one more untimed compression for the sharded / parity legs only when asked)\n")
    f.write("# rows of this repo's kernels and the runtime's copy/fill kernels; torch's data-generation kernels are left out\n")
    w = csv.writer(f)
    w.writerow(rows[0])
    for r in rows[1:]:
        if "anonymous namespace" in r[0] and "at::native" not in r[0] or r[0].startswith("__amd_rocclr"):
            w.writerow(r)
print("saved", tag)
