#!/bin/bash
# GPU box: address-translation counters of the fused encoder for four contexts in one process (the same build four times:
# stenos_amd/lib/exp/libstenos_<v>{,b,c,d}.so), one launch each -- the per-dispatch rows tell the contexts apart.
# usage: bash tools/pmc_tlb.sh <variant>
R=$GRAFT_REPO_ROOT
v=$1
cd /tmp && export TMPDIR=/tmp
out=$R/gpurun_out/pmc_tlb
rm -rf $out
STENOS_DEBUG_ADDR=1 timeout -k 10 200 rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum --output-format csv -d $out -- python3 $R/tools/ab_inproc.py "rand12:4" $v ${v}b ${v}c ${v}d --rounds 1 --reps 1 > $out.log 2>&1
grep -v "amdgpu.ids\|^ADDR" $out.log | tail -5
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
rows = collections.defaultdict(dict)
for p in glob.glob(out + "/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(p)):
        if "encode_superblocks" in r["Kernel_Name"]:
            rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
for d in sorted(rows):
    print(d, {k: f"{v:.3g}" for k, v in sorted(rows[d].items())})
PY
