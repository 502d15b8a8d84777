#!/bin/bash
# GPU box: instruction counters of the encoder for library variants (stenos_amd/lib/exp/libstenos_<v>.so), one compression of 1 GiB.
# usage: bash tools/pmc_ab.sh <kind> <T> variant...   -> gpurun_out/pmcab_<v>_<kind>/... and a summary line per variant
R=$GRAFT_REPO_ROOT
kind=$1; T=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU"
for v in "$@"; do
  out=$R/gpurun_out/pmcab_${v}_${kind}$T
  rm -rf $out
  STENOS_LIB_PATH=$R/stenos_amd/lib/exp/libstenos_$v.so timeout -k 10 150 rocprofv3 --pmc $P1 --output-format csv -d $out -- python3 $R/tools/one_encode.py $kind $T $PMC_DECODE > $out.log 2>&1
  python3 - "$out" "$v" "$kind" "$T" <<'PY'
import csv, glob, sys, collections
out, v, kind, T = sys.argv[1:5]
import os
KERNEL = os.environ.get("PMC_KERNEL", "encode_superblocks")
agg = collections.defaultdict(float)
for p in glob.glob(out + "/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(p)):
        if KERNEL in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
blocks = (1 << 30) / (256 * int(T))
if agg:
    w = agg["SQ_WAVE_CYCLES"]
    print(f"[{v}] {kind} T={T}: per block VALU {agg['SQ_INSTS_VALU']/blocks:.1f} SALU {agg['SQ_INSTS_SALU']/blocks:.1f}  wave-quads {w/blocks:.0f}: active {agg['SQ_ACTIVE_INST_ANY']/w:.3f} wait_inst {agg['SQ_WAIT_INST_ANY']/w:.3f} wait_any {agg['SQ_WAIT_ANY']/w:.3f} busy_cycles {agg['SQ_BUSY_CYCLES']:.0f}")
else:
    print(f"[{v}] no counters")
PY
done
