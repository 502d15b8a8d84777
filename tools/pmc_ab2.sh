#!/bin/bash
# GPU box: two counter passes of the encoder for library variants (stenos_amd/lib/exp/libstenos_<v>.so), one compression of 1 GiB each.
# usage: bash tools/pmc_ab2.sh <kind> <T> variant...   -> one summary line per variant and pass
R=$GRAFT_REPO_ROOT
kind=$1; T=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU"
P2="SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS"
for v in "$@"; do
  for pass in 1 2 3 4; do
    case $pass in 1) P="$P1";; 2) P="$P2";; 3) P="FETCH_SIZE";; 4) P="WRITE_SIZE";; esac
    out=$R/gpurun_out/pmcab2_${v}_${kind}${T}_p$pass
    rm -rf $out
    STENOS_LIB_PATH=$R/stenos_amd/lib/exp/libstenos_$v.so timeout -k 10 150 rocprofv3 --pmc $P --output-format csv -d $out -- python3 $R/tools/one_encode.py $kind $T $PMC_DECODE > $out.log 2>&1
    python3 - "$out" "$v" "$kind" "$T" <<'PY'
import csv, glob, sys, collections, os
out, v, kind, T = sys.argv[1:5]
KERNEL = os.environ.get("PMC_KERNEL", "encode_superblocks")
agg = collections.defaultdict(float)
for p in glob.glob(out + "/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(p)):
        if KERNEL in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
blocks = (1 << 30) / (256 * int(T))
print(f"[{v}] {kind} T={T}: " + "  ".join(f"{k} {val/blocks:.2f}/blk" for k, val in sorted(agg.items())))
PY
  done
done
