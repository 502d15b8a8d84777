#!/bin/bash
# Instruction counts per 256-element block of the encoder (and decoder) for build/libstenos_base.so and the tree's library.
# usage (GPU box): bash tools/pmc_ab.sh [kind] [T]
R=$GRAFT_REPO_ROOT
K=${1:-rand12}; T=${2:-4}
cd /tmp && export TMPDIR=/tmp
for lib in base tree; do
  [ $lib = base ] && { [ -f $R/build/libstenos_base.so ] || continue; export STENOS_LIB_PATH=$R/build/libstenos_base.so; } || unset STENOS_LIB_PATH
  rm -rf $R/gpurun_out/pmcab_$lib
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH --output-format csv -d $R/gpurun_out/pmcab_$lib -- python3 $R/tools/one_encode.py $K $T decode > /dev/null 2>&1
  python3 - $R/gpurun_out/pmcab_$lib $lib $T <<'PY'
import csv, glob, sys, collections
d, lib, T = sys.argv[1], sys.argv[2], int(sys.argv[3])
blocks = (1 << 30) // (256 * T)
agg = collections.defaultdict(dict)
for p in glob.glob(d + "/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"]
        for name in ("encode_superblocks", "decode_superblocks"):
            if name in k:
                agg[name][r["Counter_Name"]] = agg[name].get(r["Counter_Name"], 0) + float(r["Counter_Value"])
for name, cs in agg.items():
    print(lib, name, "per block:", " ".join(f"{c[3:]} {v / blocks:.1f}" for c, v in sorted(cs.items()) if c != "SQ_WAVES"))
PY
done
