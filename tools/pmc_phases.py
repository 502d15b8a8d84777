#!/usr/bin/env python3
"""Per-block instruction counts from gpurun_out/pmcph* (tools/pmc_phases.sh)."""
import collections
import csv
import glob
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d in (0, 1, 3, 7):
    files = glob.glob(os.path.join(ROOT, f"gpurun_out/pmcph{d}/**/*counter_collection.csv"), recursive=True)
    if not files:
        continue
    newest = max(files, key=os.path.getmtime)
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(newest)):
        agg[r["Kernel_Name"][:48]][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in agg.items():
        if "encode_superblocks" in k:
            blocks = (v["SQ_WAVES"] - 1) * 32  # int32: 32 blocks per wave; minus the scanner wave
            print(f"phases off = {d}:", {c: round(x / blocks, 1) for c, x in v.items() if c != "SQ_WAVES"}, "per block")
