#!/usr/bin/env python3
"""Summarise gpurun_out/pmc (tools/pmc_run.sh) per kernel and write profiles/pmc_traffic.json.
usage: pmc_summary.py <tag> [input bytes] [directory under gpurun_out (default pmc)] [config name]
With a config name (tools/pmc_config.sh: int16, double, float32 ...) the summary goes to profiles/<tag>_pmc_<config>.json and
profiles/pmc_traffic.json is left alone (it belongs to the headline).
HBM bytes per launch = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024: on gfx950 FETCH_SIZE reports half of a
wide coalesced streaming read (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact for 16-byte stores."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
input_bytes = float(sys.argv[2]) if len(sys.argv) > 2 else 8.0 * (1 << 30)  # bytes of the profiled workload
subdir = sys.argv[3] if len(sys.argv) > 3 else "pmc"
config = sys.argv[4] if len(sys.argv) > 4 else None
agg = collections.defaultdict(lambda: collections.defaultdict(list))
full_names = collections.defaultdict(lambda: collections.defaultdict(int))  # base name -> counter -> {full kernel name: rows}
# gpurun merges new outputs into the local directory: keep only the newest collection of every pass
newest = {}
for p in glob.glob(os.path.join(ROOT, "gpurun_out", subdir, "p*/*/*_counter_collection.csv")):
    d = os.path.dirname(p)
    if d not in newest or os.path.getmtime(p) > os.path.getmtime(newest[d]):
        newest[d] = p
for p in sorted(newest.values()):
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"]
        if "anonymous namespace" not in k or "elementwise_kernel_with_index" in k or "at::native" in k:
            continue
        name = k.split("::")[1].split("(")[0].split("<")[0]
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
        full_names[name][(r["Counter_Name"], k)] += 1
out = {}
for name, cs in agg.items():
    first = next(iter(cs))
    launches = max(n for (c, k), n in full_names[name].items() if c == first)
    out[name] = {c: sum(v) / launches for c, v in cs.items()}
    out[name]["launches_averaged"] = launches
res = {"source": "rocprofv3 --pmc (tools/pmc_run.sh), bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-full-entropy", "kernels": out}
# On this pool the per-dispatch counters cover only a fraction of the chip's shader engines (it varies from run
# to run: 7/16, 0.55 ... observed), for every kernel and counter alike.  decode_superblocks stores exactly the
# input size with 16-byte stores (WRITE_SIZE is exact for those), which calibrates the fraction.
scale = 1.0
if "decode_superblocks" in out and out["decode_superblocks"].get("WRITE_SIZE"):
    scale = input_bytes / (out["decode_superblocks"]["WRITE_SIZE"] * 1024)
res["counter_coverage"] = round(1.0 / scale, 4)
for k in ("encode_superblocks", "encode_blocks", "pack_frame", "decode_superblocks"):
    if k in out and "FETCH_SIZE" in out[k] and "WRITE_SIZE" in out[k]:
        res[f"{k}_hbm_bytes_per_launch"] = int(scale * (2 * out[k]["FETCH_SIZE"] * 1024 + out[k]["WRITE_SIZE"] * 1024))
        res[f"{k}_hbm_read_bytes"] = int(scale * 2 * out[k]["FETCH_SIZE"] * 1024)
        res[f"{k}_hbm_write_bytes"] = int(scale * out[k]["WRITE_SIZE"] * 1024)
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
import subprocess  # noqa: E402

try:
    build = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
    if subprocess.check_output(["git", "-C", ROOT, "status", "--porcelain", "--", "stenos_amd/csrc"], text=True).strip():
        build += "+local changes"
except Exception:
    build = "unknown"
name_out = f"{tag}_pmc_{config}.json" if config else f"{tag}_pmc_counters.json"
res["profile"] = name_out
res["build"] = build  # the sources the counters were collected from: bench.py quotes it next to roofline.traffic
if config:
    res["config"] = config
    res["source"] = f"rocprofv3 --pmc (tools/pmc_config.sh {config}), bench.py --config ... --steps 1 --warmup 0, separate passes"
with open(os.path.join(ROOT, "profiles", name_out), "w") as f:
    json.dump(res, f, indent=1)
if not config:
    with open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w") as f:
        json.dump({k: v for k, v in res.items() if k != "kernels"}, f, indent=1)
for name, cs in out.items():
    w = cs.get("SQ_WAVES", 0)
    line = f"{name:20s}"
    if w:
        line += f" waves {w:.0f} valu/wave {cs.get('SQ_INSTS_VALU',0)/w:.0f} salu/wave {cs.get('SQ_INSTS_SALU',0)/w:.0f}"
        wc = cs.get("SQ_WAVE_CYCLES", 1)
        line += f" active {cs.get('SQ_ACTIVE_INST_ANY',0)/wc:.2f} wait_inst {cs.get('SQ_WAIT_INST_ANY',0)/wc:.2f} wait_any {cs.get('SQ_WAIT_ANY',0)/wc:.2f}"
    if "SQ_INSTS_LDS" in cs and w:
        line += f" lds/wave {cs['SQ_INSTS_LDS']/w:.0f} bank_conflict/idx_active {cs.get('SQ_LDS_BANK_CONFLICT',0)/max(cs.get('SQ_LDS_IDX_ACTIVE',1),1):.2f}"
    if "FETCH_SIZE" in cs:
        line += f" fetchKB {cs['FETCH_SIZE']:.0f} writeKB {cs.get('WRITE_SIZE',0):.0f}"
    print(line)
