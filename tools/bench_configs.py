#!/usr/bin/env python3
"""Throughput of the other BASELINE.json configs (parity cases, not the headline bench line):
  configs[2] double sine, level 2 (and level 1)     configs[3] int16 random walk, level 1
  configs[4] bytes, level 3                          + int32 sorted (README) level 1
Device-resident timing for level 1 (GPU only); levels >= 2 include the host strategy layer (zstd), which
dominates.  Prints one JSON object per config.  usage: python tools/bench_configs.py [GiB level1] [GiB levels>=2]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from stenos_amd.api import Stenos  # noqa: E402
from stenos_amd.datagen import generate, generate_torch  # noqa: E402

g1 = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
g2 = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5


def run(name, kind, T, level, gib, steps=3):
    n = int(gib * (1 << 30)) // T
    if kind == "smooth8":
        src = torch.from_numpy(generate(kind, T, n, 9)).cuda()
    else:
        src = generate_torch(kind, T, n, 42)
    st = Stenos(level=level)
    st.set_profiling(True)
    dst = torch.empty(st.bound(src.numel()), dtype=torch.uint8, device="cuda")
    back = torch.empty_like(src)
    csize = st.compress(src, T, dst)
    idx, nsb = st.last_index()
    st.decompress(dst, T, csize, back, index_ptr=idx if nsb else None)
    assert torch.equal(back, src)
    torch.cuda.synchronize()
    te = td = 0.0
    for _ in range(steps):
        t = time.perf_counter()
        csize = st.compress(src, T, dst)
        te += time.perf_counter() - t
        idx, nsb = st.last_index()
        t = time.perf_counter()
        st.decompress(dst, T, csize, back, index_ptr=idx if nsb else None)
        td += time.perf_counter() - t
    nb = src.numel()
    out = {"config": name, "bytesoftype": T, "level": level, "GiB": gib, "ratio": round(nb / csize, 4), "encode_gbps": round(nb * steps / te / 1e9, 2),
           "decode_gbps": round(nb * steps / td / 1e9, 2)}
    if level == 1 and T > 1:
        out["encode_blocks_ms"] = round(st.kernel_ms(0), 3)
        out["decode_superblocks_ms"] = round(st.kernel_ms(1), 3)
    print(json.dumps(out), flush=True)
    st.close()


run("README sorted int32", "sorted_i32", 4, 1, min(g1, 4.0))
run("configs[3] int16 random walk (one 8 GiB shard)", "walk", 2, 1, g1)
run("configs[2] double sine, level 1", "sine", 8, 1, g1)
run("configs[2] double sine, level 2", "sine", 8, 2, g2)
run("configs[4] bytes smooth signal, level 3", "smooth8", 1, 3, g2)
