#!/usr/bin/env python3
"""Whole-call throughput of the device API for one bytesoftype (wall clock around compress / decompress, data resident).
usage: python tools/wide_rate.py T [MiB] [kind]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from stenos_amd.api import Stenos  # noqa: E402
from stenos_amd.datagen import generate  # noqa: E402

T = int(sys.argv[1])
mib = int(sys.argv[2]) if len(sys.argv) > 2 else 256
kind = sys.argv[3] if len(sys.argv) > 3 else "mixed"
one = torch.from_numpy(generate(kind, T, max(512, (8 << 20) // T), 7))
src = one.repeat(max(1, (mib << 20) // one.numel())).cuda()
st = Stenos(1)
dst = torch.empty(st.bound(src.numel()), dtype=torch.uint8, device="cuda")
back = torch.empty_like(src)
best_e = best_d = 1e9
for _ in range(3):
    torch.cuda.synchronize()
    t = time.perf_counter()
    r = st.compress(src, T, dst)
    torch.cuda.synchronize()
    best_e = min(best_e, time.perf_counter() - t)
    idx, nsb = st.last_index()
    t = time.perf_counter()
    st.decompress(dst, T, r, back, index_ptr=idx)
    torch.cuda.synchronize()
    best_d = min(best_d, time.perf_counter() - t)
assert torch.equal(back, src)
print(f"T={T} {kind} {src.numel() >> 20} MiB ratio {src.numel() / r:.2f}: encode {src.numel() / best_e / 1e9:.1f} GB/s, decode {src.numel() / best_d / 1e9:.1f} GB/s")
