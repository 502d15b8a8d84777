#!/usr/bin/env python3
"""Compare the frame and the superblock index of the fused encode path with the unfused pipeline (diagnostics).
usage: python tools/dbg_fused.py [GiB] [kind] [T]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from stenos_amd.api import Stenos  # noqa: E402
from stenos_amd.datagen import generate_torch  # noqa: E402

gib = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
kind = sys.argv[2] if len(sys.argv) > 2 else "rand12"
T = int(sys.argv[3]) if len(sys.argv) > 3 else 4
src = generate_torch(kind, T, int(gib * (1 << 30)) // T, 42)
st = Stenos(level=1)
out = []
for fused in (1, 0):
    if fused:
        os.environ.pop("STENOS_NO_FUSED", None)
    else:
        os.environ["STENOS_NO_FUSED"] = "1"
    dst = torch.zeros(st.bound(src.numel()), dtype=torch.uint8, device="cuda")
    c = st.compress(src, T, dst)
    idx, nsb = st.last_index()
    index = torch.empty(nsb + 1, dtype=torch.int64, device="cuda")
    ctypes.cdll.LoadLibrary("libamdhip64.so").hipMemcpy(ctypes.c_void_p(index.data_ptr()), ctypes.c_void_p(idx), ctypes.c_size_t((nsb + 1) * 8), 3)
    out.append((c, dst, index.clone(), nsb))
    print("fused" if fused else "plain", "csize", c, "nsb", nsb, flush=True)
(c1, d1, i1, n1), (c0, d0, i0, n0) = out
print("sizes equal", c1 == c0, "index equal", torch.equal(i1, i0))
if not torch.equal(i1, i0):
    bad = (i1 != i0).nonzero()[0].item()
    print("first index mismatch at superblock", bad, i1[max(0, bad - 2):bad + 3].tolist(), i0[max(0, bad - 2):bad + 3].tolist())
n = min(c1, c0)
neq = (d1[:n] != d0[:n])
if neq.any():
    pos = neq.nonzero()[0].item()
    print("first byte mismatch at", pos, "of", n, d1[pos:pos + 16].tolist(), d0[pos:pos + 16].tolist())
    sb = int((i0 <= pos).sum().item()) - 1
    print("in superblock", sb, "offset in superblock", pos - int(i0[sb].item()))
else:
    print("frames equal")
