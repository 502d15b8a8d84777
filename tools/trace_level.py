#!/usr/bin/env python3
"""Host-phase trace (STENOS_HOST_TRACE) of one strategy-level compress + decompress.  usage: [level] [GiB] [kind] [T]"""
import os
import sys
import time

os.environ["STENOS_HOST_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from stenos_amd.api import Stenos  # noqa: E402
from stenos_amd.datagen import generate, generate_torch  # noqa: E402

level = int(sys.argv[1]) if len(sys.argv) > 1 else 2
gib = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
kind = sys.argv[3] if len(sys.argv) > 3 else "sine"
T = int(sys.argv[4]) if len(sys.argv) > 4 else 8
n = int(gib * (1 << 30)) // T
src = torch.from_numpy(generate(kind, T, n, 9)).cuda() if kind == "smooth8" else generate_torch(kind, T, n, 42)
st = Stenos(level=level)
dst = torch.empty(st.bound(src.numel()), dtype=torch.uint8, device="cuda")
back = torch.empty_like(src)
for rep in range(2):
    print(f"--- compress (rep {rep})", file=sys.stderr, flush=True)
    t = time.perf_counter()
    c = st.compress(src, T, dst)
    te = time.perf_counter() - t
    print(f"--- decompress (rep {rep})", file=sys.stderr, flush=True)
    t = time.perf_counter()
    st.decompress(dst, T, c, back)
    td = time.perf_counter() - t
    print(f"rep {rep}: compress {te * 1e3:.1f} ms ({src.numel() / te / 1e9:.2f} GB/s), decompress {td * 1e3:.1f} ms ({src.numel() / td / 1e9:.2f} GB/s), ratio {src.numel() / c:.4f}", file=sys.stderr, flush=True)
assert torch.equal(back, src)
