#!/usr/bin/env python3
"""Wall-clock of one device-resident compress / decompress call against input size (int32, level 1)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from stenos_amd.api import Stenos  # noqa: E402
from stenos_amd.datagen import generate_torch  # noqa: E402

st = Stenos(1)
for mib in (0.125, 1, 16, 128, 1024):
    n = int(mib * (1 << 20)) // 4
    src = generate_torch("rand12", 4, n, 42)
    dst = torch.empty(st.bound(src.numel()), dtype=torch.uint8, device="cuda")
    back = torch.empty_like(src)
    c = st.compress(src, 4, dst)
    idx, nsb = st.last_index()
    st.decompress(dst, 4, c, back, index_ptr=idx)
    torch.cuda.synchronize()
    reps = 20
    t = time.perf_counter()
    for _ in range(reps):
        c = st.compress(src, 4, dst)
    te = (time.perf_counter() - t) / reps
    idx, nsb = st.last_index()
    t = time.perf_counter()
    for _ in range(reps):
        st.decompress(dst, 4, c, back, index_ptr=idx)
    td = (time.perf_counter() - t) / reps
    t = time.perf_counter()
    for _ in range(reps):
        st.decompress(dst, 4, c, back)
    tw = (time.perf_counter() - t) / reps
    print(f"{mib:9.3f} MiB  compress {te * 1e6:9.1f} us ({src.numel() / te / 1e9:8.2f} GB/s)   decompress {td * 1e6:9.1f} us ({src.numel() / td / 1e9:8.2f} GB/s)"
          f"   without index {tw * 1e6:9.1f} us")
