#!/usr/bin/env python3
"""Kernel times (HIP events around the dominant encode / decode launch) of the level-1 workloads, one line each, in one
process so that they come from the same device.  Round trips are checked.
usage: python tools/kernel_times.py [GiB] [kind:T ...]     default: the BASELINE.json configs at level 1"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from stenos_amd.api import Stenos  # noqa: E402
from stenos_amd.datagen import generate_torch  # noqa: E402

gib = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
work = [w.split(":") for w in sys.argv[2:]] or [["rand12", "4"], ["rand", "4"], ["sorted_i32", "4"], ["rand8", "4"], ["sine", "4"], ["walk", "2"], ["rand8", "2"], ["sine", "8"]]
tag = os.path.basename(os.environ.get("STENOS_LIB_PATH", "tree"))
for kind, T in work:
    T = int(T)
    src = generate_torch(kind, T, int(gib * (1 << 30)) // T, 42)
    st = Stenos(1)
    st.set_profiling(True)
    dst = torch.empty(st.bound(src.numel()), dtype=torch.uint8, device="cuda")
    back = torch.empty_like(src)
    enc, dec = [], []
    for _ in range(4):
        c = st.compress(src, T, dst)
        enc.append(st.kernel_ms(0))
        idx, _ = st.last_index()
        st.decompress(dst, T, c, back, index_ptr=idx)
        dec.append(st.kernel_ms(1))
    ok = bool(torch.equal(back, src))
    n = src.numel()
    e, d = sorted(enc[1:])[1], sorted(dec[1:])[1]
    print(f"[{tag}] {kind:10s} T={T} {gib} GiB ratio {n / c:8.4f} encode {e:7.3f} ms ({(n + c) / e / 8e9 * 100:4.1f}% of 8 TB/s) "
          f"decode {d:7.3f} ms ({(n + c) / d / 8e9 * 100:4.1f}%) roundtrip_ok {ok}", flush=True)
    st.close()
    del src, dst, back
    torch.cuda.empty_cache()
