#!/usr/bin/env python3
"""Kernel time of decode_superblocks for one configuration.  usage: python tools/time_decode.py [GiB] [reps] [kind] [T]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from stenos_amd.api import Stenos  # noqa: E402
from stenos_amd.datagen import generate_torch  # noqa: E402

gib = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
kind = sys.argv[3] if len(sys.argv) > 3 else "rand12"
T = int(sys.argv[4]) if len(sys.argv) > 4 else 4
src = generate_torch(kind, T, int(gib * (1 << 30)) // T, 42)
enc = Stenos(1)
dst = torch.empty(enc.bound(src.numel()), dtype=torch.uint8, device="cuda")
c = enc.compress(src, T, dst)
idx, nsb = enc.last_index()
st = Stenos(1)
st.set_profiling(True)
back = torch.empty_like(src)
ms = []
for _ in range(reps + 1):
    st.decompress(dst, T, c, back, index_ptr=idx)
    ms.append(st.kernel_ms(1))
ok = bool(torch.equal(back, src))
v = sorted(ms[1:])
print(f"decode {kind} T={T} {gib} GiB [{os.environ.get('STENOS_LIB_PATH', 'tree')}] kernel_ms min {v[0]:.3f} median {v[len(v) // 2]:.3f} max {v[-1]:.3f} roundtrip_ok {ok}")
