#!/usr/bin/env python3
"""Kernel time (HIP events around the dominant encode launch) of one configuration; frames are not checked, so
this also runs experimental builds.  usage: python tools/time_encode.py [GiB] [reps] [kind] [T]"""
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
st = Stenos(1)
st.set_profiling(True)
dst = torch.empty(st.bound(src.numel()), dtype=torch.uint8, device="cuda")
ms = []
for _ in range(reps + 1):
    try:
        st.compress(src, T, dst)
    except Exception as e:  # experimental builds may produce inconsistent frames
        print("error", e)
    ms.append(st.kernel_ms(0))
v = sorted(ms[1:])
print(f"{kind} T={T} {gib} GiB [{os.environ.get('STENOS_LIB_PATH', 'tree')}] kernel_ms min {v[0]:.3f} median {v[len(v) // 2]:.3f} max {v[-1]:.3f}")
