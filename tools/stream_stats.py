#!/usr/bin/env python3
"""Where the wavefronts of encode_stream spend their cycles (needs a -DSTENOS_EXP_STATS build).  usage: [GiB] [kind] [T]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from stenos_amd.api import Stenos  # noqa: E402
from stenos_amd.datagen import generate_torch  # noqa: E402

gib = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
kind = sys.argv[2] if len(sys.argv) > 2 else "rand12"
T = int(sys.argv[3]) if len(sys.argv) > 3 else 4
src = generate_torch(kind, T, int(gib * (1 << 30)) // T, 42)
st = Stenos(1)
st.set_profiling(True)
dst = torch.empty(st.bound(src.numel()), dtype=torch.uint8, device="cuda")
hip = ctypes.cdll.LoadLibrary("libamdhip64.so")
for _ in range(3):
    st.compress(src, T, dst)
    idx, nsb = st.last_index()
    word = torch.zeros(16, dtype=torch.int64, device="cuda")
    hip.hipMemcpy(ctypes.c_void_p(word.data_ptr()), ctypes.c_void_p(idx + (nsb - 4) * 8), ctypes.c_size_t(128), 3)
    w = [int(x) for x in word.tolist()]
    print("raw", w)
    print(f"kernel_ms {st.kernel_ms(0):.3f} nsb {nsb}")
