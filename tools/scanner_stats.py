#!/usr/bin/env python3
"""Rounds / empty polls of the fused kernel's scanner wave (needs a -DSTENOS_EXP_STATS build).  usage: [GiB]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from stenos_amd.api import Stenos  # noqa: E402
from stenos_amd.datagen import generate_torch  # noqa: E402

gib = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
src = generate_torch("rand12", 4, int(gib * (1 << 30)) // 4, 42)
st = Stenos(1)
st.set_profiling(True)
dst = torch.empty(st.bound(src.numel()), dtype=torch.uint8, device="cuda")
for _ in range(3):
    st.compress(src, 4, dst)
    idx, nsb = st.last_index()
    word = torch.zeros(6, dtype=torch.int64, device="cuda")
    ctypes.cdll.LoadLibrary("libamdhip64.so").hipMemcpy(ctypes.c_void_p(word.data_ptr()), ctypes.c_void_p(idx + (nsb + 1) * 8), ctypes.c_size_t(48), 3)
    v = int(word[0].item())
    cyc = [int(x) / nsb for x in word[1:].tolist()]
    print("cycles per superblock: wait(mid) %.0f store(mid) %.0f wait(last) %.0f store(last) %.0f encode %.0f" % tuple(cyc))
    ms = st.kernel_ms(0)
    rounds, empty = v & 0xFFFFFFFF, (v >> 32) & 0xFFFFFFFF
    print(f"kernel_ms {ms:.3f} nsb {nsb} rounds {rounds} empty polls {empty} -> {1000 * ms / max(1, rounds + empty):.2f} us per poll, {nsb / max(1, rounds):.1f} superblocks per round")
