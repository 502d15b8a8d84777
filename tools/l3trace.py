#!/usr/bin/env python3
"""Decode of a level-3 frame of bytes several times over, wall time and the strategy layer's stage clocks per call.
usage: python tools/l3trace.py [GiB] [lib.so]   (a library built with -DSTENOS_HOST_TRACE prints every mark to stderr)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from stenos_amd.api import Stenos, load_library  # noqa: E402
from stenos_amd.datagen import generate  # noqa: E402

gib = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
lib = load_library(sys.argv[2]) if len(sys.argv) > 2 else None
n = int(gib * (1 << 30))
piece = generate("smooth8", 1, 1 << 28, 9)
src = torch.from_numpy(piece).cuda().repeat(n // (1 << 28))
st = Stenos(level=3, lib=lib) if lib else Stenos(level=3)
dst = torch.empty(st.bound(src.numel()), dtype=torch.uint8, device="cuda")
c = st.compress(src, 1, dst)
back = torch.empty_like(src)
for i in range(7):
    st.stage_ms(reset=True)
    print("---- decode", i, file=sys.stderr, flush=True)
    torch.cuda.synchronize()
    t = time.perf_counter()
    st.decompress(dst, 1, c, back)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    sm = st.stage_ms(reset=True)
    print(f"decode {i}: {dt * 1e3:.1f} ms  {n / dt / 1e9:.1f} GB/s  inflate {sm['inflate']:.1f} device_decode {sm['device_decode']:.1f}", flush=True)
print("roundtrip", bool(torch.equal(back, src)))
