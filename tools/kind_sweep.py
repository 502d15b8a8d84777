#!/usr/bin/env python3
"""Kernel times of every data kind of stenos_amd/datagen.py (256 MiB generated on the host, repeated to 2 GiB on the device; times
scaled to 8 GiB, and in brackets the fraction of the 8 TB/s roofline = (input + frame bytes) / time): which inputs leave the fast paths.  usage: [KINDS="dict16 cycle130"] python tools/kind_sweep.py [T ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from stenos_amd.api import Stenos
from stenos_amd.datagen import generate

KINDS = os.environ.get("KINDS", "rand same sorted walk ramp dict16 runs burst mixed lzmix noise_low steps slopes cycle130 rand8 sine").split()
for T in [int(x) for x in sys.argv[1:]] or [4, 2, 8]:
    for kind in KINDS:
        if kind == "sine" and T == 2:
            continue
        n = ((256 << 20) // T) // 32768 * 32768
        base = torch.from_numpy(generate(kind, T, n, 7)).cuda()
        src = base.repeat(8)
        st = Stenos(1)
        st.set_profiling(True)
        dst = torch.empty(st.bound(src.numel()), dtype=torch.uint8, device="cuda")
        back = torch.empty_like(src)
        e, d = [], []
        for _ in range(3):
            r = st.compress(src, T, dst)
            e.append(st.kernel_ms(0))
            idx, _n = st.last_index()
            st.decompress(dst, T, r, back, index_ptr=idx)
            d.append(st.kernel_ms(1))
        ok = bool(torch.equal(back, src))
        f = 8.0 * (1 << 30) / src.numel()
        # fractions of the 8 TB/s roofline: input + frame bytes over the kernel's time (the algorithmic bytes of both directions)
        algo = src.numel() + r
        fe, fd = algo / (min(e[1:]) * 1e-3) / 8e12, algo / (min(d[1:]) * 1e-3) / 8e12
        print(f"T={T} {kind:10s} ratio {src.numel() / r:7.3f}  encode {min(e[1:]) * f:7.2f} ms ({fe:.3f})  decode {min(d[1:]) * f:7.2f} ms ({fd:.3f}) per 8 GiB  {'ok' if ok else 'MISMATCH'}", flush=True)
        st.close()
        del src, dst, back, base
