#!/usr/bin/env python3
"""Kernel times of several library builds inside ONE process, on the same source and destination tensors, round robin.
Separate processes see different physical placements of their buffers, and on this pool that alone moves the fused encoder
by up to 15 % (three clusters of times for one binary): variants are compared here with everything but the code equal.
usage: python tools/ab_inproc.py "<kind:T ...>" <variant|tree> ... [--gib G] [--rounds R] [--reps K]
  variant v: stenos_amd/lib/exp/libstenos_<v>.so; tree: stenos_amd/lib/libstenos.so"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from stenos_amd.api import Stenos, load_library  # noqa: E402
from stenos_amd.datagen import generate, generate_torch  # noqa: E402

args = sys.argv[1:]


def opt(name, default):
    if name in args:
        i = args.index(name)
        v = args[i + 1]
        del args[i:i + 2]
        return type(default)(v)
    return default


gib, rounds, reps = opt("--gib", 8.0), opt("--rounds", 3), opt("--reps", 3)
work, variants = args[0].split(), args[1:]
libs = {}
for v in variants:
    path = os.path.join(ROOT, "stenos_amd", "lib", "libstenos.so") if v == "tree" else os.path.join(ROOT, "stenos_amd", "lib", "exp", f"libstenos_{v}.so")
    libs[v] = load_library(path)
for w in work:
    kind, T = w.split(":")
    T = int(T)
    try:
        src = generate_torch(kind, T, int(gib * (1 << 30)) // T, 42)
    except ValueError:  # (a kind only the host generator has: 256 MiB of it, repeated)
        n = ((256 << 20) // T) // 32768 * 32768
        src = torch.from_numpy(generate(kind, T, n, 7)).cuda().repeat(max(1, int(gib * 4)))
    ctx = {v: Stenos(1, lib=libs[v]) for v in variants}
    for st in ctx.values():
        st.set_profiling(True)
    dst = torch.empty(next(iter(ctx.values())).bound(src.numel()), dtype=torch.uint8, device="cuda")
    back = torch.empty_like(src)
    enc = {v: [] for v in variants}
    dec = {v: [] for v in variants}
    ok = {v: True for v in variants}
    for r in range(rounds + 1):
        for v in variants:
            st = ctx[v]
            for _ in range(reps):
                c = st.compress(src, T, dst)
                e = st.kernel_ms(0)
                idx, _ = st.last_index()
                try:
                    st.decompress(dst, T, c, back, index_ptr=idx)
                    d = st.kernel_ms(1)
                except Exception:  # (an experimental encoder may write frames that do not decode)
                    d = float("nan")
                if r:  # (round 0 warms up)
                    enc[v].append(e)
                    dec[v].append(d)
            if r == rounds:
                ok[v] = bool(torch.equal(back, src))
    for v in variants:
        e, d = sorted(enc[v]), sorted(dec[v])
        print(f"{kind} T={T} {gib} GiB [{v}] encode min {e[0]:.3f} med {e[len(e) // 2]:.3f} max {e[-1]:.3f}  decode min {d[0]:.3f} med {d[len(d) // 2]:.3f} ms  "
              f"roundtrip {'ok' if ok[v] else 'MISMATCH'}", flush=True)
    for st in ctx.values():
        st.close()
    del src, dst, back
    torch.cuda.empty_cache()
