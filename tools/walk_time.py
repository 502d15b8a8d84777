#!/usr/bin/env python3
"""Frame-only decode against indexed decode (wall time of the call, HIP events): what the header walk adds.  usage: walk_time.py [GiB] [kind] [T]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stenos_amd.api import Stenos
from stenos_amd.datagen import generate_torch
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
kind = sys.argv[2] if len(sys.argv) > 2 else "rand12"
T = int(sys.argv[3]) if len(sys.argv) > 3 else 4
src = generate_torch(kind, T, int(gib * (1 << 30)) // T, 42)
st = Stenos(1)
dst = torch.empty(st.bound(src.numel()), dtype=torch.uint8, device="cuda")
back = torch.empty_like(src)
c = st.compress(src, T, dst)
idx, _ = st.last_index()
res = {}
for name, ip in (("indexed", idx), ("frame only", None)):
    best = 1e9
    for _ in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        st.decompress(dst, T, c, back, index_ptr=ip, wait=False)
        e1.record()
        st.finish()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1))
    res[name] = best
print(f"{kind} T={T} {gib} GiB [{os.path.basename(os.environ.get('STENOS_LIB_PATH', 'tree'))}]: indexed {res['indexed']:.3f} ms, frame only {res['frame only']:.3f} ms (+{res['frame only'] - res['indexed']:.3f})  ok {bool(torch.equal(back, src))}")
