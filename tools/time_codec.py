#!/usr/bin/env python3
"""Kernel times (HIP events around the dominant encode and decode launches) of one configuration.  The round trip is
checked only when the build is the tree's (experimental builds may produce inconsistent frames).
usage: python tools/time_codec.py [GiB] [reps] [kind] [T] [byte offset of the source and of the decoded copy from a 256-byte boundary]"""
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
off = int(sys.argv[5]) if len(sys.argv) > 5 else 0
src = generate_torch(kind, T, int(gib * (1 << 30)) // T, 42)
if off:
    shifted = torch.empty(src.numel() + 256, dtype=torch.uint8, device="cuda")
    shifted[off:off + src.numel()] = src
    src = shifted[off:off + src.numel()]
st = Stenos(1)
st.set_profiling(True)
dst = torch.empty(st.bound(src.numel()), dtype=torch.uint8, device="cuda")
back = torch.empty(src.numel() + 256, dtype=torch.uint8, device="cuda")[off:off + src.numel()]
enc, dec = [], []
for _ in range(reps + 1):
    r = st.compress(src, T, dst)
    enc.append(st.kernel_ms(0))
    idx, nsb = st.last_index()
    try:
        st.decompress(dst, T, r, back, index_ptr=idx)
        dec.append(st.kernel_ms(1))
    except Exception:  # (an experimental encoder may write frames that do not decode)
        dec.append(float("nan"))
ok = torch.equal(back, src)
e, d = sorted(enc[1:]), sorted(dec[1:])
print(f"{kind} T={T} {gib} GiB +{off} [{os.path.basename(os.environ.get('STENOS_LIB_PATH', 'tree'))}] encode min {e[0]:.3f} med {e[len(e) // 2]:.3f}  decode min {d[0]:.3f} med {d[len(d) // 2]:.3f} ms  roundtrip {'ok' if ok else 'MISMATCH'}")
