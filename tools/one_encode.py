"""One compression of 1 GiB (for counter collection).  usage: one_encode.py [kind] [T] [decode]"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from stenos_amd.api import Stenos
from stenos_amd.datagen import generate_torch
kind = sys.argv[1] if len(sys.argv) > 1 else "rand12"
T = int(sys.argv[2]) if len(sys.argv) > 2 else 4
n = (1 << 30) // T
src = generate_torch(kind, T, n, 42)
st = Stenos(1)
dst = torch.empty(st.bound(src.numel()), dtype=torch.uint8, device="cuda")
try:
    c = st.compress(src, T, dst)
    if len(sys.argv) > 3:
        back = torch.empty_like(src)
        idx, _ = st.last_index()
        st.decompress(dst, T, c, back, index_ptr=idx)
except Exception as e:
    print("error", e)
