import sys, os, time, torch
sys.path.insert(0, "/root/repo")
from stenos_amd.api import Stenos
from stenos_amd.datagen import generate, generate_torch
for kind, T, level in (("sine", 8, 2), ("smooth8", 1, 3)):
    n = (2 << 30) // T
    src = torch.from_numpy(generate(kind, T, n, 9)).cuda() if kind == "smooth8" else generate_torch(kind, T, n, 42)
    st = Stenos(level=level)
    dst = torch.empty(st.bound(src.numel()), dtype=torch.uint8, device="cuda")
    st.compress(src, T, dst)
    print("----", kind, T, level, file=sys.stderr, flush=True)
    t = time.perf_counter(); c = st.compress(src, T, dst); print("compress s", time.perf_counter() - t, "ratio", src.numel() / c, file=sys.stderr, flush=True)
    st.close()
