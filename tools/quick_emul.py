#!/usr/bin/env python3
"""Fast parity check of the slot encoders (bytesoftype 2, 4 and 8; encode_run, as a wave of the fused kernel runs it) in the host emulation against the oracle: the block streams of
many kinds and sizes, byte for byte.  For iterating on slot_codec.h; the full matrix is tests/test_emulation_vs_oracle.py.
usage: python tools/quick_emul.py [T ...]"""
import ctypes, os, subprocess, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from _libs import ROOT, load_oracle, np_ptr
from stenos_amd.datagen import generate

d = os.path.join(ROOT, "tests", "emul")
subprocess.check_call(["make", "-C", d], stdout=subprocess.DEVNULL)
emul = ctypes.CDLL(os.path.join(d, "libstenos_emul_enc.so"))
emul.emul_run_compress.restype = ctypes.c_size_t
emul.emul_run_compress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p]
oracle = load_oracle()
KINDS = ["rand", "same", "sorted", "walk", "ramp", "dict16", "runs", "burst", "mixed", "lzmix", "rand8", "sine"]
t0 = time.time()
cases = 0
for T in [int(x) for x in sys.argv[1:]] or [4, 2, 8]:
    for kind in [k for k in KINDS if k != "sine" or T in (4, 8)] + (["rand12", "sorted_i32"] if T == 4 else []):
        for n in (256, 512, 768, 1024, 1280, 4096, 19968, 33 * 256):  # (whole blocks: a run of the fused kernel)
            for seed in (1, 2):
                data = generate(kind, T, n, seed * 1000 + n)
                nb = data.nbytes
                ref = np.zeros(nb * 2 + 4096, dtype=np.uint8)
                r1 = oracle.so_block_compress(np_ptr(data), T, nb, np_ptr(ref), ref.nbytes)
                out = np.zeros(nb * 2 + 4096, dtype=np.uint8)
                r2 = emul.emul_run_compress(np_ptr(data), T, n // 256, np_ptr(out))
                assert r1 == r2 and np.array_equal(ref[:r1], out[:r1]), (T, kind, n, seed, r1, r2)
                cases += 1
print(f"quick_emul: {cases} cases identical to the oracle in {time.time() - t0:.1f} s")
