#!/usr/bin/env python3
"""Generate tests/golden/manifest.json (+ a few small compressed frames) from the UNMODIFIED
reference compiled by oracle/Makefile (oracle/_ref/libstenos_ref_det.so: reference sources built
with -ftrivial-auto-var-init=pattern so that the mini-LZ table of block_compress.h:1211 starts
empty and the output is a pure function of the input).

Run in the build container only (needs /root/reference):  python tests/golden/make_golden.py
Inputs are regenerated from seeds by stenos_amd.datagen; only sizes, hashes and a handful of small
compressed frames are committed.
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

from _libs import has_error, load_ref, ref_compress  # noqa: E402
from stenos_amd.datagen import generate  # noqa: E402

# (kind, T, n, seed)
CASES = [("sorted_i32", 4, 1_000_000, 0)]
SIZES = [1, 15, 16, 100, 255, 256, 257, 1280, 10317, 33013]
for T in (2, 3, 4, 5, 7, 8, 12, 16):
    for kind in ("rand", "same", "sorted", "walk", "ramp", "dict16", "runs", "burst"):
        for n in SIZES:
            CASES.append((kind, T, n, 42))
for n in SIZES + [300_000]:
    CASES.append(("rand12", 4, n, 42))
# multi-superblock, exact superblock multiples (the reference ENCODER is fine on those) and +-1 element
for T in (2, 4, 8):
    per_sb = 131072 // T
    for kind in ("walk", "burst", "dict16"):
        for n in (per_sb - 1, per_sb, per_sb + 1, 3 * per_sb, 3 * per_sb + 777):
            CASES.append((kind, T, n, 7))
CASES.append(("walk", 2, 1_000_000, 7))
CASES.append(("rand12", 4, 2_097_152 + 1000, 42))
CASES.append(("rand", 4, 100_000, 42))

SMALL_FRAMES = 4096  # frames up to this many bytes are committed verbatim (hex) for a few cases


def main():
    ref = load_ref(det=True)
    if ref is None:
        raise SystemExit("oracle/_ref/libstenos_ref_det.so missing: run `make -C oracle ref` first")
    stock = load_ref(det=False)
    out = []
    for kind, T, n, seed in CASES:
        data = generate(kind, T, n, seed)
        entry = {"kind": kind, "T": T, "n": n, "seed": seed, "bytes": int(data.nbytes)}
        for level in (0, 1):
            r, frame = ref_compress(ref, data, T, level)
            assert not has_error(r), (kind, T, n, level, r)
            entry[f"l{level}_size"] = int(r)
            entry[f"l{level}_sha256"] = hashlib.sha256(frame.tobytes()).hexdigest()
            if level == 1 and r <= SMALL_FRAMES and n in (100, 257, 1280) and T in (2, 4, 8):
                entry["l1_hex"] = frame.tobytes().hex()
        # does the stock (non pattern-initialised) build agree on this input?
        if stock is not None:
            r2, f2 = ref_compress(stock, data, T, 1)
            entry["stock_equal"] = bool(r2 == entry["l1_size"] and hashlib.sha256(f2.tobytes()).hexdigest() == entry["l1_sha256"])
        out.append(entry)
    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump({"generator": "tests/golden/make_golden.py", "reference": "Thermadiag/stenos v0.2 (unmodified sources, clang -ftrivial-auto-var-init=pattern)", "cases": out}, f, indent=0)
    print(len(out), "cases;", sum(1 for e in out if "l1_hex" in e), "with frames;",
          sum(1 for e in out if not e.get("stock_equal", True)), "where the stock build differs")


if __name__ == "__main__":
    main()
