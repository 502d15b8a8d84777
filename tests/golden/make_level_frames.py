#!/usr/bin/env python3
"""Frames produced by the UNMODIFIED reference (oracle/_ref/libstenos_ref_det.so) at levels 2..9 and for
bytesoftype 1: decode-side fixtures (superblock codes 2..5 need the reference's lz4-dry + zstd strategy layer
to be produced, which this repository does not restate yet).  Small inputs only; frames are committed in
tests/golden/level_frames.json (base64), inputs are regenerated from seeds.
Run in the build container:  python tests/golden/make_level_frames.py"""
import base64
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
from _libs import has_error, load_oracle, load_ref, np_ptr, ref_compress  # noqa: E402
from stenos_amd.datagen import generate  # noqa: E402


def text_like(n, seed):
    words = [b"stenos", b"block", b"codec", b"delta", b"plane", b"the", b"of", b"and", b"mi355x", b"wavefront"]
    rng = np.random.default_rng(seed)
    out = b" ".join(words[i] for i in rng.integers(0, len(words), size=n // 5 + 8))
    return np.frombuffer(out[:n], dtype=np.uint8).copy()


CASES = []
for level in (2, 3, 5, 9):
    for kind, T, n in (("sorted_i32", 4, 6000), ("walk", 2, 20000), ("walk", 4, 9000), ("burst", 4, 12000), ("runs", 8, 3000), ("rand12", 4, 9000),
                       ("sine", 8, 5000), ("ramp", 3, 7000), ("dict16", 4, 6000), ("walk", 8, 20000)):
        CASES.append((kind, T, n, level))
CASES += [("same", 2, 20000, 4), ("same", 4, 20000, 5), ("burst", 3, 3000, 9)]  # superblock code 3 (TRANSPOSED_ZSTD)
for level in (1, 3, 9):
    CASES.append(("smooth8", 1, 40000, level))
    CASES.append(("text", 1, 30000, level))
    CASES.append(("rand", 1, 5000, level))


def main():
    ref = load_ref(det=True)
    if ref is None:
        raise SystemExit("oracle/_ref/libstenos_ref_det.so missing: run `make -C oracle ref` first")
    oracle = load_oracle()
    out, codes = [], set()
    for kind, T, n, level in CASES:
        data = text_like(n, 1) if kind == "text" else generate(kind, T, n, 42)
        if (data.nbytes % so_sb(T, data.nbytes, level)) == 0:
            data = data[:-T]
        r, frame = ref_compress(ref, data, T, level)
        assert not has_error(r), (kind, T, n, level, hex(r))
        # the reference must decode its own frame, and so must the oracle
        back = np.zeros(data.nbytes, dtype=np.uint8)
        assert ref.stenos_decompress(np_ptr(frame), T, r, np_ptr(back), back.nbytes) == data.nbytes and np.array_equal(back, data)
        back[:] = 0
        assert oracle.so_decompress(np_ptr(frame), T, r, np_ptr(back), back.nbytes, 1) == data.nbytes and np.array_equal(back, data), (kind, T, level)
        # superblock codes present
        p, shift = 8, int(frame[0])
        cs = []
        while p < r:
            cs.append(int(frame[p]))
            p += 4 + int.from_bytes(frame[p + 1:p + 4].tobytes(), "little")
        codes |= set(cs)
        e = {"kind": kind, "T": T, "n": int(data.nbytes // T), "level": level, "codes": sorted(set(cs)), "frame_b64": base64.b64encode(frame.tobytes()).decode()}
        if kind == "text":
            e["input_b64"] = base64.b64encode(data.tobytes()).decode()
        out.append(e)
    with open(os.path.join(HERE, "level_frames.json"), "w") as f:
        json.dump({"generator": "tests/golden/make_level_frames.py", "cases": out}, f)
    print(len(out), "frames, codes seen:", sorted(codes), "bytes:", sum(len(e["frame_b64"]) * 3 // 4 for e in out))


def so_sb(T, nbytes, level):
    bs = 256 * T
    sb = bs if bs > 131072 else (131072 // bs) * bs
    return sb << ((level - 1) // 2) if nbytes > sb else sb


if __name__ == "__main__":
    main()
