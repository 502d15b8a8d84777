"""The C oracle against the golden vectors generated from the compiled, unmodified reference
(tests/golden/make_golden.py).  This is the pin that makes the oracle trustworthy on the GPU box,
where /root/reference and oracle/_ref may be absent."""
import hashlib
import json
import os

import numpy as np
import pytest

from _libs import (STAT_COUNT, STAT_LZ, STAT_PARTIAL, STAT_PLANE_TYPE, STAT_ROW_HDR, STAT_SB_CODE, frame_stats, has_error,
                   np_ptr, oracle_compress)
from stenos_amd.datagen import generate

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "manifest.json")) as f:
    MANIFEST = json.load(f)["cases"]


def _id(e):
    return f"{e['kind']}-T{e['T']}-n{e['n']}"


@pytest.mark.parametrize("entry", MANIFEST, ids=_id)
def test_oracle_matches_golden(oracle, entry):
    data = generate(entry["kind"], entry["T"], entry["n"], entry["seed"])
    assert data.nbytes == entry["bytes"]
    for level in (0, 1):
        r, frame = oracle_compress(oracle, data, entry["T"], level)
        assert not has_error(r)
        assert r == entry[f"l{level}_size"]
        assert hashlib.sha256(frame.tobytes()).hexdigest() == entry[f"l{level}_sha256"]
    if "l1_hex" in entry:
        assert frame.tobytes().hex() == entry["l1_hex"]
    # decode what the reference produced (== frame) and compare with the input
    out = np.zeros(data.nbytes + 16, dtype=np.uint8)
    r2 = oracle.so_decompress(np_ptr(frame), entry["T"], frame.nbytes, np_ptr(out), data.nbytes, 1)
    assert r2 == data.nbytes
    assert np.array_equal(out[: data.nbytes], data)
    assert not out[data.nbytes:].any()


def test_golden_covers_the_bitstream(oracle):
    """Every plane type, every row header the encoder can emit, LZ blocks, partial blocks and the
    superblock codes 1/2/6 occur in the golden set."""
    total = np.zeros(STAT_COUNT, dtype=np.uint64)
    for e in MANIFEST:
        if e["n"] > 40000 and e["kind"] != "sorted_i32":
            continue
        data = generate(e["kind"], e["T"], e["n"], e["seed"])
        _, frame = oracle_compress(oracle, data, e["T"], 1)
        total += frame_stats(oracle, frame, e["T"])
    assert all(total[STAT_PLANE_TYPE + t] > 0 for t in range(4)), total[:4]
    emitted = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15]
    missing = [h for h in emitted if total[STAT_ROW_HDR + h] == 0]
    assert not missing, f"row headers never produced: {missing}"
    assert total[STAT_LZ] > 0 and total[STAT_PARTIAL] > 0
    assert total[STAT_SB_CODE + 1] > 0 and total[STAT_SB_CODE + 2] > 0 and total[STAT_SB_CODE + 6] > 0


def test_readme_example_known_answer(oracle):
    """README example (reference README.md, SURVEY.md section 8a worked example): 1M sorted int32."""
    data = generate("sorted_i32", 4, 1_000_000, 0)
    r, frame = oracle_compress(oracle, data, 4, 1)
    assert r == 70464
    assert frame[:8].tobytes() == bytes([0, 0x00, 0x09, 0x3D, 0, 0, 0, 0])
    assert frame[8:12].tobytes() == bytes([1, 0x00, 0x09, 0x00])
    assert frame[12:30].tobytes() == bytes.fromhex("03008988888888888888" "fdff01" "feff" "000000")
