"""Decode-side fixtures: frames the unmodified reference produced at levels 2..9 and for bytesoftype 1
(tests/golden/level_frames.json, generator make_level_frames.py).  They carry every superblock code 1..6.
CPU: the oracle decodes them.  GPU: the shipped library decodes them through the C ABI (zstd on the host,
unshuffle / delta_inv / block decoder on the device)."""
import base64
import json
import os

import numpy as np
import pytest

from _libs import np_ptr
from stenos_amd.datagen import generate

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "level_frames.json")) as f:
    CASES = json.load(f)["cases"]


def _input(e):
    if "input_b64" in e:
        return np.frombuffer(base64.b64decode(e["input_b64"]), dtype=np.uint8).copy()
    return generate(e["kind"], e["T"], e["n"], 42)


def _id(e):
    return f"{e['kind']}-T{e['T']}-l{e['level']}-codes{''.join(map(str, e['codes']))}"


def test_fixture_covers_all_codes():
    seen = set()
    for e in CASES:
        seen |= set(e["codes"])
    assert seen == {1, 2, 3, 4, 5, 6}


@pytest.mark.parametrize("e", CASES, ids=_id)
def test_oracle_decodes_reference_frames(oracle, e):
    frame = np.frombuffer(base64.b64decode(e["frame_b64"]), dtype=np.uint8).copy()
    data = _input(e)
    out = np.zeros(data.nbytes + 16, dtype=np.uint8)
    assert oracle.so_decompress(np_ptr(frame), e["T"], frame.nbytes, np_ptr(out), data.nbytes, 1) == data.nbytes
    assert np.array_equal(out[: data.nbytes], data)


@pytest.mark.gpu
@pytest.mark.parametrize("e", CASES, ids=_id)
def test_gpu_decodes_reference_frames(e):
    from stenos_amd.api import load_library

    lib = load_library()
    frame = np.frombuffer(base64.b64decode(e["frame_b64"]), dtype=np.uint8).copy()
    data = _input(e)
    out = np.full(data.nbytes + 64, 0x5A, dtype=np.uint8)
    r = lib.stenos_decompress(np_ptr(frame), e["T"], frame.nbytes, np_ptr(out), data.nbytes)
    assert r == data.nbytes, hex(r)
    assert np.array_equal(out[: data.nbytes], data)
    assert (out[data.nbytes:] == 0x5A).all()
