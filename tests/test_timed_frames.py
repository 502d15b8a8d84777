"""Decode-side fixtures for the time-limited mode: frames the unmodified reference produced under
stenos_set_max_nanoseconds (tests/golden/timed_frames.json, generator make_timed_frames.py).  They are the only frames
that carry [252][raw 256*T] blocks (block_compress.h:1158-1176, decode :1823-1828) next to ordinary blocks, blocks coded
at the lower block levels the clock picked, and a custom superblock size (frame byte 255).
CPU: the oracle and the host emulation of the kernel source decode them.  GPU: the shipped library, through the C ABI
and through the device-pointer entry point."""
import base64
import ctypes
import json
import os
import subprocess
import zlib
from ctypes import c_int, c_size_t, c_void_p

import numpy as np
import pytest

from _libs import ROOT, STAT_COPY_BLOCKS, frame_stats, np_ptr
from stenos_amd.datagen import generate

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "timed_frames.json")) as f:
    CASES = json.load(f)["cases"]


def _frame(e):
    return np.frombuffer(zlib.decompress(base64.b64decode(e["frame_zb64"])), dtype=np.uint8).copy()


def _input(e):
    return generate(e["kind"], e["T"], e["n"], 42).view(np.uint8).ravel()


def _id(e):
    return f"{e['kind']}-T{e['T']}-copy{e['copy_blocks']}"


def test_fixture_has_copy_blocks_between_coded_ones(oracle):
    assert {e["T"] for e in CASES} >= {2, 3, 4, 8}
    for e in CASES:
        st = frame_stats(oracle, _frame(e), e["T"])
        assert int(st[STAT_COPY_BLOCKS]) == e["copy_blocks"] > 0 and e["coded_planes"] > 0 and e["shift_byte"] == 255


@pytest.mark.parametrize("e", CASES, ids=_id)
def test_oracle_decodes_timed_frames(oracle, e):
    frame, data = _frame(e), _input(e)
    out = np.zeros(data.nbytes + 16, dtype=np.uint8)
    assert oracle.so_decompress(np_ptr(frame), e["T"], frame.nbytes, np_ptr(out), data.nbytes, 1) == data.nbytes
    assert np.array_equal(out[: data.nbytes], data)


@pytest.mark.parametrize("e", CASES, ids=_id)
def test_emulation_decodes_timed_superblocks(e):
    """Every BLOCK superblock of the frame through the kernel source's decoder (both the image path and the register path)."""
    d = os.path.join(ROOT, "tests", "emul")
    subprocess.check_call(["make", "-C", d], stdout=subprocess.DEVNULL)
    emul = ctypes.CDLL(os.path.join(d, "libstenos_emul.so"))
    emul.emul_block_decompress.restype = c_size_t
    emul.emul_block_decompress.argtypes = [c_void_p, c_size_t, c_size_t, c_size_t, c_void_p, c_int]
    emul.emul_set_dec_regs.restype = None
    emul.emul_set_dec_regs.argtypes = [c_int]
    frame, data, T = _frame(e), _input(e), e["T"]
    assert frame[0] == 255
    sb = int.from_bytes(frame[8:12].tobytes(), "little")
    p, s, seen = 12, 0, 0
    while p < frame.nbytes:
        code, csize = int(frame[p]), int.from_bytes(frame[p + 1:p + 4].tobytes(), "little")
        want = data[s * sb:(s + 1) * sb]
        if code == 1:
            payload = frame[p + 4:p + 4 + csize].copy()
            for regs in (0, 1):
                emul.emul_set_dec_regs(regs)
                out = np.zeros(want.nbytes + 64, dtype=np.uint8)
                assert emul.emul_block_decompress(np_ptr(payload), csize, T, want.nbytes, np_ptr(out), 0) == want.nbytes, (s, regs)
                assert np.array_equal(out[: want.nbytes], want), (s, regs)
            seen += 1
        p += 4 + csize
        s += 1
    emul.emul_set_dec_regs(1)
    assert seen > 0 and s * sb >= data.nbytes


@pytest.mark.gpu
@pytest.mark.parametrize("e", CASES, ids=_id)
def test_gpu_decodes_timed_frames(e):
    import torch

    from stenos_amd.api import Stenos, load_library

    lib = load_library()
    frame, data = _frame(e), _input(e)
    out = np.full(data.nbytes + 64, 0x5A, dtype=np.uint8)
    r = lib.stenos_decompress(np_ptr(frame), e["T"], frame.nbytes, np_ptr(out), data.nbytes)
    assert r == data.nbytes, hex(r)
    assert np.array_equal(out[: data.nbytes], data)
    assert (out[data.nbytes:] == 0x5A).all()
    # device-resident, at an odd byte address (the register decoder's stores are unaligned then)
    st = Stenos(level=1)
    d_frame = torch.from_numpy(frame).to("cuda:0")
    for mis in (0, 3):
        d_out = torch.zeros(data.nbytes + 16 + mis, dtype=torch.uint8, device="cuda:0")
        assert st.decompress(d_frame, e["T"], frame.nbytes, d_out[mis:mis + data.nbytes]) == data.nbytes
        assert np.array_equal(d_out[mis:mis + data.nbytes].cpu().numpy(), data)
        assert not d_out[mis + data.nbytes:].any().item()
    st.close()
