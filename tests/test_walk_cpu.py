"""The parallel walk of the superblock chain (stenos_amd/csrc/walk.h, run by walk_kernels.hip on frames that come without
an index) against the serial walk it replaces (reference stenos.cpp:1126-1134, 1166-1182).  The per-lane functions of the
kernels are replayed on the host (tests/emul/emul.cpp, emul_walk_parallel).  The contract: either the index and the status
equal the serial walk's exactly, or the speculation reports failure (-1) and the serial walk runs -- never a wrong index.
Well-formed frames must not need the fallback."""
import ctypes
import os
import subprocess
from ctypes import c_int, c_uint32, c_uint64, c_void_p

import numpy as np
import pytest

from _libs import ROOT, np_ptr, oracle_compress
from stenos_amd.datagen import generate


@pytest.fixture(scope="module")
def emul():
    d = os.path.join(ROOT, "tests", "emul")
    subprocess.check_call(["make", "-C", d], stdout=subprocess.DEVNULL)
    lib = ctypes.CDLL(os.path.join(d, "libstenos_emul.so"))
    lib.emul_walk_parallel.restype = c_int
    lib.emul_walk_parallel.argtypes = [c_void_p, c_uint64, c_uint64, c_uint64, c_uint32, c_uint64, c_void_p, c_void_p]
    return lib


def serial_walk(frame: np.ndarray, size: int, first: int, nsb: int):
    """walk_superblocks of walk_kernels.hip, statement by statement"""
    off = np.zeros(nsb + 1, dtype=np.uint64)
    status = 0
    p = first
    b = frame
    for s in range(nsb):
        if p + 4 > size:
            off[s:] = size
            return off, 1
        off[s] = p
        p += 4 + (int(b[p + 1]) | (int(b[p + 2]) << 8) | (int(b[p + 3]) << 16))
    off[nsb] = p
    if p > size:
        status = 1
    return off, status


def parallel_walk(emul, frame, size, first, nsb, sb_bytes, seg_len):
    buf = np.concatenate([frame[:size], np.full(64, 0xEE, dtype=np.uint8)])  # (reads past `size` would show)
    off = np.full(nsb + 1, 0xABABABAB, dtype=np.uint64)
    status = np.zeros(1, dtype=np.uint32)
    r = emul.emul_walk_parallel(np_ptr(buf), size, first, nsb, sb_bytes, seg_len, np_ptr(off), np_ptr(status))
    return r, off, int(status[0])


def synthetic_frame(rng, nsb, sb_bytes, sizes, payload):
    """[8-byte header][code][csize:3][payload] ...: what the walk looks at; the payloads are not block streams"""
    parts = [np.zeros(8, dtype=np.uint8)]
    for s in range(nsb):
        if sizes == "uniform":
            c = int(rng.integers(0, sb_bytes + 1))
        elif sizes == "small":
            c = int(rng.integers(0, max(2, sb_bytes // 50)))
        elif sizes == "copies":
            c = sb_bytes
        else:
            c = int(rng.choice([0, 1, sb_bytes // 3, sb_bytes - 1, sb_bytes]))
        h = np.array([int(rng.integers(1, 7)), c & 255, (c >> 8) & 255, (c >> 16) & 255], dtype=np.uint8)
        if payload == "random":
            body = rng.integers(0, 256, c, dtype=np.uint8)
        elif payload == "zeros":
            body = np.zeros(c, dtype=np.uint8)
        elif payload == "headers":  # every fourth byte starts something that looks like a header
            k = int(rng.integers(0, 64))
            body = np.tile(np.array([1, k, 0, 0], dtype=np.uint8), c // 4 + 1)[:c]
        else:  # "nested": a well-formed chain of small superblocks stored inside the payload
            body = synthetic_frame(rng, max(1, c // 64), 120, "uniform", "random")[8 : 8 + c]
            body = np.concatenate([body, np.zeros(c - body.size, dtype=np.uint8)])
        parts += [h, body]
    return np.concatenate(parts)


def check(emul, frame, size, first, nsb, sb_bytes, seg_len, must_succeed):
    want_off, want_status = serial_walk(frame, size, first, nsb)
    r, off, status = parallel_walk(emul, frame, size, first, nsb, sb_bytes, seg_len)
    if r > 0:
        assert np.array_equal(off, want_off), (size, nsb, sb_bytes, seg_len)
        assert status == want_status
    elif must_succeed and r < 0:  # (r == 0: the plan chose the serial walk, a frame too short to cut)
        raise AssertionError(f"fallback (r={r}) on a well-formed frame: size {size} nsb {nsb} sb {sb_bytes} seg_len {seg_len}")
    return r


@pytest.mark.parametrize("sb_bytes", [256, 1000, 4096])
@pytest.mark.parametrize("sizes", ["uniform", "small", "copies", "mixed"])
@pytest.mark.parametrize("payload", ["random", "zeros"])
def test_well_formed_frames_need_no_fallback(emul, sb_bytes, sizes, payload):
    rng = np.random.default_rng(sb_bytes * 7 + len(sizes) + len(payload))
    segmented = 0
    for nsb in (8, 9, 40, 333, 1500):
        frame = synthetic_frame(rng, nsb, sb_bytes, sizes, payload)
        for seg_len in (0, sb_bytes + 4, 2 * (sb_bytes + 4) + 16, 7 * sb_bytes):
            r = check(emul, frame, frame.size, 8, nsb, sb_bytes, seg_len, must_succeed=True)
            segmented += r > 0
    assert segmented >= 4  # (short frames take the serial walk by plan: r == 0)


@pytest.mark.parametrize("payload", ["headers", "nested"])
@pytest.mark.parametrize("sb_bytes", [512, 4096])
def test_payloads_that_look_like_chains(emul, payload, sb_bytes):
    """Payload bytes full of plausible headers, or holding whole chains: the speculation may give up, it must not be wrong."""
    rng = np.random.default_rng(99 + sb_bytes)
    outcomes = []
    for nsb in (16, 100, 700):
        frame = synthetic_frame(rng, nsb, sb_bytes, "uniform", payload)
        for seg_len in (0, sb_bytes + 4, 3 * sb_bytes):
            outcomes.append(check(emul, frame, frame.size, 8, nsb, sb_bytes, seg_len, must_succeed=False))
    assert any(r > 0 for r in outcomes) or payload == "headers"


def test_truncated_and_padded_frames(emul):
    rng = np.random.default_rng(5)
    sb_bytes = 2048
    nsb = 300
    frame = synthetic_frame(rng, nsb, sb_bytes, "uniform", "random")
    ok = 0
    # cut anywhere: inside a payload, inside a header, right behind one
    for size in [frame.size - k for k in (1, 2, 3, 4, 5, 100, 2000, 2047, 2052, 50_000)] + [int(x) for x in rng.integers(20_000, frame.size, 25)]:
        for seg_len in (0, sb_bytes + 4):
            ok += check(emul, frame, size, 8, nsb, sb_bytes, seg_len, must_succeed=False) > 0
    # bytes behind the last superblock; a header that announces fewer or more superblocks than the chain holds
    padded = np.concatenate([frame, rng.integers(0, 256, 30_000, dtype=np.uint8)])
    for n in (nsb, nsb - 1, nsb - 57, nsb + 1, nsb + 40):
        for seg_len in (0, sb_bytes + 4, 5 * sb_bytes):
            check(emul, padded, padded.size, 8, n, sb_bytes, seg_len, must_succeed=False)
            ok += check(emul, frame, frame.size, 8, n, sb_bytes, seg_len, must_succeed=False) > 0
    assert ok > 20  # most of these are chains that simply end early: no reason to fall back


def test_broken_chains(emul):
    """A header with an unknown code or an impossible size in the middle: the serial walk hops on regardless (the decode
    kernel reports the superblock), so the parallel walk must either follow it or step aside."""
    rng = np.random.default_rng(6)
    sb_bytes = 1024
    nsb = 400
    base = synthetic_frame(rng, nsb, sb_bytes, "uniform", "random")
    off, _ = serial_walk(base, base.size, 8, nsb)
    for s in (0, 1, 7, 100, 250, nsb - 2, nsb - 1):
        for kind in ("code0", "code9", "huge"):
            f = base.copy()
            p = int(off[s])
            if kind == "code0":
                f[p] = 0
            elif kind == "code9":
                f[p] = 9
            else:
                f[p + 3] = 0x7F  # csize beyond the frame
            for seg_len in (0, sb_bytes + 4, 4 * sb_bytes):
                check(emul, f, f.size, 8, nsb, sb_bytes, seg_len, must_succeed=False)
    # random bytes: no chain at all
    noise = rng.integers(0, 256, 300_000, dtype=np.uint8)
    for seg_len in (0, sb_bytes + 4):
        check(emul, noise, noise.size, 8, 500, sb_bytes, seg_len, must_succeed=False)


@pytest.mark.parametrize("kind,T,mib", [("rand12", 4, 3), ("rand", 4, 2), ("sorted", 4, 12), ("walk", 2, 3), ("sine", 8, 2)])
def test_frames_of_the_oracle(oracle, emul, kind, T, mib):
    data = generate(kind, T, (mib << 20) // T + 321, 3)
    r, frame = oracle_compress(oracle, data, T, 1)
    sb = oracle.so_superblock_size(T, data.nbytes, 1)
    nsb = (data.nbytes + sb - 1) // sb
    got = [check(emul, frame, r, 8, nsb, sb, seg_len, must_succeed=True) for seg_len in (0, sb + 4, 2 * sb)]
    assert max(got) > 0 or r < 4 * (sb + 4)
