"""Levels >= 2 and bytesoftype 1: the strategy layer (LZ4-dry estimator + zstd orchestration,
stenos.cpp:451-604, 617-678).  CPU: the oracle against hashes of frames produced by the compiled reference
(tests/golden/levels_manifest.json; valid with the image's zstd 1.4.9).  GPU: the shipped library (block codec,
shuffle and delta on the device, estimator and zstd on the host) against the oracle and the same hashes."""
import ctypes
import hashlib
import json
import os

import numpy as np
import pytest

from _libs import has_error, np_ptr, oracle_compress
from stenos_amd.datagen import generate

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "levels_manifest.json")) as f:
    CASES = json.load(f)["cases"]


def _zstd_version():
    for name in ("/opt/conda/lib/libzstd.so.1", "libzstd.so.1"):
        try:
            z = ctypes.CDLL(name)
            z.ZSTD_versionNumber.restype = ctypes.c_uint
            return z.ZSTD_versionNumber()
        except OSError:
            continue
    return 0


needs_zstd_149 = pytest.mark.skipif(_zstd_version() != 10409, reason="byte parity of zstd superblocks needs zstd 1.4.9")


def _id(e):
    return f"{e['kind']}-T{e['T']}-n{e['n']}-l{e['level']}"


@needs_zstd_149
@pytest.mark.parametrize("e", CASES, ids=_id)
def test_oracle_matches_reference_hashes(oracle, e):
    data = generate(e["kind"], e["T"], e["n"], e.get("seed", 42))
    r, frame = oracle_compress(oracle, data, e["T"], e["level"])
    assert not has_error(r) and r == e["size"]
    assert hashlib.sha256(frame.tobytes()).hexdigest() == e["sha256"]
    out = np.zeros(data.nbytes + 8, dtype=np.uint8)
    assert oracle.so_decompress(np_ptr(frame), e["T"], r, np_ptr(out), data.nbytes, 1) == data.nbytes
    assert np.array_equal(out[: data.nbytes], data)


@pytest.fixture(scope="module")
def lib():
    from stenos_amd.api import load_library

    return load_library()


@pytest.mark.gpu
@needs_zstd_149
@pytest.mark.parametrize("e", CASES, ids=_id)
def test_gpu_matches_reference_hashes(lib, e):
    data = generate(e["kind"], e["T"], e["n"], e.get("seed", 42))
    out = np.full(lib.stenos_bound(data.nbytes) + 64, 0xA5, dtype=np.uint8)
    r = lib.stenos_compress(np_ptr(data), e["T"], data.nbytes, np_ptr(out), out.nbytes - 64, e["level"])
    assert not has_error(r), hex(r)
    assert r == e["size"]
    assert hashlib.sha256(out[:r].tobytes()).hexdigest() == e["sha256"]
    assert (out[out.nbytes - 64:] == 0xA5).all()
    back = np.zeros(data.nbytes + 8, dtype=np.uint8)
    assert lib.stenos_decompress(np_ptr(out), e["T"], r, np_ptr(back), data.nbytes) == data.nbytes
    assert np.array_equal(back[: data.nbytes], data)


@pytest.mark.gpu
@pytest.mark.parametrize("T,kind,n,level", [(4, "walk", 50_000, 2), (8, "sine", 100_003, 2), (2, "burst", 70_001, 3), (1, "smooth8", 300_000, 3),
                                            (4, "rand", 33_000, 5), (4, "dict16", 40_000, 9)])
def test_gpu_levels_shrinking_dst_equal_oracle(lib, oracle, T, kind, n, level):
    """Same frame or both an error as the oracle's serial path for every dst_size (zstd sees the same capacity)."""
    data = generate(kind, T, n, 5)
    bound = lib.stenos_bound(data.nbytes)
    step = max(10, data.nbytes // 7)
    dst_size = bound
    while True:
        r1, f1 = oracle_compress(oracle, data, T, level, dst_size)
        out = np.full(dst_size + 64, 0xA5, dtype=np.uint8)
        r2 = lib.stenos_compress(np_ptr(data), T, data.nbytes, np_ptr(out), dst_size, level)
        assert (out[dst_size:] == 0xA5).all()
        assert has_error(r1) == has_error(r2), (dst_size, hex(r1), hex(r2))
        if not has_error(r1):
            assert r1 == r2 and np.array_equal(f1, out[:r2]), dst_size
        if dst_size == 0:
            break
        dst_size = max(0, dst_size - step)


@pytest.mark.gpu
def test_device_api_levels(lib):
    import torch

    from stenos_amd.api import Stenos

    for T, kind, n, level in ((8, "sine", 500_001, 2), (1, "smooth8", 1_000_000, 3), (4, "rand12", 700_000, 4)):
        data = generate(kind, T, n, 42)
        st = Stenos(level=level)
        src = torch.from_numpy(data).cuda()
        dst = torch.empty(st.bound(src.numel()), dtype=torch.uint8, device="cuda")
        csize = st.compress(src, T, dst)
        out = np.zeros(st.bound(data.nbytes), dtype=np.uint8)
        r = lib.stenos_compress(np_ptr(data), T, data.nbytes, np_ptr(out), out.nbytes, level)
        assert r == csize and np.array_equal(dst[:csize].cpu().numpy(), out[:r])
        back = torch.zeros_like(src)
        assert st.decompress(dst, T, csize, back) == src.numel()
        assert torch.equal(back, src)
        st.close()


@pytest.mark.gpu
@needs_zstd_149
@pytest.mark.parametrize("T,kind,level", [(4, "walk", 2), (4, "rand12", 3), (8, "sine", 2), (1, "smooth8", 1), (1, "smooth8", 3), (1, "rand", 5), (2, "burst", 4), (4, "dict16", 9)])
def test_private_single_superblock_api_at_all_levels(lib, ref_det, T, kind, level):
    """stenos_private_compress_block / stenos_private_decompress_block as stenos::cvector calls them (cvector.hpp:1397-1416),
    at levels >= 2 and for bytesoftype 1: the superblock is the reference's, and it decodes, for chunk sizes of 256 << shift
    elements with and without a short last chunk."""
    if ref_det is None:
        pytest.skip("oracle/_ref not built")
    c = lib.stenos_make_context()
    rc = ref_det.stenos_make_context()
    lib.stenos_set_level(c, level)
    ref_det.stenos_set_level(rc, level)
    for shift, used in ((0, 256), (2, 1024), (4, 4096), (4, 3000), (6, 16384)):
        sb = (256 << shift) * T
        data = generate(kind, T, used, 11 + shift)
        nb = data.nbytes
        exp = np.zeros(nb + 64, dtype=np.uint8)
        out = np.full(nb + 64 + 64, 0xA5, dtype=np.uint8)
        r1 = ref_det.stenos_private_compress_block(rc, np_ptr(data), T, sb, nb, np_ptr(exp), nb + 64)
        r2 = lib.stenos_private_compress_block(c, np_ptr(data), T, sb, nb, np_ptr(out), nb + 64)
        assert not has_error(r1) and r1 == r2, (shift, used, hex(r1), hex(r2))
        assert np.array_equal(exp[:r1], out[:r2]) and (out[nb + 64:] == 0xA5).all(), (shift, used, int(exp[0]))
        back = np.zeros(nb, dtype=np.uint8)
        assert lib.stenos_private_decompress_block(c, np_ptr(out), T, sb, r2, np_ptr(back), nb) == nb
        assert np.array_equal(back, data), (shift, used)
    lib.stenos_destroy_context(c)
    ref_det.stenos_destroy_context(rc)
