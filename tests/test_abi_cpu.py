"""CPU-side checks of the product library: it loads, exports every symbol include/*.h declares, and
the host logic that needs no GPU (bounds, parameter validation, level 0 framing, get_info) matches the
oracle / the reference.  No codec call is made here (there is no GPU in this environment)."""
import os
import re

import numpy as np
import pytest

from _libs import ROOT, has_error, np_ptr, oracle_compress
from stenos_amd.api import ERR_BASE, load_library
from stenos_amd.datagen import generate

E = lambda k: (1 << 64) - k  # noqa: E731


@pytest.fixture(scope="module")
def lib():
    so = os.path.join(ROOT, "stenos_amd", "lib", "libstenos.so")
    if not os.path.exists(so):
        import __graft_entry__

        __graft_entry__.build()
    return load_library()


def _declared(text):
    return set(re.findall(r"STENOS_EXPORT[^;(]*?\b(stenos_\w+)\s*\(", text))


def _exported(path):
    import subprocess

    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    return {line.split()[-1] for line in out.splitlines() if line.strip()}


def test_every_declared_symbol_is_exported_and_nothing_else(lib):
    """libstenos.so exports exactly what include/*.h declares outside the test-only block: every name is stenos_*, none of the
    C++ runtime's weak template instances or kernel host stubs leaks (csrc/libstenos.map), and the test suite's switches
    (stenos_hip.h, #ifdef STENOS_TEST_HOOKS) are neither declared for a normal build nor exported."""
    declared, hooks = set(), set()
    for h in ("stenos.h", "stenos_hip.h"):
        text = open(os.path.join(ROOT, "include", h)).read()
        m = re.search(r"#ifdef STENOS_TEST_HOOKS(.*?)#endif", text, re.S)
        if m:
            hooks |= _declared(m.group(1))
            text = text.replace(m.group(0), "")
        declared |= _declared(text)
    assert len(declared) >= 30 and hooks == {"stenos_hip_test_lanes", "stenos_hip_test_walk", "stenos_hip_test_fused_timeouts"}
    for name in sorted(declared):
        assert hasattr(lib, name), name
    exported = _exported(os.path.join(ROOT, "stenos_amd", "lib", "libstenos.so"))
    assert exported == declared, (sorted(exported - declared)[:10], sorted(declared - exported)[:10])


def test_the_test_build_has_the_switches():
    """tests/hooks/libstenos_hooks.so: the same sources with -DSTENOS_TEST_HOOKS -- the product's exports plus the three switches."""
    from _libs import load_hooks_library

    load_hooks_library()
    exported = _exported(os.path.join(ROOT, "tests", "hooks", "libstenos_hooks.so"))
    product = _exported(os.path.join(ROOT, "stenos_amd", "lib", "libstenos.so"))
    assert exported - product == {"stenos_hip_test_lanes", "stenos_hip_test_walk", "stenos_hip_test_fused_timeouts"} and product <= exported


def test_bound_matches_reference_formula(lib, oracle):
    for n in (0, 1, 127, 65791, 65792, 65793, 131072, 10**6, 2**33 + 5):
        assert lib.stenos_bound(n) == oracle.so_bound(n)
    assert lib.stenos_has_error(E(6)) == 1 and lib.stenos_has_error(12345) == 0


def test_parameter_validation(lib):
    ctx = lib.stenos_make_context()
    buf = np.zeros(4096, dtype=np.uint8)
    out = np.zeros(8192, dtype=np.uint8)
    # bytesoftype 0 / >= 65535 (stenos.cpp:119-120)
    assert lib.stenos_compress_generic(ctx, np_ptr(buf), 0, 4096, np_ptr(out), 8192) == E(7)
    assert lib.stenos_compress_generic(ctx, np_ptr(buf), 65535, 4096, np_ptr(out), 8192) == E(7)
    assert lib.stenos_decompress_generic(ctx, np_ptr(buf), 0, 4096, np_ptr(out), 8192) == E(7)
    # block shift >= 16 rejected (stenos.cpp:276-286)
    assert lib.stenos_set_block_size(ctx, 16) == E(9)
    assert lib.stenos_set_block_size(ctx, 3) == 0
    assert lib.stenos_set_block_size(ctx, (1 << 64) - 1) == 0
    # dst too small for the frame header (stenos.cpp:862-863)
    assert lib.stenos_compress_generic(ctx, np_ptr(buf), 4, 4096, np_ptr(out), 7) == E(6)
    # decoder: truncated header, bad shift byte, dst too small (stenos.cpp:1078-1090)
    assert lib.stenos_decompress_generic(ctx, np_ptr(buf), 4, 7, np_ptr(out), 8192) == E(2)
    bad = np.array([9, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0], dtype=np.uint8)
    assert lib.stenos_decompress_generic(ctx, np_ptr(bad), 4, 12, np_ptr(out), 8192) == E(4)
    big = np.array([0, 0, 0, 1, 0, 0, 0, 0, 6, 0, 0, 0], dtype=np.uint8)
    assert lib.stenos_decompress_generic(ctx, np_ptr(big), 4, 12, np_ptr(out), 8192) == E(6)
    lib.stenos_destroy_context(ctx)


@pytest.mark.parametrize("T,n", [(4, 0), (4, 1), (4, 1000), (2, 70000), (8, 33000), (3, 999), (17, 5000)])
def test_level0_frames_match_oracle_and_roundtrip(lib, oracle, T, n):
    """Level 0 is framing + memcpy (stenos.cpp:431-433): done on the host, byte-identical to the oracle."""
    data = generate("rand", T, n, 11)
    r0, ref = oracle_compress(oracle, data, T, 0)
    out = np.zeros(lib.stenos_bound(data.nbytes) + 8, dtype=np.uint8)
    r = lib.stenos_compress(np_ptr(data), T, data.nbytes, np_ptr(out), out.nbytes - 8, 0)
    assert r == r0 and np.array_equal(out[:r], ref)
    back = np.zeros(data.nbytes + 8, dtype=np.uint8)
    assert lib.stenos_decompress(np_ptr(out), T, r, np_ptr(back), data.nbytes) == data.nbytes
    assert np.array_equal(back[: data.nbytes], data)
    # shrinking dst: error, never a write past dst_size (tests_comp_decomp.cpp:103-121)
    if data.nbytes:
        out[:] = 0xAB
        assert has_error(lib.stenos_compress(np_ptr(data), T, data.nbytes, np_ptr(out), r - 1, 0))
        assert (out[r - 1:] == 0xAB).all()


def test_get_info_and_private_header(lib):
    import ctypes

    class Info(ctypes.Structure):
        _fields_ = [("decompressed_size", ctypes.c_size_t), ("superblock_size", ctypes.c_size_t)]

    hdr = np.zeros(16, dtype=np.uint8)
    assert lib.stenos_private_create_compression_header(123456, 4096, np_ptr(hdr), 16) == 12
    info = Info()
    assert lib.stenos_get_info(np_ptr(hdr), 4, 12, ctypes.byref(info)) == 12
    assert (info.decompressed_size, info.superblock_size) == (123456, 4096)
    plain = np.array([1, 0x40, 0x42, 0x0F, 0, 0, 0, 0], dtype=np.uint8)  # shift 1, 1 000 000 bytes
    assert lib.stenos_get_info(np_ptr(plain), 4, 8, ctypes.byref(info)) == 8
    assert (info.decompressed_size, info.superblock_size) == (1000000, 262144)
    assert lib.stenos_private_block_size(np_ptr(np.array([1, 0x10, 0x02, 0x00], dtype=np.uint8)), 4) == 0x210 + 4


def test_no_gpu_means_loud_failure_not_fallback(lib):
    """Without a usable device every codec call must fail with an error code; it must not produce a
    frame by some other route."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    data = generate("walk", 4, 5000, 1)
    out = np.zeros(lib.stenos_bound(data.nbytes), dtype=np.uint8)
    r = lib.stenos_compress(np_ptr(data), 4, data.nbytes, np_ptr(out), out.nbytes, 1)
    assert r >= ERR_BASE
    assert not out.any()


def test_timer(lib):
    t = lib.stenos_make_timer()
    lib.stenos_tick(t)
    a = lib.stenos_tock(t)
    b = lib.stenos_tock(t)
    assert 0 <= a <= b < 10**9
    lib.stenos_destroy_timer(t)
