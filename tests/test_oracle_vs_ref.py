"""The C oracle against the compiled reference itself (build container only; skipped on the GPU box).
Mirrors the matrix of the reference's own round-trip test (tests/tests_comp_decomp.cpp:182-211):
bytesoftype 2..15, same/sorted/random, odd sizes, shrinking dst_size."""
import numpy as np
import pytest

from _libs import has_error, np_ptr, oracle_compress, ref_compress
from stenos_amd.datagen import generate, splitmix64

KINDS = ["rand", "same", "sorted", "walk", "dict16", "runs", "burst", "ramp"]


@pytest.mark.parametrize("T", list(range(2, 19)))
def test_byte_identical_to_reference(oracle, ref_det, T):
    # T >= 19 is excluded: the reference overflows its own partial-block scratch there
    # (block_compress.h:321-330 vs :967-968), see DESIGN.md "reference findings".
    sizes = [0, 1, 15, 16, 17, 100, 255, 256, 257, 511, 1280, 4099, 131072 // T, 131072 // T + 3, 70000]
    for kind in KINDS:
        for n in sizes:
            data = generate(kind, T, n, 1234 + n)
            for level in (0, 1):
                r1, f1 = ref_compress(ref_det, data, T, level)
                r2, f2 = oracle_compress(oracle, data, T, level)
                assert r1 == r2, (kind, n, level)
                assert np.array_equal(f1, f2), (kind, n, level)
                out = np.zeros(data.nbytes + 8, dtype=np.uint8)
                r3 = oracle.so_decompress(np_ptr(f1), T, r1, np_ptr(out), data.nbytes, 1)
                assert r3 == data.nbytes and np.array_equal(out[: data.nbytes], data)


@pytest.mark.parametrize("T,kind,n", [(4, "rand", 300), (4, "walk", 5000), (2, "burst", 70001), (8, "dict16", 3000),
                                      (4, "rand", 33000), (3, "runs", 999), (4, "rand12", 32768 + 200)])
def test_shrinking_dst_matches_reference(oracle, ref_det, T, kind, n):
    """Same result (same bytes or an error) as the reference for every dst_size from bound to 0
    (tests_comp_decomp.cpp:163-177)."""
    data = generate(kind, T, n, 5)
    bound = oracle.so_bound(data.nbytes)
    assert bound == ref_det.stenos_bound(data.nbytes)
    steps = splitmix64(99, 64) % np.uint64(max(10, data.nbytes // 10))
    dst_size, k = bound, 0
    while True:
        r1, f1 = ref_compress(ref_det, data, T, 1, dst_size)
        r2, f2 = oracle_compress(oracle, data, T, 1, dst_size)
        assert has_error(r1) == has_error(r2), (dst_size, r1, r2)
        if not has_error(r1):
            assert r1 == r2 and np.array_equal(f1, f2), dst_size
        if dst_size == 0:
            break
        dst_size = max(0, dst_size - int(steps[k % 64]))
        k += 1


def test_exact_multiple_bug_is_reproduced_and_fixed(oracle, ref_det):
    """SURVEY finding 1: the reference decoder rejects frames whose size is an exact multiple of
    the superblock size (stenos.cpp:1115-1116, 1131)."""
    data = generate("walk", 4, 2 * 32768, 3)
    r, frame = ref_compress(ref_det, data, 4, 1)
    out = np.zeros(data.nbytes, dtype=np.uint8)
    rr = ref_det.stenos_decompress(np_ptr(frame), 4, r, np_ptr(out), data.nbytes)
    assert has_error(rr)
    assert has_error(oracle.so_decompress(np_ptr(frame), 4, r, np_ptr(out), data.nbytes, 0))
    assert oracle.so_decompress(np_ptr(frame), 4, r, np_ptr(out), data.nbytes, 1) == data.nbytes
    assert np.array_equal(out, data)
