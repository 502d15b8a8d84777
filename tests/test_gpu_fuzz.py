"""Randomised differential test of the shipped library against the oracle: bytesoftype, data kind, length and destination
capacity drawn at random (seeded).  Runs for STENOS_FUZZ_SECONDS (default 20) seconds."""
import os
import time

import numpy as np
import pytest

from _libs import has_error, np_ptr, oracle_compress
from stenos_amd.api import load_library
from stenos_amd.datagen import generate

pytestmark = pytest.mark.gpu

KINDS = ["rand", "same", "sorted", "walk", "ramp", "dict16", "runs", "burst", "mixed", "lzmix", "noise_low", "steps", "slopes", "cycle130", "edge_noise"]


def test_random_cases_against_oracle(oracle):
    lib = load_library()
    ctx = lib.stenos_make_context()
    rng = np.random.default_rng(int(os.environ.get("STENOS_FUZZ_SEED", "20261003")))
    budget = float(os.environ.get("STENOS_FUZZ_SECONDS", "20"))
    t0, cases = time.time(), 0
    while time.time() - t0 < budget:
        r = rng.random()
        T = int(rng.integers(2, 17)) if r < 0.55 else int(rng.integers(17, 65)) if r < 0.8 else int(rng.integers(65, 600)) if r < 0.97 else int(rng.integers(600, 9000))
        kind = KINDS[int(rng.integers(len(KINDS)))]
        bs = 256 * T
        per = (131072 // bs * 256) if bs <= 131072 else 256
        shape = rng.random()
        n = int(rng.integers(1, 300)) if shape < 0.2 else int(rng.integers(1, 3 * per)) if shape < 0.9 else int(rng.integers(3 * per, 8 * per))
        n = min(n, (48 << 20) // T)
        seed = int(rng.integers(1 << 30))
        data = generate(kind, T, n, seed)
        bound = lib.stenos_bound(data.nbytes)
        cap = bound if rng.random() < 0.6 else int(rng.integers(0, bound + 1))
        r1, ref = oracle_compress(oracle, data, T, 1, cap)
        out = np.full(cap + 64, 0xA5, dtype=np.uint8)
        lib.stenos_set_level(ctx, 1)
        r2 = lib.stenos_compress_generic(ctx, np_ptr(data), T, data.nbytes, np_ptr(out), cap)
        what = (T, kind, n, seed, cap)
        assert (out[cap:] == 0xA5).all(), what
        assert has_error(r1) == has_error(r2), what
        if not has_error(r1):
            assert r1 == r2 and np.array_equal(ref, out[:r2]), what
            back = np.full(data.nbytes + 64, 0x5A, dtype=np.uint8)
            r3 = lib.stenos_decompress_generic(ctx, np_ptr(out), T, r2, np_ptr(back), data.nbytes)
            assert r3 == data.nbytes and np.array_equal(back[: data.nbytes], data) and (back[data.nbytes:] == 0x5A).all(), what
        cases += 1
    lib.stenos_destroy_context(ctx)
    assert cases > 0
    print(f"{cases} random cases")


def test_random_cases_at_every_level_against_oracle(oracle):
    """The same for the strategy layer: level 0..9, bytesoftype 1..16 (bytesoftype 1 and levels >= 2 go through the LZ4-dry
    estimates and zstd on the host around the GPU passes), small inputs so that zstd stays cheap, destinations from roomy
    to too small.  The oracle calls the same libzstd, so frames are compared byte for byte; where the compiled reference is
    there (oracle/_ref) it has to agree as well and to decode the library's frame."""
    from _libs import load_ref

    lib = load_library()
    ref_det = load_ref(det=True)  # (None where oracle/_ref is not built)
    ctx = lib.stenos_make_context()
    rng = np.random.default_rng(int(os.environ.get("STENOS_FUZZ_SEED", "20261004")))
    budget = float(os.environ.get("STENOS_FUZZ_LEVEL_SECONDS", "25"))
    t0, cases, by_level = time.time(), 0, {}
    kinds = KINDS + ["smooth8"]
    while time.time() - t0 < budget:
        level = int(rng.integers(0, 10))
        T = 1 if rng.random() < 0.3 else int(rng.integers(1, 17))
        kind = kinds[int(rng.integers(len(kinds)))]
        if kind == "smooth8" and T != 1:
            kind = "walk"
        shape = rng.random()
        nbytes = int(rng.integers(1, 2000)) if shape < 0.15 else int(rng.integers(2000, 300_000)) if shape < 0.7 else int(rng.integers(300_000, 1_500_000))
        n = max(1, nbytes // T)
        seed = int(rng.integers(1 << 30))
        data = generate(kind, T, n, seed)
        bound = lib.stenos_bound(data.nbytes)
        cap = bound if rng.random() < 0.7 else int(rng.integers(0, bound + 1))
        r1, ref = oracle_compress(oracle, data, T, level, cap)
        out = np.full(cap + 64, 0xA5, dtype=np.uint8)
        lib.stenos_set_level(ctx, level)
        r2 = lib.stenos_compress_generic(ctx, np_ptr(data), T, data.nbytes, np_ptr(out), cap)
        what = (level, T, kind, n, seed, cap)
        assert (out[cap:] == 0xA5).all(), what
        assert has_error(r1) == has_error(r2), what
        if not has_error(r1):
            assert r1 == r2 and np.array_equal(ref, out[:r2]), what
            back = np.full(data.nbytes + 64, 0x5A, dtype=np.uint8)
            r3 = lib.stenos_decompress_generic(ctx, np_ptr(out), T, r2, np_ptr(back), data.nbytes)
            assert r3 == data.nbytes and np.array_equal(back[: data.nbytes], data) and (back[data.nbytes:] == 0x5A).all(), what
            if ref_det is not None and cap == bound and cases % 3 == 0:
                exp = np.zeros(cap + 64, dtype=np.uint8)
                r4 = ref_det.stenos_compress(np_ptr(data), T, data.nbytes, np_ptr(exp), cap, level)
                assert r4 == r2 and np.array_equal(exp[:r4], out[:r2]), what
        by_level[level] = by_level.get(level, 0) + 1
        cases += 1
    lib.stenos_destroy_context(ctx)
    assert cases > 0 and len(by_level) >= 5
    print(f"{cases} random cases over levels {sorted(by_level)}")
