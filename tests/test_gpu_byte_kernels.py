"""shuffle / unshuffle / delta / delta_inv kernels (include/stenos_hip.h) against the oracle's restatement
of shuffle-generic.h:33-125 and delta.cpp:30-71, 230-268: bit-exact, odd sizes, leftover bytes, the
2048-byte single-stream threshold and the four quarter streams."""
import numpy as np
import pytest

from _libs import np_ptr
from stenos_amd.datagen import generate

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch

    from stenos_amd.api import load_library

    assert torch.cuda.is_available()
    return load_library(), torch


def _run(lib, torch, fn, data, *args):
    src = torch.from_numpy(data).cuda()
    dst = torch.full((data.nbytes + 64,), 0x77, dtype=torch.uint8, device="cuda")
    r = fn(src.data_ptr(), *args, dst.data_ptr(), torch.cuda.current_stream().cuda_stream) if fn.__name__.endswith("shuffle") else \
        fn(src.data_ptr(), dst.data_ptr(), *args, torch.cuda.current_stream().cuda_stream)
    assert r == 0
    torch.cuda.synchronize()
    out = dst.cpu().numpy()
    assert (out[data.nbytes:] == 0x77).all(), "wrote past the end"
    return out[: data.nbytes]


@pytest.mark.parametrize("T", [1, 2, 3, 4, 5, 7, 8, 12, 16, 33, 64])
def test_shuffle_roundtrip_and_oracle(env, oracle, T):
    lib, torch = env
    for nbytes in (0, 1, T, 5 * T + 3, 255 * T, 1024 * T + 1, 131072, 131072 + T + 1, 1_000_003):
        data = generate("rand", 1, nbytes, 3 + nbytes % 7)
        ref = np.zeros(nbytes, dtype=np.uint8)
        oracle.so_shuffle(T, nbytes, np_ptr(data), np_ptr(ref))
        got = _run(lib, torch, lib.stenos_hip_shuffle, data, T, nbytes)
        assert np.array_equal(got, ref), (T, nbytes)
        back = _run(lib, torch, lib.stenos_hip_unshuffle, ref, T, nbytes)
        assert np.array_equal(back, data), (T, nbytes)


def test_delta_roundtrip_and_oracle(env, oracle):
    lib, torch = env
    for nbytes in (0, 1, 2, 15, 16, 17, 2047, 2048, 2049, 2051, 4096, 65537, 131072, 262147, 3_000_001):
        data = generate("walk", 1, nbytes, 5) if nbytes % 2 else generate("rand", 1, nbytes, 5)
        ref = np.zeros(nbytes, dtype=np.uint8)
        oracle.so_delta(np_ptr(data), np_ptr(ref), nbytes)
        got = _run(lib, torch, lib.stenos_hip_delta, data, nbytes)
        assert np.array_equal(got, ref), nbytes
        back = _run(lib, torch, lib.stenos_hip_delta_inv, ref, nbytes)
        assert np.array_equal(back, data), nbytes
