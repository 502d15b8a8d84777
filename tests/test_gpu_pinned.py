"""Host-pointer calls on page-locked memory (capi.cpp, device_alias): when the caller's source and / or destination are
page-locked (hipHostMalloc, hipHostRegister, a pinned tensor) the kernels read and write them through the link instead of
staging a copy in device memory.  Same frames, same decoded bytes, nothing written past the destination."""
import numpy as np
import pytest

from _libs import has_error, np_ptr
from stenos_amd.datagen import generate

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    from stenos_amd.api import load_library

    return load_library()


def _pinned(nbytes):
    import torch

    return torch.empty(nbytes, dtype=torch.uint8, pin_memory=True)


@pytest.mark.parametrize("kind,T,n", [("rand12", 4, 9_000_123), ("walk", 2, 5_000_001), ("sine", 8, 1_500_003), ("mixed", 3, 700_001), ("rand", 4, 3_000_000), ("sorted_i32", 4, 4000)])
@pytest.mark.parametrize("pin_src,pin_dst", [(True, True), (True, False), (False, True)])
def test_pinned_buffers_give_the_same_frame(lib, kind, T, n, pin_src, pin_dst):
    data = generate(kind, T, n, 3).view(np.uint8).ravel()
    nb = data.nbytes
    cap = lib.stenos_bound(nb)
    ref = np.zeros(cap, dtype=np.uint8)
    r0 = lib.stenos_compress(np_ptr(data), T, nb, np_ptr(ref), cap, 1)
    assert not has_error(r0)
    keep = []  # the tensors own the pinned memory
    if pin_src:
        t = _pinned(nb + 64)
        keep.append(t)
        src = t.numpy()[32:32 + nb]  # (an odd offset inside the registration)
        src[:] = data
    else:
        src = data
    if pin_dst:
        t = _pinned(cap + 128)
        keep.append(t)
        whole = t.numpy()
        whole[:] = 0x5A
        dst = whole[64:64 + cap]
    else:
        whole = np.full(cap + 128, 0x5A, dtype=np.uint8)
        dst = whole[64:64 + cap]
    r = lib.stenos_compress(np_ptr(src), T, nb, np_ptr(dst), cap, 1)
    assert r == r0 and np.array_equal(dst[:r], ref[:r0])
    assert (whole[:64] == 0x5A).all() and (whole[64 + cap:] == 0x5A).all()
    # decode: frame pinned or not, destination pinned or not
    frame = dst[:r]
    if pin_src:
        t = _pinned(nb + 128)
        keep.append(t)
        w2 = t.numpy()
    else:
        w2 = np.zeros(nb + 128, dtype=np.uint8)
    w2[:] = 0xA5
    back = w2[48:48 + nb]
    assert lib.stenos_decompress(np_ptr(frame), T, r, np_ptr(back), nb) == nb
    assert np.array_equal(back, data)
    assert (w2[:48] == 0xA5).all() and (w2[48 + nb:] == 0xA5).all()


def test_pinned_destination_too_small_is_an_error_not_an_overrun(lib):
    T = 4
    data = generate("rand12", T, 2_000_000, 1).view(np.uint8).ravel()
    nb = data.nbytes
    ref = np.zeros(lib.stenos_bound(nb), dtype=np.uint8)
    r0 = lib.stenos_compress(np_ptr(data), T, nb, np_ptr(ref), ref.nbytes, 1)
    t = _pinned(r0 + 256)
    whole = t.numpy()
    for cap in (r0 + 4096, r0, r0 - 1, r0 // 2, 7):  # (the frame depends on the room near its end: compare with the pageable path at the same capacity)
        whole[:] = 0x5A
        r = lib.stenos_compress(np_ptr(data), T, nb, np_ptr(whole[: r0 + 200]), cap, 1) if cap <= r0 + 200 else None
        if r is None:
            continue
        assert (whole[cap:] == 0x5A).all(), cap
        exp = np.full(cap + 64, 0x5A, dtype=np.uint8)
        re = lib.stenos_compress(np_ptr(data), T, nb, np_ptr(exp), cap, 1)
        assert has_error(r) == has_error(re), (cap, hex(r), hex(re))
        if not has_error(r):
            assert r == re and np.array_equal(whole[:r], exp[:re])
        if cap < r0 // 2 + 1:
            assert has_error(r)
