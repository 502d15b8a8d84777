"""Incompressible stretches on the GPU (kernels.hip, speculative copy): a superblock that follows a copy is only measured,
and while it is, its raw bytes are put where a copy behind copies stands; they stay there when the guess holds and are
overwritten by whoever owns the place when it does not.  Frames must equal the oracle's byte for byte (the reference stores
such superblocks as code 6, stenos.cpp:609-610) -- all copies, copies and coded superblocks in every alternation, and with
more superblocks than resident workgroups, so that every workgroup gets past its first superblock."""
import numpy as np
import pytest
import torch

from _libs import oracle_compress
from stenos_amd.api import Stenos
from stenos_amd.datagen import generate

pytestmark = pytest.mark.gpu


def _check(oracle, data, T):
    st = Stenos(1)
    src = torch.from_numpy(data.view(np.uint8).ravel()).cuda()
    cap = st.bound(src.numel())
    dst = torch.full((cap + 256,), 0xC3, dtype=torch.uint8, device="cuda")
    c = st.compress(src, T, dst[:cap])
    r, ref = oracle_compress(oracle, data, T, 1)
    assert c == r
    got = dst[:c].cpu().numpy()
    assert np.array_equal(got, ref), np.nonzero(got != ref)[0][:8]
    assert bool((dst[cap:] == 0xC3).all()), "wrote past dst_size"
    back = torch.zeros_like(src)
    assert st.decompress(dst, T, c, back) == src.numel() and torch.equal(back, src)
    st.close()
    return ref


@pytest.mark.parametrize("T,mib", [(4, 640), (2, 512), (8, 512), (3, 300)])
def test_all_copies(oracle, T, mib):
    data = generate("rand", T, (mib << 20) // T, 31)
    frame = _check(oracle, data, T)
    assert frame[8] == 6  # (the case is what it says)


@pytest.mark.parametrize("T,mib,seed", [(4, 640, 1), (4, 400, 2), (2, 384, 3), (8, 384, 4)])
def test_copies_and_coded_superblocks_mixed(oracle, T, mib, seed):
    """Stretches of 1..40 superblocks of noise, 12-bit values, sorted values and constants in random order: guesses fail in
    both directions all the time."""
    rng = np.random.default_rng(seed)
    sb = 131072 // (256 * T) * 256 * T
    total = (mib << 20) // sb
    parts, left = [], total
    kinds = ["rand", "rand12" if T == 4 else "walk" if T == 2 else "sorted", "sorted", "same", "rand"]
    while left:
        n = int(min(left, rng.integers(1, 41)))
        kind = kinds[int(rng.integers(0, len(kinds)))]
        parts.append(generate(kind, T, n * sb // T, int(rng.integers(0, 1 << 30))).view(np.uint8).ravel())
        left -= n
    data = np.concatenate(parts)
    extra = generate("rand", T, 1000, 5).view(np.uint8).ravel()  # a partial superblock and block at the end
    data = np.concatenate([data, extra])
    frame = _check(oracle, np.ascontiguousarray(data), T)
    codes = set()
    p = 8
    while p + 4 <= frame.size:
        codes.add(int(frame[p]))
        p += 4 + (int(frame[p + 1]) | (int(frame[p + 2]) << 8) | (int(frame[p + 3]) << 16))
    assert {1, 6} <= codes
