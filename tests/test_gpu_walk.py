"""The parallel walk of the superblock chain on the GPU (walk_kernels.hip, walk.h) against the serial walk by one lane
(the form it replaces and falls back to; reference stenos.cpp:1126-1134, 1166-1182) and against the index the encoder
leaves behind.  Frames without an index must decode the same way, whole, cut short or followed by other bytes."""
import time

import numpy as np
import pytest
import torch

from stenos_amd.api import Stenos, StenosError
from stenos_amd.datagen import generate_torch

pytestmark = pytest.mark.gpu


def _index(st, frame, T, csize, serial):
    st.lib.stenos_hip_test_walk(st.ctx, 1 if serial else 0)
    try:
        return st.frame_index(frame, T, csize)
    except StenosError:
        return None
    finally:
        st.lib.stenos_hip_test_walk(st.ctx, 0)


@pytest.mark.parametrize("kind,T,mib", [("rand12", 4, 300), ("rand", 4, 64), ("sorted_i32", 4, 512), ("sorted_i32", 4, 3000), ("walk", 2, 200), ("sine", 8, 100), ("sine", 4, 90), ("rand8", 2, 33)])
def test_parallel_walk_equals_serial_walk_and_encoder_index(kind, T, mib, hooks_lib):
    n = (mib << 20) // T + 4321
    src = generate_torch(kind, T, n, 17)
    st = Stenos(1, lib=hooks_lib)
    dst = torch.empty(st.bound(src.numel()), dtype=torch.uint8, device="cuda")
    c = st.compress(src, T, dst)
    p, nsb = st.last_index()
    enc = torch.empty(nsb + 1, dtype=torch.int64)
    import ctypes

    hip = ctypes.CDLL("libamdhip64.so")
    assert hip.hipMemcpy(ctypes.c_void_p(enc.data_ptr()), ctypes.c_void_p(p), ctypes.c_size_t((nsb + 1) * 8), 2) == 0
    par = _index(st, dst, T, c, serial=False)
    fell_back = st.lib.stenos_hip_test_walk(st.ctx, 0)
    ser = _index(st, dst, T, c, serial=True)
    assert par is not None and par == ser
    assert fell_back == 0, "a well-formed frame needed the serial walk"
    assert par == enc.tolist()
    # frame-only decode: no index handed over
    back = torch.zeros_like(src)
    assert st.decompress(dst, T, c, back) == src.numel()
    assert torch.equal(back, src)
    st.close()


def test_cut_and_padded_frames_agree_with_the_serial_walk(hooks_lib):
    T = 4
    src = generate_torch("rand12", T, (96 << 20) // T + 99, 3)
    st = Stenos(1, lib=hooks_lib)
    dst = torch.full((st.bound(src.numel()) + 70000,), 0x11, dtype=torch.uint8, device="cuda")
    c = st.compress(src, T, dst)
    rng = np.random.default_rng(1)
    sizes = [c - 1, c - 2, c - 3, c - 4, c - 5, c - 1000, c + 1, c + 4, c + 5, c + 60000] + [int(x) for x in rng.integers(c // 2, c, 12)]
    for size in sizes:
        par = _index(st, dst, T, size, serial=False)
        ser = _index(st, dst, T, size, serial=True)
        assert par == ser, size
    # a header in the middle of the chain damaged: whatever the serial walk makes of it
    idx = _index(st, dst, T, c, serial=True)
    for s in (1, len(idx) // 2, len(idx) - 2):
        saved = dst[idx[s] : idx[s] + 4].clone()
        for patch in ([0, None, None, None], [None, None, None, 0x7F], [9, 0, 0, 0]):
            for k, v in enumerate(patch):
                if v is not None:
                    dst[idx[s] + k] = v
            assert _index(st, dst, T, c, serial=False) == _index(st, dst, T, c, serial=True), (s, patch)
            dst[idx[s] : idx[s] + 4] = saved
    back = torch.zeros_like(src)
    assert st.decompress(dst, T, c, back) == src.numel() and torch.equal(back, src)
    st.close()


def test_walk_time_is_small_next_to_the_decode(hooks_lib):
    """8 GiB-class frames have tens of thousands of superblocks: one lane needs 0.35 us for each; the parallel walk must
    stay well below the decode kernel's time (2.4 ms per 8 GiB of int32)."""
    T = 4
    src = generate_torch("rand12", T, (2 << 30) // T, 42)
    st = Stenos(1, lib=hooks_lib)
    dst = torch.empty(st.bound(src.numel()), dtype=torch.uint8, device="cuda")
    c = st.compress(src, T, dst)
    times = {}
    for serial in (True, False):
        st.lib.stenos_hip_test_walk(st.ctx, 1 if serial else 0)
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            st.frame_index(dst, T, c)
            best = min(best, time.perf_counter() - t0)
        times[serial] = best
    st.lib.stenos_hip_test_walk(st.ctx, 0)
    st.close()
    print(f"frame index of 2 GiB int32: serial {times[True] * 1e3:.2f} ms, parallel {times[False] * 1e3:.2f} ms")
    assert times[False] < times[True] / 4
