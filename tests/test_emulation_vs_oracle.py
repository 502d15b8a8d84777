"""The wave-level codec SOURCE of the product (stenos_amd/csrc/*.h) compiled for the host as a
64-lane lockstep emulation (tests/emul) and diffed against the oracle.  Runs without a GPU; it proves
the kernel logic, not the shipped binary (the -m gpu tests do that through the C ABI)."""
import ctypes
import os
import subprocess
from ctypes import c_int, c_size_t, c_void_p

import numpy as np
import pytest

from _libs import ROOT, has_error, np_ptr
from stenos_amd.datagen import generate

KINDS = ["rand", "same", "sorted", "walk", "ramp", "dict16", "runs", "burst", "mixed", "lzmix", "edge_noise", "steps", "slopes"]


@pytest.fixture(scope="module", params=["libstenos_emul.so", "libstenos_emul_enc.so"], ids=["decode-shapes", "encode-shapes"])
def emul(request):
    """Both builds of tests/emul/Makefile: the shared copy helpers as the decode kernels and as the encoders compile them."""
    d = os.path.join(ROOT, "tests", "emul")
    subprocess.check_call(["make", "-C", d], stdout=subprocess.DEVNULL)
    lib = ctypes.CDLL(os.path.join(d, request.param))
    lib.emul_block_compress.restype = c_size_t
    lib.emul_block_compress.argtypes = [c_void_p, c_size_t, c_size_t, c_void_p, c_int]
    lib.emul_block_decompress.restype = c_size_t
    lib.emul_block_decompress.argtypes = [c_void_p, c_size_t, c_size_t, c_size_t, c_void_p, c_int]
    lib.emul_copy_g2g_wide.restype = None
    lib.emul_copy_g2g_wide.argtypes = [c_void_p, c_void_p, c_size_t]
    lib.emul_set_fused.restype = None
    lib.emul_set_fused.argtypes = [c_int]
    lib.emul_set_slots.restype = None
    lib.emul_set_slots.argtypes = [c_int]
    lib.emul_set_dec_regs.restype = None
    lib.emul_set_dec_regs.argtypes = [c_int]
    return lib


@pytest.mark.parametrize("T", [2, 3, 4, 5, 7, 8, 12, 16, 24, 33, 64])
def test_kernel_logic_matches_oracle(oracle, emul, T):
    sizes = [16, 100, 255, 256, 257, 300, 512, 1280, 4099] if T <= 16 else [16, 257, 1280]
    for kind in KINDS + (["rand12"] if T == 4 else []):
        for n in sizes:
            data = generate(kind, T, n, 42 + n)
            nb = data.nbytes
            ref = np.zeros(nb * 2 + 4096, dtype=np.uint8)
            r1 = oracle.so_block_compress(np_ptr(data), T, nb, np_ptr(ref), ref.nbytes)
            out = np.zeros(nb * 2 + 4096, dtype=np.uint8)
            r2 = emul.emul_block_compress(np_ptr(data), T, nb, np_ptr(out), 1)
            assert r1 == r2, (kind, n)
            assert np.array_equal(ref[:r1], out[:r1]), (kind, n)
            for mis in (0, 7):
                dec = np.zeros(nb + 64, dtype=np.uint8)
                r3 = emul.emul_block_decompress(np_ptr(ref), r1, T, nb, np_ptr(dec), mis)
                assert r3 == nb, (kind, n, mis)
                assert np.array_equal(dec[:nb], data), (kind, n, mis)
                assert not dec[nb:].any()


def test_truncated_streams_are_rejected_not_overrun(oracle, emul):
    """Every prefix of a valid payload must decode to an error (block_compress.h:1560, 1575, 1591-1598, 1642)."""
    for kind, T in (("burst", 4), ("dict16", 4), ("walk", 2), ("runs", 8)):
        data = generate(kind, T, 700, 3)
        nb = data.nbytes
        ref = np.zeros(nb * 2 + 4096, dtype=np.uint8)
        r1 = oracle.so_block_compress(np_ptr(data), T, nb, np_ptr(ref), ref.nbytes)
        for cut in list(range(0, min(r1, 80))) + list(range(max(0, r1 - 40), r1)):
            dec = np.zeros(nb + 64, dtype=np.uint8)
            r3 = emul.emul_block_decompress(np_ptr(ref), cut, T, nb, np_ptr(dec), 0)
            assert has_error(r3) or (cut == 0 and r3 == 0), (kind, cut, r3)


def test_wide_copy_any_alignment(emul):
    """copy_g2g_wide (staging run -> frame, COPY superblocks): every source / destination misalignment, sizes around the
    16-byte group and around the 256 groups of one round; nothing is written outside the destination range."""
    rng = np.random.default_rng(1)
    src = rng.integers(0, 256, size=6000, dtype=np.uint8)
    for so in range(0, 17):
        for do in range(0, 18):
            for n in (0, 1, 2, 15, 16, 17, 31, 32, 33, 63, 64, 65, 255, 1023, 1040, 3001, 4095, 4113, 5000):
                dst = np.zeros(6000, dtype=np.uint8)
                emul.emul_copy_g2g_wide(np_ptr(dst) + do, np_ptr(src) + so, n)
                assert np.array_equal(dst[do:do + n], src[so:so + n]), (so, do, n)
                assert not dst[:do].any() and not dst[do + n:].any(), (so, do, n)


@pytest.mark.parametrize("fused", [1, 0])
@pytest.mark.parametrize("T", [2, 4, 8, 3, 12])
def test_frame_pipeline_and_capacity_rules(oracle, emul, T, fused):
    """The whole encode pipeline (encode_superblocks for the leading superblocks with ample room, then
    encode_blocks, plan, scan, resolve, pack, as capi.cpp enqueues them; fused=0: without the fused
    kernel) against the oracle's serial path for dst_size = bound, larger, and smaller: same frame or
    both an error.  Covers the reference's capacity-dependent LZ attempt (block_compress.h:1214) and
    dst_end tests (:1225, 1241, 1284) that decide BLOCK vs COPY near the end of the buffer."""
    from stenos_amd.datagen import splitmix64

    emul.emul_set_fused(fused)

    emul.emul_compress_frame.restype = c_size_t
    emul.emul_compress_frame.argtypes = [c_void_p, c_size_t, c_size_t, c_void_p, c_size_t, c_int]
    from _libs import oracle_compress

    per = 131072 // (256 * T) * 256
    emul.emul_last_fused.restype = c_size_t
    replayed = fused_superblocks = 0
    for kind in KINDS:
        for n in [1, 15, 17, 100, 255, 256, 257, 511, 1280, 4099, per - 1, per, per + 1, 2 * per + 300]:
            data = generate(kind, T, n, 77 + n)
            bound = oracle.so_bound(data.nbytes)
            caps = [bound + 5000, bound] + [max(0, bound - int(x)) for x in (splitmix64(n, 4) % np.uint64(max(10, data.nbytes // 3)))]
            big = None
            for cap in caps:
                for level in (1, 0):
                    r1, f1 = oracle_compress(oracle, data, T, level, cap)
                    out = np.zeros(cap + 64, dtype=np.uint8)
                    r2 = emul.emul_compress_frame(np_ptr(data), T, data.nbytes, np_ptr(out), cap, level)
                    assert has_error(r1) == has_error(r2), (kind, n, cap, level, hex(r1), hex(r2))
                    fused_superblocks += emul.emul_last_fused() if level == 1 else 0
                    if not has_error(r1):
                        assert r1 == r2 and np.array_equal(f1, out[:r2]), (kind, n, cap, level)
                        if level == 1 and cap == bound + 5000:
                            big = r1
                        if level == 1 and cap == bound and big is not None and r1 != big:
                            replayed += 1
    emul.emul_set_fused(1)
    assert (fused_superblocks > 0) == bool(fused)
    if T % 4 == 0:
        assert replayed > 0, "no case exercised the capacity replay"


def test_fused_path_plane_group_loop_for_int32(oracle, emul):
    """The twin of the fused kernel that the plane probe selects for int32 data with three or four non-constant planes
    per block (kernels.hip, probe_planes) runs encode_run without the slots."""
    from _libs import oracle_compress

    emul.emul_set_fused(1)
    emul.emul_set_slots(0)
    emul.emul_compress_frame.restype = c_size_t
    emul.emul_compress_frame.argtypes = [c_void_p, c_size_t, c_size_t, c_void_p, c_size_t, c_int]
    try:
        per = 131072 // (256 * 4) * 256
        for kind in KINDS:
            data = generate(kind, 4, 2 * per + 300, 21)
            r1, f1 = oracle_compress(oracle, data, 4, 1)
            out = np.zeros(r1 + 5000, dtype=np.uint8)
            r2 = emul.emul_compress_frame(np_ptr(data), 4, data.nbytes, np_ptr(out), out.nbytes, 1)
            assert r2 == r1 and np.array_equal(out[:r2], f1), kind
    finally:
        emul.emul_set_slots(1)


@pytest.mark.parametrize("T", [2, 4])
def test_fused_path_with_misaligned_source(oracle, emul, T):
    """Blocks that do not start on 16-byte boundaries take the byte-wise loads of the slot encoder (superblock_codec.h, encode_run)."""
    from _libs import oracle_compress

    emul.emul_set_fused(1)
    emul.emul_compress_frame.restype = c_size_t
    emul.emul_compress_frame.argtypes = [c_void_p, c_size_t, c_size_t, c_void_p, c_size_t, c_int]
    per = 131072 // (256 * T) * 256
    for kind in ("mixed", "lzmix", "walk", "sorted"):
        data = generate(kind, T, 2 * per + 300, 5)
        r1, f1 = oracle_compress(oracle, data, T, 1)
        for off in (1, 4, 8, 15):
            buf = np.zeros(data.nbytes + 64, dtype=np.uint8)
            buf[off:off + data.nbytes] = data
            out = np.zeros(r1 + 5000, dtype=np.uint8)
            r2 = emul.emul_compress_frame(np_ptr(buf) + off, T, data.nbytes, np_ptr(out), out.nbytes, 1)
            assert r2 == r1 and np.array_equal(out[:r2], f1), (kind, off)


def test_decoder_lds_covers_the_reach_of_an_unchecked_block(emul):
    """The decoder checks the bytes a block consumed once per block (block_codec.h, decode_block), so whatever a block
    whose first byte lies inside the window can read must be inside the wave's LDS allocation, and the image must fit."""
    import ctypes

    emul.emul_dec_layout.restype = None
    emul.emul_dec_layout.argtypes = [c_size_t, ctypes.POINTER(ctypes.c_uint32)]
    for T in list(range(1, 65)):
        out = (ctypes.c_uint32 * 5)()
        emul.emul_dec_layout(T, out)
        win, img, total, wcap, reach = list(out)
        hs = (T + 1) // 2
        # worst case per plane: 8 header bytes + 18 (mask16 + 16 mins) + 16 rows of 18 bytes, + a 16-byte read at the end
        assert reach >= hs + T * (8 + 18 + 16 * 18) + 16, T
        assert win == 0 and img >= wcap + 16, T
        assert total >= wcap + reach, T          # a block starting at the window's last byte stays inside
        assert total >= img + 256 * T, T         # the decoded block
        assert total <= 64 * 1024, T             # one workgroup's limit


@pytest.mark.parametrize("T", [2, 4])
def test_slot_rows_encoder_on_plane_mixtures(oracle, emul, T):
    """The row-lane encoder of the fused path (slot_codec.h) on inputs whose blocks differ in which planes are constant,
    run-length coded, delta coded, raw or LZ coded, so that batches of one and two blocks with every slot assignment
    occur: frames equal the oracle's."""
    from _libs import oracle_compress

    emul.emul_set_fused(1)
    emul.emul_set_slots(1)
    emul.emul_compress_frame.restype = c_size_t
    emul.emul_compress_frame.argtypes = [c_void_p, c_size_t, c_size_t, c_void_p, c_size_t, c_int]
    per = 131072 // (256 * T) * 256
    for kind in KINDS + (["rand12"] if T == 4 else []):
        for seed in (1, 2, 3):
            data = generate(kind, T, 3 * per + 256 * seed + 77, 100 + seed)
            cap = oracle.so_bound(data.nbytes) + 5000
            r1, f1 = oracle_compress(oracle, data, T, 1, cap)
            out = np.zeros(cap, dtype=np.uint8)
            r2 = emul.emul_compress_frame(np_ptr(data), T, data.nbytes, np_ptr(out), cap, 1)
            assert emul.emul_last_fused() == 3
            assert r2 == r1 and np.array_equal(out[:r2], f1), (kind, seed)


def test_groups_of_blocks_are_taken_where_they_are_meant_to(oracle, emul):
    """The inputs the group paths of the fused encoder were made for go through them (superblock_codec.h): 12-bit integers in
    32-bit elements in groups of four blocks (the same two planes in every block), a random walk of 16-bit samples -- one or
    two planes to code per block, in no fixed pattern -- in groups of any shape; frames equal the oracle's either way."""
    from _libs import oracle_compress

    emul.emul_set_fused(1)
    emul.emul_set_slots(1)
    emul.emul_compress_frame.restype = c_size_t
    emul.emul_compress_frame.argtypes = [c_void_p, c_size_t, c_size_t, c_void_p, c_size_t, c_int]
    emul.emul_group4_count.restype = c_size_t
    emul.emul_group_any_count.restype = c_size_t
    for kind, T, want4, want_any in (("rand12", 4, True, False), ("walk", 2, False, True), ("mixed", 2, False, True), ("dict16", 4, False, False)):
        per = 131072 // (256 * T) * 256
        data = generate(kind, T, 3 * per + 333, 9)
        blocks = data.nbytes // (256 * T)
        cap = oracle.so_bound(data.nbytes) + 5000
        r1, f1 = oracle_compress(oracle, data, T, 1, cap)
        out = np.zeros(cap, dtype=np.uint8)
        g4, ga = emul.emul_group4_count(), emul.emul_group_any_count()
        r2 = emul.emul_compress_frame(np_ptr(data), T, data.nbytes, np_ptr(out), cap, 1)
        g4, ga = emul.emul_group4_count() - g4, emul.emul_group_any_count() - ga
        assert r2 == r1 and np.array_equal(out[:r2], f1), kind
        assert (g4 * 4 >= blocks - 8) == want4, (kind, g4, blocks)
        assert (ga > blocks // 16) == want_any, (kind, ga, blocks)



def test_planes_of_run_length_coded_values_take_the_short_form(oracle, emul):
    """decode_plane_runs / decode_plane_slopes (block_codec.h): planes all of whose rows are run-length coded values (`steps`,
    `runs`) or differences (`slopes`) are decoded by the short forms, to the same bytes; other planes keep the general form."""
    emul.emul_plane_runs_count.restype = c_size_t
    for kind, T, at_least in (("steps", 4, 40), ("runs", 2, 40), ("steps", 8, 40), ("slopes", 2, 20), ("slopes", 4, 20), ("rand", 4, 0)):
        data = generate(kind, T, 40 * 256, 3)
        nb = data.nbytes
        ref = np.zeros(nb * 2 + 4096, dtype=np.uint8)
        r1 = oracle.so_block_compress(np_ptr(data), T, nb, np_ptr(ref), ref.nbytes)
        for mis in (0, 5):
            dec = np.zeros(nb + 64, dtype=np.uint8)
            before = emul.emul_plane_runs_count()
            r3 = emul.emul_block_decompress(np_ptr(ref), r1, T, nb, np_ptr(dec), mis)
            assert r3 == nb and np.array_equal(dec[:nb], data), (kind, T, mis)
            took = emul.emul_plane_runs_count() - before
            assert took >= at_least and (at_least or took == 0), (kind, T, took)


def test_mini_lz_blocks_take_the_two_phase_decoder(oracle, emul):
    """lz_decode_256 (block_codec.h) gives up on nothing a valid stream holds -- near and far distances, raw groups, literals with
    their top bit set: the serial decoder behind it is for damaged streams (and the other element sizes) only."""
    emul.emul_lz_serial_count.restype = c_size_t
    for kind in ("dict16", "cycle130", "lzmix"):
        for T in (4, 8):
            data = generate(kind, T, 64 * 256, 5)
            nb = data.nbytes
            ref = np.zeros(nb * 2 + 4096, dtype=np.uint8)
            r1 = oracle.so_block_compress(np_ptr(data), T, nb, np_ptr(ref), ref.nbytes)
            assert (ref[:r1] == 253).any(), (kind, T)  # (a sanity check only: the marker byte of a mini-LZ block is around)
            dec = np.zeros(nb + 64, dtype=np.uint8)
            before = emul.emul_lz_serial_count()
            r3 = emul.emul_block_decompress(np_ptr(ref), r1, T, nb, np_ptr(dec), 0)
            assert r3 == nb and np.array_equal(dec[:nb], data), (kind, T)
            assert emul.emul_lz_serial_count() == before, (kind, T)


@pytest.mark.parametrize("T", [2, 4, 8])
def test_fused_path_across_copy_and_block_superblocks(oracle, emul, T):
    """After a superblock that ended up as a copy the fused kernel only measures the next one and encodes it for real when
    it does compress after all (kernels.hip, encode_superblocks): inputs that alternate between noise and compressible
    stretches, cut in and off superblock boundaries."""
    from _libs import oracle_compress

    emul.emul_set_fused(1)
    emul.emul_set_slots(1)
    emul.emul_compress_frame.restype = c_size_t
    emul.emul_compress_frame.argtypes = [c_void_p, c_size_t, c_size_t, c_void_p, c_size_t, c_int]
    per = 131072 // (256 * T) * 256
    for pattern in (("rand", "walk", "rand", "rand", "mixed", "walk", "rand"), ("walk", "rand", "walk"), ("rand", "rand", "burst")):
        for cut in (per, per + per // 3):
            data = np.concatenate([generate(kind, T, cut, 50 + i) for i, kind in enumerate(pattern)])
            cap = oracle.so_bound(data.nbytes) + 5000
            r1, f1 = oracle_compress(oracle, data, T, 1, cap)
            out = np.zeros(cap, dtype=np.uint8)
            r2 = emul.emul_compress_frame(np_ptr(data), T, data.nbytes, np_ptr(out), cap, 1)
            assert emul.emul_last_fused() > 0
            assert r2 == r1 and np.array_equal(out[:r2], f1), (pattern, cut)


@pytest.mark.parametrize("T", [65, 100, 128, 132, 508, 512, 516, 1000, 4100])
def test_wide_types_block_codec(oracle, emul, T):
    """bytesoftype above 64 (kernels_wide.hip runs this same source with its scratch in HBM): planes are handled 64 at a
    time (plane_offsets, type nibbles), the mini-LZ is tried up to bytesoftype 512 only (lz_compress.h:281-283, 15-bit
    distances) and a block no longer fits 16-bit window offsets."""
    for kind in ["rand", "same", "walk", "dict16", "runs", "burst", "mixed", "lzmix", "sorted"]:
        for n in (256, 300, 515):
            data = generate(kind, T, n, 42 + n)
            nb = data.nbytes
            ref = np.zeros(nb * 2 + 4096, dtype=np.uint8)
            r1 = oracle.so_block_compress(np_ptr(data), T, nb, np_ptr(ref), ref.nbytes)
            out = np.zeros(nb * 2 + 4096, dtype=np.uint8)
            r2 = emul.emul_block_compress(np_ptr(data), T, nb, np_ptr(out), 1)
            assert r1 == r2 and np.array_equal(ref[:r1], out[:r1]), (kind, n)
            dec = np.zeros(nb + 64, dtype=np.uint8)
            r3 = emul.emul_block_decompress(np_ptr(ref), r1, T, nb, np_ptr(dec), 5)
            assert r3 == nb and np.array_equal(dec[:nb], data) and not dec[nb:].any(), (kind, n)


def test_widest_type_block_codec(oracle, emul):
    T = 65534  # stenos.h:65: the largest bytesoftype the reference accepts; plane offsets reach 280 * T
    for kind, n in (("mixed", 300), ("walk", 515), ("rand", 256)):
        data = generate(kind, T, n, 5)
        nb = data.nbytes
        ref = np.zeros(nb * 2 + 4096, dtype=np.uint8)
        r1 = oracle.so_block_compress(np_ptr(data), T, nb, np_ptr(ref), ref.nbytes)
        out = np.zeros(nb * 2 + 4096, dtype=np.uint8)
        r2 = emul.emul_block_compress(np_ptr(data), T, nb, np_ptr(out), 1)
        assert r1 == r2 and np.array_equal(ref[:r1], out[:r1]), kind
        dec = np.zeros(nb + 64, dtype=np.uint8)
        r3 = emul.emul_block_decompress(np_ptr(ref), r1, T, nb, np_ptr(dec), 0)
        assert r3 == nb and np.array_equal(dec[:nb], data), kind


@pytest.mark.parametrize("T", [65, 100, 512, 516, 700])
def test_wide_types_frame_pipeline_and_capacity_rules(oracle, emul, T):
    """encode_blocks / plan / scan / resolve / pack for bytesoftype above 64 against the oracle's serial path, with roomy
    and tight destinations (superblocks of a few blocks up to 512, of one block above)."""
    from _libs import oracle_compress
    from stenos_amd.datagen import splitmix64

    emul.emul_compress_frame.restype = c_size_t
    emul.emul_compress_frame.argtypes = [c_void_p, c_size_t, c_size_t, c_void_p, c_size_t, c_int]
    bs = 256 * T
    per = (131072 // bs * 256) if bs <= 131072 else 256
    for kind in ["rand", "walk", "dict16", "mixed", "lzmix", "burst"]:
        for n in [1, 100, 256, 257, per + 1, 2 * per + 300]:
            data = generate(kind, T, n, 77 + n)
            bound = oracle.so_bound(data.nbytes)
            caps = [bound + 5000, bound] + [max(0, bound - int(x)) for x in (splitmix64(n, 3) % np.uint64(max(10, data.nbytes // 3)))]
            for cap in caps:
                r1, f1 = oracle_compress(oracle, data, T, 1, cap)
                out = np.zeros(cap + 64, dtype=np.uint8)
                r2 = emul.emul_compress_frame(np_ptr(data), T, data.nbytes, np_ptr(out), cap, 1)
                assert has_error(r1) == has_error(r2), (kind, n, cap, hex(r1), hex(r2))
                if not has_error(r1):
                    assert r1 == r2 and np.array_equal(f1, out[:r2]), (kind, n, cap)


@pytest.mark.parametrize("T", [2, 4, 8])
def test_decoder_image_path_for_small_types(oracle, emul, T):
    """The kernels instantiated for bytesoftype 2, 4 and 8 write blocks made of planes to HBM from registers
    (decode_planes_to); a destination that is not 16-byte aligned sends them through the LDS image like every other
    bytesoftype.  Same bytes either way."""
    for kind in KINDS:
        data = generate(kind, T, 1280 + 77, 11)
        nb = data.nbytes
        ref = np.zeros(nb * 2 + 4096, dtype=np.uint8)
        r1 = oracle.so_block_compress(np_ptr(data), T, nb, np_ptr(ref), ref.nbytes)
        for regs in (0, 1):
            emul.emul_set_dec_regs(regs)
            dec = np.zeros(nb + 64, dtype=np.uint8)
            r3 = emul.emul_block_decompress(np_ptr(ref), r1, T, nb, np_ptr(dec), 3)
            assert r3 == nb and np.array_equal(dec[:nb], data) and not dec[nb:].any(), (kind, regs)
    emul.emul_set_dec_regs(1)


@pytest.mark.parametrize("T,dtype", [(4, "<u4"), (2, "<u2")])
def test_fused_path_wide_batches(oracle, emul, T, dtype):
    """Blocks with at most one non-constant plane share a pass three or four at a time (SlotBatch4, slot_codec.h): random
    sequences of such blocks (one varying byte anywhere in the element, all constant, runs) mixed with blocks of two
    planes that end a batch, through the fused path."""
    from _libs import oracle_compress

    emul.emul_compress_frame.restype = c_size_t
    emul.emul_compress_frame.argtypes = [c_void_p, c_size_t, c_size_t, c_void_p, c_size_t, c_int]
    emul.emul_last_fused.restype = c_size_t
    rng = np.random.default_rng(5 + T)
    for trial in range(6):
        parts = []
        for b in range(int(rng.integers(300, 700))):
            mode = int(rng.integers(0, 8))
            if mode == 0:
                v = np.full(256, int(rng.integers(0, 1 << (8 * T))), dtype=np.uint64)
            elif mode <= 3:
                v = rng.integers(0, 256, 256, dtype=np.uint64) + (int(rng.integers(0, 1 << (8 * (T - 1)))) << 8)
            elif mode == 4:
                v = (rng.integers(0, 256, 256, dtype=np.uint64) << 8) + int(rng.integers(0, 256))
            elif mode == 5:
                v = rng.integers(0, 256, 256, dtype=np.uint64) << (8 * (T - 1))
            elif mode == 6:
                v = rng.integers(0, 65536, 256, dtype=np.uint64)
            else:
                v = np.repeat(rng.integers(0, 256, 16, dtype=np.uint64), 16)
            parts.append(v)
        if trial % 2:
            parts.append(rng.integers(0, 256, int(rng.integers(1, 256)), dtype=np.uint64))
        data = np.concatenate(parts).astype(dtype).view(np.uint8)
        r1, ref = oracle_compress(oracle, data, T, 1)
        out = np.zeros(oracle.so_bound(data.nbytes) + 4096, dtype=np.uint8)  # (ample room: the leading superblocks take the fused path)
        r2 = emul.emul_compress_frame(np_ptr(data), T, data.nbytes, np_ptr(out), out.nbytes, 1)
        assert emul.emul_last_fused() > 0
        assert r1 == r2 and np.array_equal(ref, out[:r2]), trial


@pytest.mark.parametrize("T,dtype", [(2, np.uint16), (4, np.uint32)])
def test_fused_path_planes_of_noise(oracle, emul, T, dtype):
    """Blocks whose byte planes are, independently, constant, noise, noise of a few bits, a slow walk or runs -- so that RAW
    planes stand in front of, behind and between compressible planes and constant ones, blocks consist of RAW planes only,
    and batches of the slot encoder break in every place -- through the fused path against the oracle."""
    from _libs import oracle_compress

    emul.emul_compress_frame.restype = c_size_t
    emul.emul_compress_frame.argtypes = [c_void_p, c_size_t, c_size_t, c_void_p, c_size_t, c_int]
    emul.emul_last_fused.restype = c_size_t
    emul.emul_set_fused(1)
    emul.emul_set_slots(1)
    rng = np.random.default_rng(11 + T)
    for trial in range(8):
        stationary = trial % 4 != 3  # both a steady and a changing mix of planes
        kinds = rng.integers(0, 5, T)
        if stationary:
            kinds[int(rng.integers(0, T))] = 1  # at least one plane of noise
        parts = []
        for b in range(int(rng.integers(300, 650))):
            if not stationary or rng.integers(0, 40) == 0:
                kinds = rng.integers(0, 5, T)
                if stationary:
                    kinds[int(rng.integers(0, T))] = 1
            v = np.zeros(256, dtype=np.uint64)
            for k in range(T):
                kind = int(kinds[k])
                if kind == 0:
                    pl = np.full(256, int(rng.integers(0, 256)), dtype=np.uint64)
                elif kind == 1:
                    pl = rng.integers(0, 256, 256, dtype=np.uint64)
                elif kind == 2:
                    pl = rng.integers(0, 16, 256, dtype=np.uint64) + int(rng.integers(0, 200))
                elif kind == 3:
                    pl = (np.cumsum(rng.integers(-3, 4, 256)) + 128).astype(np.uint64) & 0xFF
                else:
                    pl = np.repeat(rng.integers(0, 256, 32, dtype=np.uint64), 8)
                v |= pl << np.uint64(8 * k)
            parts.append(v)
        if trial % 3 == 0:
            parts.append(rng.integers(0, 256, int(rng.integers(1, 256)), dtype=np.uint64))
        data = np.concatenate(parts).astype(dtype).view(np.uint8)
        r1, ref = oracle_compress(oracle, data, T, 1)
        out = np.zeros(oracle.so_bound(data.nbytes) + 4096, dtype=np.uint8)
        r2 = emul.emul_compress_frame(np_ptr(data), T, data.nbytes, np_ptr(out), out.nbytes, 1)
        assert emul.emul_last_fused() > 0
        assert r1 == r2 and np.array_equal(ref, out[:r2]), trial


@pytest.mark.parametrize("T", [2, 4, 8])
def test_plane_forms_of_round_4(oracle, emul, T):
    """The short plane forms of the decoder (block_codec.h decode_plane_packed: bit-packed rows, raw rows, run-length coded
    differences; header-7 rows through the general form) and the encoder's run-length rows by quads (more than sixteen per
    pass as well) and passes of noise (slot_codec.h), on inputs built to hold them -- the oracle's own counters say that they
    do -- against the oracle, byte for byte, both ways; the decoder with and without its register path."""
    from _libs import STAT_PLANE_TYPE, STAT_ROW_HDR, frame_stats, oracle_compress

    emul.emul_run_compress.restype = c_size_t
    emul.emul_run_compress.argtypes = [c_void_p, c_size_t, c_size_t, c_void_p]
    seen = np.zeros(16, dtype=np.uint64)
    raw_planes = 0
    for kind in ["sine", "walk", "noise_low", "steps", "slopes"]:
        if kind == "sine" and T == 2:
            continue
        for n in (256 * 7, 256 * 64):
            data = generate(kind, T, n, 5 + n)
            nb = data.nbytes
            r0, frame = oracle_compress(oracle, data, T, 1)
            st = frame_stats(oracle, frame, T)
            seen += st[STAT_ROW_HDR:STAT_ROW_HDR + 16]
            raw_planes += int(st[STAT_PLANE_TYPE + 1])
            if kind == "steps":  # (the gathered emission takes sixteen run-length rows at a time: a pass has to hold more)
                assert st[STAT_ROW_HDR + 7] > (n // 256) * 20, st[STAT_ROW_HDR:STAT_ROW_HDR + 16]
            if kind == "slopes":
                assert st[STAT_ROW_HDR + 6] > (n // 256) * 8, st[STAT_ROW_HDR:STAT_ROW_HDR + 16]
            ref = np.zeros(nb * 2 + 4096, dtype=np.uint8)
            r1 = oracle.so_block_compress(np_ptr(data), T, nb, np_ptr(ref), ref.nbytes)
            out = np.zeros(nb * 2 + 4096, dtype=np.uint8)
            r2 = emul.emul_run_compress(np_ptr(data), T, n // 256, np_ptr(out))
            assert r1 == r2 and np.array_equal(ref[:r1], out[:r1]), (kind, n)
            for regs in (1, 0):
                emul.emul_set_dec_regs(regs)
                for mis in (0, 3):
                    dec = np.zeros(nb + 64, dtype=np.uint8)
                    r3 = emul.emul_block_decompress(np_ptr(ref), r1, T, nb, np_ptr(dec), mis)
                    assert r3 == nb and np.array_equal(dec[:nb], data), (kind, n, regs, mis)
            emul.emul_set_dec_regs(1)
    # every row kind has been through: absolute rows, rows of differences, both run-length kinds, raw rows; and RAW planes
    assert seen[0:6].sum() > 0 and seen[8:15].sum() > 0 and seen[6] > 0 and seen[7] > 0 and seen[15] > 0 and raw_planes > 0, (seen, raw_planes)


@pytest.mark.parametrize("T", [4, 8])
def test_mini_lz_blocks_both_decoders(oracle, emul, T):
    """Blocks the mini-LZ took (block_codec.h): the two-phase decoder for 256 items (matches of one byte: small
    dictionaries) and the chain of groups it falls back to (a cycle of 130 values: distances of two bytes), on frames that are made
    of such blocks -- the oracle's counter says so."""
    from _libs import STAT_LZ, frame_stats, oracle_compress

    for kind in ["dict16", "lzmix", "cycle130"]:
        lz_blocks = 0
        for n in (256 * 5, 256 * 40 + 77):
            data = generate(kind, T, n, 9 + n)
            nb = data.nbytes
            r0, frame = oracle_compress(oracle, data, T, 1)
            st = frame_stats(oracle, frame, T)
            lz_blocks += int(st[STAT_LZ])
            ref = np.zeros(nb * 2 + 4096, dtype=np.uint8)
            r1 = oracle.so_block_compress(np_ptr(data), T, nb, np_ptr(ref), ref.nbytes)
            for regs in (1, 0):
                emul.emul_set_dec_regs(regs)
                for mis in (0, 5):
                    dec = np.zeros(nb + 64, dtype=np.uint8)
                    r3 = emul.emul_block_decompress(np_ptr(ref), r1, T, nb, np_ptr(dec), mis)
                    assert r3 == nb and np.array_equal(dec[:nb], data), (kind, n, regs, mis)
            emul.emul_set_dec_regs(1)
            # cut short and damaged streams: an error or the oracle's answer, never a crash
            for cut in (r1 - 1, r1 // 2, 40):
                dec = np.zeros(nb + 64, dtype=np.uint8)
                r4 = emul.emul_block_decompress(np_ptr(ref), cut, T, nb, np_ptr(dec), 0)
                assert has_error(r4) or r4 == nb, (kind, n, cut)
        assert lz_blocks >= 5 or (kind == "lzmix" and T == 8), (kind, lz_blocks)  # (12-bit values in 8 bytes: too few planes for an attempt)
