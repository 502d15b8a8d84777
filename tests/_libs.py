"""ctypes loaders used by the tests: the C oracle, the compiled reference (when present) and the
product library.  The oracle and the reference are CHECKERS only; the product never loads them."""
from __future__ import annotations

import ctypes
import os
import subprocess
from ctypes import c_int, c_size_t, c_void_p

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ERR_BASE = (1 << 64) - 100

STAT_PLANE_TYPE, STAT_ROW_HDR, STAT_LZ, STAT_PARTIAL, STAT_SB_CODE, STAT_COPY_BLOCKS, STAT_COUNT = 0, 4, 20, 21, 22, 30, 32


def has_error(r: int) -> bool:
    return r >= ERR_BASE


def load_oracle() -> ctypes.CDLL:
    so = os.path.join(ORACLE_DIR, "libstenos_oracle.so")
    src = os.path.join(ORACLE_DIR, "stenos_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "oracle"], stdout=subprocess.DEVNULL)
    lib = ctypes.CDLL(so)
    lib.so_bound.restype = c_size_t
    lib.so_bound.argtypes = [c_size_t]
    lib.so_compress.restype = c_size_t
    lib.so_compress.argtypes = [c_void_p, c_size_t, c_size_t, c_void_p, c_size_t, c_int]
    lib.so_decompress.restype = c_size_t
    lib.so_decompress.argtypes = [c_void_p, c_size_t, c_size_t, c_void_p, c_size_t, c_int]
    lib.so_block_compress.restype = c_size_t
    lib.so_block_compress.argtypes = [c_void_p, c_size_t, c_size_t, c_void_p, c_size_t]
    lib.so_block_decompress.restype = c_size_t
    lib.so_block_decompress.argtypes = [c_void_p, c_size_t, c_size_t, c_size_t, c_void_p]
    lib.so_encode_block.restype = c_size_t
    lib.so_encode_block.argtypes = [c_void_p, c_size_t, c_void_p, c_int]
    lib.so_frame_stats.restype = c_size_t
    lib.so_frame_stats.argtypes = [c_void_p, c_size_t, c_size_t, c_void_p]
    lib.so_superblock_size.restype = c_size_t
    lib.so_superblock_size.argtypes = [c_size_t, c_size_t, c_int]
    for f in (lib.so_shuffle, lib.so_unshuffle):
        f.restype = None
        f.argtypes = [c_size_t, c_size_t, c_void_p, c_void_p]
    for f in (lib.so_delta, lib.so_delta_inv):
        f.restype = None
        f.argtypes = [c_void_p, c_void_p, c_size_t]
    return lib


def bind_stenos_abi(lib: ctypes.CDLL) -> ctypes.CDLL:
    """argtypes/restypes of the frozen C ABI (reference stenos/stenos.h:115-301)."""
    lib.stenos_bound.restype = c_size_t
    lib.stenos_bound.argtypes = [c_size_t]
    lib.stenos_has_error.restype = c_int
    lib.stenos_has_error.argtypes = [c_size_t]
    lib.stenos_compress.restype = c_size_t
    lib.stenos_compress.argtypes = [c_void_p, c_size_t, c_size_t, c_void_p, c_size_t, c_int]
    lib.stenos_decompress.restype = c_size_t
    lib.stenos_decompress.argtypes = [c_void_p, c_size_t, c_size_t, c_void_p, c_size_t]
    lib.stenos_make_context.restype = c_void_p
    lib.stenos_make_context.argtypes = []
    lib.stenos_destroy_context.restype = None
    lib.stenos_destroy_context.argtypes = [c_void_p]
    lib.stenos_reset_context.restype = None
    lib.stenos_reset_context.argtypes = [c_void_p]
    lib.stenos_set_level.restype = c_size_t
    lib.stenos_set_level.argtypes = [c_void_p, c_int]
    lib.stenos_set_threads.restype = c_size_t
    lib.stenos_set_threads.argtypes = [c_void_p, c_int]
    lib.stenos_set_max_nanoseconds.restype = c_size_t
    lib.stenos_set_max_nanoseconds.argtypes = [c_void_p, ctypes.c_uint64]
    lib.stenos_set_block_size.restype = c_size_t
    lib.stenos_set_block_size.argtypes = [c_void_p, c_size_t]
    lib.stenos_memory_footprint.restype = c_size_t
    lib.stenos_memory_footprint.argtypes = [c_void_p]
    lib.stenos_compress_generic.restype = c_size_t
    lib.stenos_compress_generic.argtypes = [c_void_p, c_void_p, c_size_t, c_size_t, c_void_p, c_size_t]
    lib.stenos_decompress_generic.restype = c_size_t
    lib.stenos_decompress_generic.argtypes = [c_void_p, c_void_p, c_size_t, c_size_t, c_void_p, c_size_t]
    lib.stenos_get_info.restype = c_size_t
    lib.stenos_get_info.argtypes = [c_void_p, c_size_t, c_size_t, c_void_p]
    for f in (lib.stenos_private_compress_block, lib.stenos_private_decompress_block):
        f.restype = c_size_t
        f.argtypes = [c_void_p, c_void_p, c_size_t, c_size_t, c_size_t, c_void_p, c_size_t]
    return lib


def load_ref(det: bool = True):
    """The unmodified reference compiled by oracle/Makefile, or None when it was not built."""
    so = os.path.join(ORACLE_DIR, "_ref", "libstenos_ref_det.so" if det else "libstenos_ref.so")
    if not os.path.exists(so):
        return None
    try:
        return bind_stenos_abi(ctypes.CDLL(so))
    except OSError:
        return None


def load_hooks_library():
    """The library the test suite builds for itself with its switches compiled in (tests/hooks/Makefile, -DSTENOS_TEST_HOOKS):
    stenos_hip_test_lanes / _walk / _fused_timeouts exist there and nowhere else."""
    from stenos_amd.api import load_library

    d = os.path.join(ROOT, "tests", "hooks")
    subprocess.check_call(["make", "-C", d], stdout=subprocess.DEVNULL)
    lib = load_library(os.path.join(d, "libstenos_hooks.so"))
    assert lib._stenos_test_hooks
    return lib


def np_ptr(a: np.ndarray) -> int:
    return a.ctypes.data


def oracle_compress(lib, data: np.ndarray, T: int, level: int = 1, dst_size: int | None = None):
    nb = data.nbytes
    cap = lib.so_bound(nb) if dst_size is None else dst_size
    dst = np.zeros(cap + 64, dtype=np.uint8)
    r = lib.so_compress(np_ptr(data), T, nb, np_ptr(dst), cap, level)
    return r, (dst[:r].copy() if not has_error(r) else None)


def ref_compress(lib, data: np.ndarray, T: int, level: int = 1, dst_size: int | None = None):
    nb = data.nbytes
    cap = lib.stenos_bound(nb) if dst_size is None else dst_size
    dst = np.zeros(cap + 64, dtype=np.uint8)
    r = lib.stenos_compress(np_ptr(data), T, nb, np_ptr(dst), cap, level)
    return r, (dst[:r].copy() if not has_error(r) else None)


def frame_stats(lib, frame: np.ndarray, T: int) -> np.ndarray:
    counts = np.zeros(STAT_COUNT, dtype=np.uint64)
    r = lib.so_frame_stats(np_ptr(frame), T, frame.nbytes, np_ptr(counts))
    assert not has_error(r), r
    return counts
