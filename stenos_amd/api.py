"""ctypes binding of libstenos.so: the frozen C ABI (include/stenos.h, mirroring the reference's
stenos/stenos.h:115-301) plus the device-pointer entry points (include/stenos_hip.h)."""
from __future__ import annotations

import ctypes
import os
import subprocess
from ctypes import c_int, c_size_t, c_uint64, c_void_p

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("STENOS_LIB_PATH") or os.path.join(HERE, "lib", "libstenos.so")  # override: experiments with other builds
ERR_BASE = (1 << 64) - 100

ERROR_NAMES = {
    (1 << 64) - 1: "STENOS_ERROR_UNDEFINED",
    (1 << 64) - 2: "STENOS_ERROR_SRC_OVERFLOW",
    (1 << 64) - 3: "STENOS_ERROR_ALLOC",
    (1 << 64) - 4: "STENOS_ERROR_INVALID_INPUT",
    (1 << 64) - 5: "STENOS_ERROR_INVALID_INSTRUCTION_SET",
    (1 << 64) - 6: "STENOS_ERROR_DST_OVERFLOW",
    (1 << 64) - 7: "STENOS_ERROR_INVALID_BYTESOFTYPE",
    (1 << 64) - 8: "STENOS_ERROR_ZSTD_INTERNAL",
    (1 << 64) - 9: "STENOS_ERROR_INVALID_PARAMETER",
}


class StenosError(RuntimeError):
    def __init__(self, code: int):
        super().__init__(ERROR_NAMES.get(code, f"stenos error {code:#x}"))
        self.code = code


def build_library() -> str:
    """Compile libstenos.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    subprocess.check_call(["make", "-C", os.path.join(HERE, "csrc")])
    return LIB_PATH


def load_library(path: str = LIB_PATH) -> ctypes.CDLL:
    if not os.path.exists(path):
        raise ImportError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` (no CPU fallback exists)")
    # PyTorch-ROCm ships its own libamdhip64 and asks for it under a name the loader does not match with an
    # already loaded /opt/rocm copy; two HIP runtimes in one process cannot both open the GPU.  Loading torch
    # first makes libstenos.so resolve to the copy torch uses.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = ctypes.CDLL(path)
    sz, vp = c_size_t, c_void_p
    sigs = {
        "stenos_make_context": (vp, []),
        "stenos_destroy_context": (None, [vp]),
        "stenos_reset_context": (None, [vp]),
        "stenos_set_level": (sz, [vp, c_int]),
        "stenos_set_threads": (sz, [vp, c_int]),
        "stenos_set_max_nanoseconds": (sz, [vp, c_uint64]),
        "stenos_set_block_size": (sz, [vp, sz]),
        "stenos_memory_footprint": (sz, [vp]),
        "stenos_has_error": (c_int, [sz]),
        "stenos_bound": (sz, [sz]),
        "stenos_compress_generic": (sz, [vp, vp, sz, sz, vp, sz]),
        "stenos_decompress_generic": (sz, [vp, vp, sz, sz, vp, sz]),
        "stenos_compress": (sz, [vp, sz, sz, vp, sz, c_int]),
        "stenos_decompress": (sz, [vp, sz, sz, vp, sz]),
        "stenos_get_info": (sz, [vp, sz, sz, vp]),
        "stenos_make_timer": (vp, []),
        "stenos_destroy_timer": (None, [vp]),
        "stenos_tick": (None, [vp]),
        "stenos_tock": (c_uint64, [vp]),
        "stenos_private_compress_block": (sz, [vp, vp, sz, sz, sz, vp, sz]),
        "stenos_private_decompress_block": (sz, [vp, vp, sz, sz, sz, vp, sz]),
        "stenos_private_block_size": (sz, [vp, sz]),
        "stenos_private_block_csize": (sz, [vp]),
        "stenos_private_create_compression_header": (sz, [sz, sz, vp, sz]),
        "stenos_hip_device_count": (c_int, []),
        "stenos_hip_last_devices": (c_int, [vp]),
        "stenos_hip_set_devices": (None, [vp, c_int]),
        "stenos_hip_stage_ms": (c_int, [vp, ctypes.POINTER(ctypes.c_double), c_int, c_int]),
        "stenos_hip_fused_fallbacks": (c_int, [vp]),
        "stenos_hip_workspace_bytes": (sz, [sz, sz]),
        "stenos_hip_compress": (sz, [vp, vp, sz, sz, vp, sz, vp]),
        "stenos_hip_compress_async": (sz, [vp, vp, sz, sz, vp, sz, vp]),
        "stenos_hip_finish": (sz, [vp]),
        "stenos_hip_last_index": (vp, [vp, ctypes.POINTER(sz)]),
        "stenos_hip_frame_index": (vp, [vp, vp, sz, sz, ctypes.POINTER(sz), vp]),
        "stenos_hip_decompress": (sz, [vp, vp, sz, sz, vp, sz, vp, vp]),
        "stenos_hip_decompress_async": (sz, [vp, vp, sz, sz, vp, sz, vp, vp]),
        "stenos_hip_shuffle": (sz, [vp, sz, sz, vp, vp]),
        "stenos_hip_unshuffle": (sz, [vp, sz, sz, vp, vp]),
        "stenos_hip_delta": (sz, [vp, vp, sz, vp]),
        "stenos_hip_delta_inv": (sz, [vp, vp, sz, vp]),
        "stenos_hip_set_profiling": (None, [vp, c_int]),
        "stenos_hip_kernel_ms": (ctypes.c_double, [vp, c_int]),
    }
    for name, (res, args) in sigs.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export the symbol
        fn.restype = res
        fn.argtypes = args
    lib._stenos_symbols = tuple(sigs)
    # the test suite's own build (tests/hooks/Makefile, -DSTENOS_TEST_HOOKS) has three switches more; libstenos.so has none
    hooks = {"stenos_hip_test_lanes": (None, [vp, c_int, c_int]), "stenos_hip_test_walk": (c_int, [vp, c_int]), "stenos_hip_test_fused_timeouts": (None, [vp, c_int])}
    lib._stenos_test_hooks = all(hasattr(lib, name) for name in hooks)
    if lib._stenos_test_hooks:
        for name, (res, args) in hooks.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
    return lib


class Stenos:
    """A compression context on torch CUDA tensors (device-resident path) -- plumbing for tests and bench."""

    def __init__(self, level: int = 1, lib: ctypes.CDLL | None = None):
        self.lib = lib or load_library()
        self.ctx = self.lib.stenos_make_context()
        if not self.ctx:
            raise MemoryError("stenos_make_context failed")
        self.lib.stenos_set_level(self.ctx, level)

    def close(self):
        if self.ctx:
            self.lib.stenos_destroy_context(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _check(r: int) -> int:
        if r >= ERR_BASE:
            raise StenosError(r)
        return r

    def bound(self, nbytes: int) -> int:
        return self.lib.stenos_bound(nbytes)

    @staticmethod
    def _stream_ptr():
        import torch

        return torch.cuda.current_stream().cuda_stream

    def compress(self, src, bytesoftype: int, dst, wait: bool = True) -> int:
        """src, dst: contiguous uint8 CUDA tensors.  Returns the frame size (0 when wait=False)."""
        fn = self.lib.stenos_hip_compress if wait else self.lib.stenos_hip_compress_async
        return self._check(fn(self.ctx, src.data_ptr(), bytesoftype, src.numel(), dst.data_ptr(), dst.numel(), self._stream_ptr()))

    def finish(self) -> int:
        return self._check(self.lib.stenos_hip_finish(self.ctx))

    def set_profiling(self, enabled: bool = True):
        self.lib.stenos_hip_set_profiling(self.ctx, 1 if enabled else 0)

    def kernel_ms(self, which: int) -> float:
        """elapsed ms of the last encode_blocks (0) / decode_superblocks (1) launch"""
        return self.lib.stenos_hip_kernel_ms(self.ctx, which)

    def stage_ms(self, reset: bool = True):
        """wall milliseconds per stage of the levels >= 2 strategy layer since the last reset (include/stenos_hip.h)"""
        names = ("gpu_pass", "estimates", "wait_block_streams", "zstd", "layout", "wait_upload", "inflate", "device_decode")
        buf = (ctypes.c_double * len(names))()
        self.lib.stenos_hip_stage_ms(self.ctx, buf, len(names), 1 if reset else 0)
        return {k: round(buf[i], 3) for i, k in enumerate(names)}

    def last_index(self):
        n = c_size_t(0)
        p = self.lib.stenos_hip_last_index(self.ctx, ctypes.byref(n))
        return p, n.value

    def frame_index(self, frame, bytesoftype: int, csize: int):
        """Offsets of the superblock headers of any frame on the device (nsb + 1 of them, the last one is the end) as a
        list of ints; raises for a malformed or truncated frame."""
        import torch

        n = c_size_t(0)
        p = self.lib.stenos_hip_frame_index(self.ctx, frame.data_ptr(), bytesoftype, csize, ctypes.byref(n), self._stream_ptr())
        if not p:
            raise StenosError((1 << 64) - 4)
        host = torch.empty(n.value + 1, dtype=torch.int64)
        hip = ctypes.CDLL("libamdhip64.so")
        if hip.hipMemcpy(c_void_p(host.data_ptr()), c_void_p(p), c_size_t((n.value + 1) * 8), 2) != 0:
            raise StenosError((1 << 64) - 1)
        return host.tolist()

    def decompress(self, frame, bytesoftype: int, csize: int, dst, index_ptr: int | None = None, wait: bool = True) -> int:
        fn = self.lib.stenos_hip_decompress if wait else self.lib.stenos_hip_decompress_async
        return self._check(fn(self.ctx, frame.data_ptr(), bytesoftype, csize, dst.data_ptr(), dst.numel(), index_ptr, self._stream_ptr()))
