/*
 * stenos_hip.h -- device-resident entry points of libstenos.so (additions next to the frozen ABI of
 * stenos.h; the reference has no counterpart: its hot loop stenos_compress_generic,
 * stenos/internal/stenos.cpp:884-1010, and stenos_decompress_generic, :1118-1202, only take host
 * pointers).  Source and destination are DEVICE pointers; nothing crosses PCIe except the 8-byte
 * result.  Plain C: pointers and sizes only, `stream` is a hipStream_t passed as void* (NULL = the
 * default stream).
 */
#ifndef STENOS_HIP_H
#define STENOS_HIP_H
#include "stenos.h"

#ifdef __cplusplus
extern "C" {
#endif

/* number of HIP devices visible, 0 when the runtime is unusable */
STENOS_EXPORT int stenos_hip_device_count(void);

/* Bytes of device workspace the two calls below need for `bytes` of input (scratch slots, size and
 * offset tables) when the destination holds stenos_bound(bytes).  With a smaller destination every block goes through
 * a scratch slot instead of the fused kernel's staging buffers and the figure is up to about 1.03 x bytes higher.
 * The context allocates and keeps it; this is for capacity planning. */
STENOS_EXPORT size_t stenos_hip_workspace_bytes(size_t bytesoftype, size_t bytes);

/* Compress `bytes` of device memory into a Stenos frame in device memory.  The bytes of d_dst behind the returned frame size
 * (and inside dst_size) are scratch: incompressible stretches are put in place before their offsets are final
 * (csrc/kernels.hip, speculative copy), so that range may hold leftovers.  Nothing is written at or past dst_size, and
 * the host-pointer calls of stenos.h only copy the frame back.  Uses ctx's level and
 * block-size settings; a time limit (stenos_set_max_nanoseconds) is a feature of the host-pointer calls only:
 * with one set these entry points return STENOS_ERROR_INVALID_PARAMETER.  Enqueues on `stream`, then waits for the 8-byte size to come back.
 * Returns the frame size or an error code (test with stenos_has_error). */
STENOS_EXPORT size_t stenos_hip_compress(stenos_context* ctx, const void* d_src, size_t bytesoftype, size_t bytes, void* d_dst, size_t dst_size, void* stream);

/* Same, without waiting: the result is read by stenos_hip_finish().  The frame bytes in d_dst and
 * the index are valid once the stream has executed the enqueued work. */
STENOS_EXPORT size_t stenos_hip_compress_async(stenos_context* ctx, const void* d_src, size_t bytesoftype, size_t bytes, void* d_dst, size_t dst_size, void* stream);
STENOS_EXPORT size_t stenos_hip_finish(stenos_context* ctx);

/* Superblock index of the last compression on ctx: device array of nsb + 1 uint64 byte offsets of the
 * superblock headers inside the frame (the last entry is the frame size).  Valid until the next call on ctx. */
STENOS_EXPORT const uint64_t* stenos_hip_last_index(stenos_context* ctx, size_t* nsb);

/* Superblock index of any frame held in device memory: the chain of [code][csize:3] headers (reference
 * stenos.cpp:1126-1134, 1166-1182) is walked on the device -- by segments of the frame in parallel, with a result that is
 * proven equal to the serial walk's before it is used (csrc/walk.h).  Returns a device array of *nsb + 1 uint64 byte offsets
 * (the last entry is the end of the last superblock), valid until the next call on ctx; NULL for an empty,
 * malformed or truncated frame.  Waits for the walk.  What a multi-GPU decoder cuts the frame with. */
STENOS_EXPORT const uint64_t* stenos_hip_frame_index(stenos_context* ctx, const void* d_src, size_t bytesoftype, size_t bytes, size_t* nsb, void* stream);

/* Decompress a frame held in device memory.  d_index may be NULL: the superblock chain
 * ([code][csize:3] headers, reference stenos.cpp:1129-1134) is then walked on the device first (in parallel,
 * csrc/walk.h: tens of microseconds); passing the index produced by stenos_hip_last_index() skips that walk.
 * Returns the decompressed size or an error code. */
STENOS_EXPORT size_t stenos_hip_decompress(stenos_context* ctx, const void* d_src, size_t bytesoftype, size_t bytes, void* d_dst, size_t dst_size, const uint64_t* d_index, void* stream);
STENOS_EXPORT size_t stenos_hip_decompress_async(stenos_context* ctx, const void* d_src, size_t bytesoftype, size_t bytes, void* d_dst, size_t dst_size, const uint64_t* d_index, void* stream);

/* Whole-buffer byte kernels of the path on device memory (reference stenos/internal/shuffle.h:33,45 and
 * delta.h:34,39): byte transpose of `bytes / bytesoftype` elements and its inverse (leftover bytes copied),
 * byte delta in four quarter streams above 2048 bytes and its inverse.  src and dst must not overlap.
 * Return 0 or an error code. */
STENOS_EXPORT size_t stenos_hip_shuffle(const void* d_src, size_t bytesoftype, size_t bytes, void* d_dst, void* stream);
STENOS_EXPORT size_t stenos_hip_unshuffle(const void* d_src, size_t bytesoftype, size_t bytes, void* d_dst, void* stream);
STENOS_EXPORT size_t stenos_hip_delta(const void* d_src, void* d_dst, size_t bytes, void* stream);
STENOS_EXPORT size_t stenos_hip_delta_inv(const void* d_src, void* d_dst, size_t bytes, void* stream);

/* Host-pointer calls on several devices (the counterpart of the reference's thread dispatcher, stenos.cpp:909-1010,
 * 1151-1202: a host-pointer call is bound by the PCIe link of its device, the codec is an order of magnitude faster).
 * OPT-IN: by default a call stays on the calling thread's current device whatever stenos_set_threads() says -- that knob
 * means CPU threads to an unmodified caller (stenos.h:140), and the other devices of a process are usually some other
 * rank's.  After stenos_hip_set_devices(ctx, n >= 2) (or with STENOS_HIP_DEVICES >= 2 in the environment, read once, for
 * callers that cannot be changed) a stenos_compress_generic / stenos_decompress_generic call of 64 MiB or more at level
 * 0/1 with bytesoftype > 1 spreads over min(n, threads of stenos_set_threads, visible devices) devices, starting with the
 * calling thread's current one: every device takes a contiguous range of superblocks through a context and a host thread
 * of its own; frames are byte-identical to single-device frames.  n <= 1 turns it off again.
 * stenos_hip_last_devices returns how many devices the last host-pointer call on ctx used (1: the single-device path). */
STENOS_EXPORT void stenos_hip_set_devices(stenos_context* ctx, int devices);
STENOS_EXPORT int stenos_hip_last_devices(stenos_context* ctx);

/* Levels >= 2 (and bytesoftype 1) run a strategy layer on the host around the GPU passes (LZ4-dry estimates, zstd).  Wall
 * time per stage in milliseconds, summed over the calls on ctx since the last reset: out[0] GPU block pass + verdicts and
 * samples to the host, [1] estimates, [2] waiting for block streams from the device, [3] zstd, [4] frame layout, [5] waiting
 * for the frame's upload (device destinations), [6] zstd inflate (decompression), [7] device decode of the inflated
 * superblocks.  Transfers that are hidden behind zstd do not show.  Returns the number of stages; reset != 0 clears the sums. */
STENOS_EXPORT int stenos_hip_stage_ms(stenos_context* ctx, double* out, int n, int reset);

/* The fused encoder's waits for frame offsets are bounded; a launch that gives up (never observed) is redone without that
 * kernel instead of failing the call.  Returns how often that has happened on ctx. */
STENOS_EXPORT int stenos_hip_fused_fallbacks(stenos_context* ctx);

#ifdef STENOS_TEST_HOOKS
/* Switches for the test suite.  They exist only in the build the tests make for themselves (tests/hooks/Makefile,
 * -DSTENOS_TEST_HOOKS: tests/hooks/libstenos_hooks.so); libstenos.so neither declares nor exports them.
 * stenos_hip_test_lanes: share_current_device != 0 lets the "devices" of a multi-device call all stand for the current device
 * (one-GPU boxes); fail_lane >= 0 keeps that lane from running, as if its device could not be made current (-1: none).
 * stenos_hip_test_walk: serial != 0 makes frames that come without an index be walked by one lane (the serial walk that the
 * parallel one of walk.h is proven against, and falls back to); returns whether the last parallel walk on ctx fell back to
 * the serial one (1), did not (0), or there was none (-1).
 * stenos_hip_test_fused_timeouts: the next n fused launches are treated as if they had given up waiting. */
STENOS_EXPORT void stenos_hip_test_lanes(stenos_context* ctx, int share_current_device, int fail_lane);
STENOS_EXPORT int stenos_hip_test_walk(stenos_context* ctx, int serial);
STENOS_EXPORT void stenos_hip_test_fused_timeouts(stenos_context* ctx, int n);
#endif

/* Kernel timing for benchmarks: when enabled, HIP events are recorded on the job's stream around the
 * dominant kernel of each direction: which = 0, the encoder (encode_superblocks, the fused kernel; encode_blocks
 * where that one does not apply), which = 1, decode_superblocks.  stenos_hip_kernel_ms returns the elapsed
 * milliseconds of the last such launch, or a negative value when none was recorded; it waits for the end event. */
STENOS_EXPORT void stenos_hip_set_profiling(stenos_context* ctx, int enabled);
STENOS_EXPORT double stenos_hip_kernel_ms(stenos_context* ctx, int which);

#ifdef __cplusplus
}
#endif
#endif
