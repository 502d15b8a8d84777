// kernels.h -- host-visible interface of kernels.hip (argument blocks and launchers).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "pipeline.h" // FrameJob (plain pointers and sizes)

enum : uint32_t {
	DECODE_STATUS_TRUNCATED = 1,  // a header or payload runs past the end of the frame -> STENOS_ERROR_SRC_OVERFLOW / INVALID_INPUT
	DECODE_STATUS_INVALID = 2,    // malformed block stream or unknown code -> STENOS_ERROR_INVALID_INPUT
	DECODE_STATUS_HOST_CODES = 4, // superblocks with zstd-based codes 2..5 are present (finished by the host)
};

struct DecodeArgs {
	const uint8_t* frame;
	uint64_t size; // frame bytes
	const uint64_t* sb_off;
	const uint32_t* sb_ids = nullptr; // superblock numbers of the launch's entries; NULL: 0, 1, 2, ...
	uint8_t* dst;
	uint64_t total_bytes;
	uint64_t nsb;
	uint32_t sb_bytes;
	uint32_t T;
	uint32_t* status;
	uint8_t* wide_scratch = nullptr; // bytesoftype above 64 (kernels_wide.hip), as in FrameJob
	uint64_t wide_scratch_bytes = 0;
};

uint32_t stenos_k_cu_count();
size_t stenos_k_encode_lds_bytes(uint32_t T);
size_t stenos_k_decode_lds_bytes(uint32_t T);
uint32_t stenos_k_slot_stride(uint32_t T);

hipError_t stenos_k_launch_encode(const codec::FrameJob& j, uint64_t b_begin, uint64_t b_end, hipStream_t stream);
hipError_t stenos_k_launch_plan(const codec::FrameJob& j, uint64_t s_begin, uint64_t s_end, hipStream_t stream);
hipError_t stenos_k_launch_scan(const codec::FrameJob& j, uint64_t s_begin, uint64_t s_end, uint64_t* carry, hipStream_t stream);
hipError_t stenos_k_launch_resolve(const codec::FrameJob& j, hipStream_t stream);
hipError_t stenos_k_launch_keep_superblocks(const uint8_t* keep, uint8_t* code, uint32_t* csize, uint64_t nsb, hipStream_t stream);
hipError_t stenos_k_launch_pack(const codec::FrameJob& j, uint64_t s_begin, uint64_t s_end, hipStream_t stream);
// walk_kernels.hip: off[0 .. nsb] of a frame without an index.  scratch: stenos_k_walk_scratch_bytes() of device memory for
// the parallel walk (walk.h); NULL: the serial walk by one lane.
hipError_t stenos_k_launch_walk(const uint8_t* frame, uint64_t size, uint64_t first, uint64_t nsb, uint32_t sb_bytes, uint64_t* off, uint32_t* status, void* scratch,
				hipStream_t stream);
size_t stenos_k_walk_scratch_bytes();
hipError_t stenos_k_launch_decode(const DecodeArgs& a, hipStream_t stream);

// kernels_wide.hip: bytesoftype above codec::MAX_T of the LDS-resident kernels (the launchers above forward to these)
constexpr uint32_t STENOS_K_LDS_MAX_T = 64;
size_t stenos_kw_scratch_stride(uint32_t T);
hipError_t stenos_kw_launch_encode(const codec::FrameJob& j, uint64_t b_begin, uint64_t b_end, hipStream_t stream);
hipError_t stenos_kw_launch_plan(const codec::FrameJob& j, uint64_t s_begin, uint64_t s_end, hipStream_t stream);
hipError_t stenos_kw_launch_resolve(const codec::FrameJob& j, hipStream_t stream);
hipError_t stenos_kw_launch_decode(const DecodeArgs& a, hipStream_t stream);

// byte_kernels.hip
hipError_t stenos_k_launch_shuffle(const uint8_t* src, uint8_t* dst, uint32_t T, uint64_t bytes, bool inverse, hipStream_t stream);
hipError_t stenos_k_launch_init(uint8_t* misc, uint64_t first_off, uint64_t* z1, uint64_t n1, uint64_t* z2, uint64_t n2, hipStream_t stream);
hipError_t stenos_k_launch_encode_fused(const codec::FrameJob& j, uint64_t nsb, uint8_t* stage, uint64_t* desc, uint32_t* ticket, uint64_t* carry, hipStream_t stream);
bool stenos_k_fused_supported(uint32_t T);
uint32_t stenos_k_fused_groups(uint64_t nsb, uint32_t T);
size_t stenos_k_fused_stage_bytes(uint32_t T, uint32_t bps, uint64_t nsb);
hipError_t stenos_k_launch_delta(const uint8_t* src, uint8_t* dst, uint64_t bytes, bool inverse, hipStream_t stream);
hipError_t stenos_k_launch_gather_pieces(const uint8_t* src, uint64_t stride, const uint64_t* off, const uint64_t* size, uint32_t count, uint8_t* dst, hipStream_t stream);
hipError_t stenos_k_launch_shuffle_superblocks(const uint8_t* src, uint8_t* dst, uint32_t T, uint64_t sb, uint64_t total, hipStream_t stream);
hipError_t stenos_k_launch_delta_middles(const uint8_t* shuffled, uint8_t* out, uint32_t T, uint64_t sb, uint64_t total, uint32_t level, bool with_delta,
					 hipStream_t stream);
