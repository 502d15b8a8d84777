// kernels.h -- host-visible interface of kernels.hip (argument blocks and launchers).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

enum : uint32_t {
	DECODE_STATUS_TRUNCATED = 1,  // a header or payload runs past the end of the frame -> STENOS_ERROR_SRC_OVERFLOW / INVALID_INPUT
	DECODE_STATUS_INVALID = 2,    // malformed block stream or unknown code -> STENOS_ERROR_INVALID_INPUT
	DECODE_STATUS_HOST_CODES = 4, // superblocks with zstd-based codes 2..5 are present (finished by the host)
};

struct SuperblockPlanArgs {
	const uint32_t* bsize; // encoded size of every block (tail block last)
	uint32_t* boff;        // out: offset of every block inside its superblock payload
	uint32_t* sb_csize;    // out: payload size per superblock
	uint8_t* sb_code;      // out: 1 BLOCK / 6 COPY (/ override)
	uint64_t nfull;        // full blocks in the input
	uint64_t nsb;          // superblocks
	uint64_t total_bytes;  // input bytes
	uint32_t tail_bytes;   // bytes of the trailing partial block (0: none)
	uint32_t bps;          // full blocks per full superblock
	uint32_t sb_bytes;     // superblock size in bytes
	uint32_t override_code; // non-zero: code of the last superblock, prepared by the host (< 128 bytes -> zstd, stenos.cpp:435-437)
	uint32_t override_size;
	uint32_t force_copy;    // level 0: every superblock is a copy (stenos.cpp:431-433)
};

struct PackArgs {
	const uint8_t* src;
	uint8_t* dst;
	uint64_t dst_size;
	const uint8_t* slots;
	const uint32_t* bsize;
	const uint32_t* boff;
	const uint32_t* sb_csize;
	const uint8_t* sb_code;
	const uint64_t* sb_off;
	const uint64_t* total;
	const uint8_t* override_payload;
	uint64_t nfull;
	uint64_t nsb;
	uint64_t total_bytes;
	uint32_t tail_bytes;
	uint32_t bps;
	uint32_t sb_bytes;
	uint32_t slot_stride;
	uint32_t T;
	uint32_t shift_byte;
	uint32_t override_code;
};

struct DecodeArgs {
	const uint8_t* frame;
	uint64_t size; // frame bytes
	const uint64_t* sb_off;
	uint8_t* dst;
	uint64_t total_bytes;
	uint64_t nsb;
	uint32_t sb_bytes;
	uint32_t T;
	uint32_t* status;
};

size_t stenos_k_encode_lds_bytes(uint32_t T);
size_t stenos_k_decode_lds_bytes(uint32_t T);
uint32_t stenos_k_slot_stride(uint32_t T);

hipError_t stenos_k_launch_encode(const uint8_t* src, uint64_t nfull, uint32_t tail_bytes, uint32_t T, uint8_t* slots, uint32_t* bsize, hipStream_t stream);
hipError_t stenos_k_launch_plan(const SuperblockPlanArgs& a, hipStream_t stream);
hipError_t stenos_k_launch_scan(const uint32_t* csize, uint64_t nsb, uint64_t header_bytes, uint64_t* off, uint64_t* total, hipStream_t stream);
hipError_t stenos_k_launch_pack(const PackArgs& a, hipStream_t stream);
hipError_t stenos_k_launch_walk(const uint8_t* frame, uint64_t size, uint64_t first, uint64_t nsb, uint64_t* off, uint32_t* status, hipStream_t stream);
hipError_t stenos_k_launch_decode(const DecodeArgs& a, hipStream_t stream);
