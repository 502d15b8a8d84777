// kernels.hip -- gfx950 kernels of the level-1 block codec and their launchers.
//
// Encode pipeline (all on one stream, no host round trip):
//   encode_blocks      one wavefront per 256-element block: HBM -> LDS -> encoded image -> 16-byte
//                      aligned slot in the workspace, size to bsize[]            (block_compress.h:1152-1298)
//   plan_superblocks   one wavefront per superblock: payload size, BLOCK vs COPY decision
//                      (stenos.cpp:606-615), offsets of its blocks
//   scan_superblocks   exclusive scan of the superblock sizes -> byte offset of every superblock header
//   pack_frame         one wavefront per block: [code][csize:3] headers, block payloads or raw copy,
//                      frame header (stenos.cpp:862-874)
// Decode pipeline:
//   walk_superblocks   (only without an index) serial walk of the [code][csize:3] chain (stenos.cpp:1129-1134)
//   decode_superblocks one wavefront per superblock (block_compress.h:2088-2175)
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "superblock_codec.h"

using namespace codec;
using namespace wv;

namespace {

extern __shared__ __attribute__((aligned(16))) uint8_t g_lds[];

__global__ __launch_bounds__(64) void encode_blocks(const uint8_t* __restrict__ src, uint64_t nfull, uint32_t tail_bytes,
						    uint32_t T, uint8_t* __restrict__ slots, uint32_t slot_stride, uint32_t* __restrict__ bsize)
{
	const Layout L = make_layout(T, true);
	const uint64_t b = blockIdx.x;
	uint32_t size;
	if (b < nfull)
		size = encode_block_job(g_lds, L, T, src + b * (uint64_t)(256 * T), slots + b * (uint64_t)slot_stride, true);
	else
		size = encode_tail_job(g_lds, L, T, src + nfull * (uint64_t)(256 * T), tail_bytes, slots + nfull * (uint64_t)slot_stride);
	if (threadIdx.x == 0)
		bsize[b] = size;
}

// One wavefront per superblock.  bps = full blocks per full superblock.
__global__ __launch_bounds__(64) void plan_superblocks(SuperblockPlanArgs a)
{
	const uint64_t s = blockIdx.x;
	const uint32_t lane = threadIdx.x;
	const uint64_t first = s * a.bps;
	const uint64_t sb_begin = s * (uint64_t)a.sb_bytes;
	const uint32_t sbytes = (uint32_t)((a.total_bytes - sb_begin) < a.sb_bytes ? (a.total_bytes - sb_begin) : a.sb_bytes);
	// blocks of this superblock: full ones, plus the tail block when this is the last superblock
	uint64_t last = first + a.bps < a.nfull ? first + a.bps : a.nfull;
	uint32_t count = (uint32_t)(last - first);
	if (s == a.nsb - 1 && a.tail_bytes)
		count += 1;
	uint32_t run = 0;
	for (uint32_t o = 0; o < count; o += 64) {
		uint32_t i = o + lane;
		uint32_t sz = (i < count && !a.force_copy) ? a.bsize[first + i] : 0u;
		uint32_t incl = wave_incl_scan(sz);
		if (i < count)
			a.boff[first + i] = run + incl - sz;
		run += readlane(incl, 63);
	}
	if (lane == 0) {
		uint32_t code = 1, csize = run;
		if (run > sbytes || a.force_copy) { // result > bytes -> memcpy (stenos.cpp:609-610); equal is kept
			code = 6;
			csize = sbytes;
		}
		if (s == a.nsb - 1 && a.override_code) { // superblock shorter than 128 bytes: prepared by the host
			code = a.override_code;
			csize = a.override_size;
		}
		a.sb_code[s] = (uint8_t)code;
		a.sb_csize[s] = csize;
	}
}

// Exclusive scan of (csize + 4) over the superblocks by one workgroup of 1024 threads.
__global__ __launch_bounds__(1024) void scan_superblocks(const uint32_t* __restrict__ csize, uint64_t nsb, uint64_t header_bytes,
							 uint64_t* __restrict__ off, uint64_t* __restrict__ total)
{
	__shared__ uint64_t partial[1024];
	const uint32_t tid = threadIdx.x;
	const uint64_t per = (nsb + 1023) / 1024;
	const uint64_t lo = tid * per, hi = lo + per < nsb ? lo + per : nsb;
	uint64_t sum = 0;
	for (uint64_t i = lo; i < hi; ++i)
		sum += (uint64_t)csize[i] + 4;
	partial[tid] = sum;
	__syncthreads();
	for (uint32_t d = 1; d < 1024; d <<= 1) { // Hillis-Steele inclusive scan
		uint64_t v = tid >= d ? partial[tid - d] : 0;
		__syncthreads();
		partial[tid] += v;
		__syncthreads();
	}
	uint64_t base = header_bytes + partial[tid] - sum;
	for (uint64_t i = lo; i < hi; ++i) {
		off[i] = base;
		base += (uint64_t)csize[i] + 4;
	}
	if (tid == 1023) {
		off[nsb] = header_bytes + partial[1023];
		*total = header_bytes + partial[1023];
	}
}

__global__ __launch_bounds__(64) void pack_frame(PackArgs a)
{
	const uint64_t b = blockIdx.x;
	const U32 lane = lane_id();
	const uint64_t total = *a.total;
	if (total > a.dst_size) // never write past the caller's buffer; the host reports DST_OVERFLOW
		return;
	const bool is_tail = b >= a.nfull;
	const uint64_t s = is_tail ? a.nsb - 1 : b / a.bps;
	const uint64_t first = s * a.bps;
	uint8_t* base = a.dst + a.sb_off[s];
	const uint32_t code = a.sb_code[s];
	const uint32_t csize = a.sb_csize[s];
	if (b == 0 && a.shift_byte != 0xFFFFFFFFu) { // frame header: [shift][bytes:7 LE] (+ [superblock size:4 LE] for custom sizes)
		uint64_t v = (uint64_t)a.shift_byte | (a.total_bytes << 8);
		gst8(a.dst, lane, U32((uint32_t)(v >> 0)) >> (lane << 3), lane < U32(4u));
		gst8(a.dst, lane, U32((uint32_t)(v >> 32)) >> ((lane - 4u) << 3), (lane >= U32(4u)) & (lane < U32(8u)));
		if (a.shift_byte == 255)
			gst8(a.dst + 8, lane, U32(a.sb_bytes) >> (lane << 3), lane < U32(4u));
	}
	if (b == first || (is_tail && a.nfull == first)) { // superblock header [code][csize:3 LE]
		uint32_t h = code | (csize << 8);
		gst8(base, lane, U32(h) >> (lane << 3), lane < U32(4u));
	}
	if (code == 1)
		copy_g2g(base + 4 + a.boff[b], a.slots + b * (uint64_t)a.slot_stride, a.bsize[b]);
	else if (code == 6 && !(s == a.nsb - 1 && a.override_code)) {
		const uint32_t bs = 256 * a.T;
		uint32_t n = is_tail ? a.tail_bytes : bs;
		copy_g2g(base + 4 + (uint32_t)(b - first) * (uint64_t)bs, a.src + b * (uint64_t)bs, n);
	}
	else if (is_tail) // payload prepared by the host (superblock < 128 bytes)
		copy_g2g(base + 4, a.override_payload, csize);
}

// Serial walk of the superblock chain by one lane: off[s] = byte offset of superblock s's header.
__global__ void walk_superblocks(const uint8_t* __restrict__ frame, uint64_t size, uint64_t first, uint64_t nsb, uint64_t* __restrict__ off,
				 uint32_t* __restrict__ status)
{
	if (threadIdx.x != 0 || blockIdx.x != 0)
		return;
	uint64_t p = first;
	for (uint64_t s = 0; s < nsb; ++s) {
		if (p + 4 > size) { // stenos.cpp:1126-1127
			atomicOr(status, DECODE_STATUS_TRUNCATED);
			for (; s <= nsb; ++s) off[s] = size;
			return;
		}
		off[s] = p;
		uint32_t csize = (uint32_t)frame[p + 1] | ((uint32_t)frame[p + 2] << 8) | ((uint32_t)frame[p + 3] << 16);
		p += 4 + (uint64_t)csize;
	}
	off[nsb] = p;
	if (p > size)
		atomicOr(status, DECODE_STATUS_TRUNCATED);
}

__global__ __launch_bounds__(64) void decode_superblocks(DecodeArgs a)
{
	const uint64_t s = blockIdx.x;
	const U32 lane = lane_id();
	const uint64_t p = a.sb_off[s];
	if (p + 4 > a.size) {
		if (threadIdx.x == 0)
			atomicOr(a.status, DECODE_STATUS_TRUNCATED);
		return;
	}
	const uint32_t code = a.frame[p];
	const uint32_t csize = (uint32_t)a.frame[p + 1] | ((uint32_t)a.frame[p + 2] << 8) | ((uint32_t)a.frame[p + 3] << 16);
	const uint64_t begin = s * (uint64_t)a.sb_bytes;
	const uint32_t dsize = (uint32_t)((a.total_bytes - begin) < a.sb_bytes ? (a.total_bytes - begin) : a.sb_bytes);
	if (p + 4 + csize > a.size) { // stenos.cpp:1133-1134
		if (threadIdx.x == 0)
			atomicOr(a.status, DECODE_STATUS_TRUNCATED);
		return;
	}
	const uint8_t* payload = a.frame + p + 4;
	uint8_t* out = a.dst + begin;
	if (code == 1) {
		const DecLayout L = make_dec_layout(a.T);
		uint32_t r = decode_superblock(g_lds, L, a.T, payload, csize, out, dsize);
		if (r == DEC_ERROR && threadIdx.x == 0)
			atomicOr(a.status, DECODE_STATUS_INVALID);
	}
	else if (code == 6) { // stenos.cpp:741-746
		if (csize != dsize) {
			if (threadIdx.x == 0)
				atomicOr(a.status, DECODE_STATUS_INVALID);
			return;
		}
		copy_g2g(out, payload, csize);
	}
	else if (code >= 2 && code <= 5) { // zstd based codes are finished by the host
		if (threadIdx.x == 0)
			atomicOr(a.status, DECODE_STATUS_HOST_CODES);
	}
	else if (threadIdx.x == 0)
		atomicOr(a.status, DECODE_STATUS_INVALID);
	(void)lane;
}

} // namespace

// ---------------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------------

size_t stenos_k_encode_lds_bytes(uint32_t T) { return make_layout(T, true).total; }
size_t stenos_k_decode_lds_bytes(uint32_t T) { return make_dec_layout(T).total; }
uint32_t stenos_k_slot_stride(uint32_t T) { return out_capacity(T); }

hipError_t stenos_k_launch_encode(const uint8_t* src, uint64_t nfull, uint32_t tail_bytes, uint32_t T, uint8_t* slots, uint32_t* bsize,
				  hipStream_t stream)
{
	const uint64_t nblocks = nfull + (tail_bytes ? 1 : 0);
	if (nblocks == 0)
		return hipSuccess;
	const size_t lds = stenos_k_encode_lds_bytes(T);
	hipError_t e = hipFuncSetAttribute((const void*)encode_blocks, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
	if (e != hipSuccess)
		return e;
	hipLaunchKernelGGL(encode_blocks, dim3((uint32_t)nblocks), dim3(64), lds, stream, src, nfull, tail_bytes, T, slots, out_capacity(T), bsize);
	return hipGetLastError();
}

hipError_t stenos_k_launch_plan(const SuperblockPlanArgs& a, hipStream_t stream)
{
	hipLaunchKernelGGL(plan_superblocks, dim3((uint32_t)a.nsb), dim3(64), 0, stream, a);
	return hipGetLastError();
}

hipError_t stenos_k_launch_scan(const uint32_t* csize, uint64_t nsb, uint64_t header_bytes, uint64_t* off, uint64_t* total, hipStream_t stream)
{
	hipLaunchKernelGGL(scan_superblocks, dim3(1), dim3(1024), 0, stream, csize, nsb, header_bytes, off, total);
	return hipGetLastError();
}

hipError_t stenos_k_launch_pack(const PackArgs& a, hipStream_t stream)
{
	const uint64_t nblocks = a.nfull + (a.tail_bytes ? 1 : 0);
	hipLaunchKernelGGL(pack_frame, dim3((uint32_t)nblocks), dim3(64), 0, stream, a);
	return hipGetLastError();
}

hipError_t stenos_k_launch_walk(const uint8_t* frame, uint64_t size, uint64_t first, uint64_t nsb, uint64_t* off, uint32_t* status, hipStream_t stream)
{
	hipLaunchKernelGGL(walk_superblocks, dim3(1), dim3(64), 0, stream, frame, size, first, nsb, off, status);
	return hipGetLastError();
}

hipError_t stenos_k_launch_decode(const DecodeArgs& a, hipStream_t stream)
{
	const size_t lds = stenos_k_decode_lds_bytes(a.T);
	hipError_t e = hipFuncSetAttribute((const void*)decode_superblocks, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
	if (e != hipSuccess)
		return e;
	hipLaunchKernelGGL(decode_superblocks, dim3((uint32_t)a.nsb), dim3(64), lds, stream, a);
	return hipGetLastError();
}
