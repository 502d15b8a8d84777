// kernels_wide.hip -- bytesoftype 65 .. 65534 (stenos.h:65): the block kernels of kernels.hip with the wave's scratch in
// HBM instead of LDS.
//
// A block is 256*T bytes, its encoded image up to 280*T, its row and plane tables 132*T, the mini-LZ chains (T <= 512)
// another 512*T: beyond bytesoftype 64 that no longer fits the LDS a workgroup can have.  The codec sources
// (block_codec.h, superblock_codec.h, pipeline.h) address their scratch through a plain pointer and 32-bit offsets and
// handle planes 64 at a time, so the same code runs here on a per-workgroup slice of a device buffer the caller
// provides (FrameJob / DecodeArgs::wide_scratch); STENOS_WIDE makes the scratch reads go to the L2 (wavevec.h).  A
// bounded number of workgroups walks the blocks / superblocks, so the buffer stays small.  This is the slow path: the
// reference's own tests stop at bytesoftype 15 (tests/tests_comp_decomp.cpp) and nothing here is tuned.
//
// scan_superblocks, pack_frame and walk_superblocks of kernels.hip do not depend on the bytesoftype and serve both.
#define STENOS_WIDE 1
#include <hip/hip_runtime.h>

#include "kernels.h"

using namespace codec;
using namespace wv;

namespace {

__global__ __launch_bounds__(64) void encode_blocks_wide(const uint8_t* __restrict__ src, uint64_t b_begin, uint64_t b_end, uint64_t nfull, uint32_t tail_bytes, uint32_t T,
							 uint8_t* __restrict__ slots, uint32_t slot_stride, uint32_t* __restrict__ bsize, uint32_t* __restrict__ binfo,
							 uint32_t* __restrict__ bneed, uint8_t* __restrict__ scratch, uint64_t stride)
{
	uint8_t* mine = scratch + blockIdx.x * stride;
	const Layout L = make_layout(T, true);
	for (uint64_t b = b_begin + blockIdx.x; b < b_end; b += gridDim.x) {
		BlockInfo r;
		if (b < nfull)
			r = encode_block_job(mine, L, T, src + b * (uint64_t)(256 * T), slots + b * (uint64_t)slot_stride, true);
		else
			r = encode_tail_job(mine, L, T, src + nfull * (uint64_t)(256 * T), tail_bytes, slots + nfull * (uint64_t)slot_stride);
		if (threadIdx.x == 0) {
			bsize[b] = r.size;
			binfo[b] = r.info;
			bneed[b] = r.need;
		}
	}
}

__global__ __launch_bounds__(64) void plan_superblocks_wide(FrameJob j, uint64_t s_begin, uint64_t s_end, uint8_t* __restrict__ scratch, uint64_t stride)
{
	const Layout L = make_layout(j.T, true);
	for (uint64_t s = s_begin + blockIdx.x; s < s_end; s += gridDim.x)
		plan_superblock(scratch + blockIdx.x * stride, L, j, s);
}

__global__ __launch_bounds__(64) void resolve_frame_wide(FrameJob j, uint8_t* __restrict__ scratch)
{
	const Layout L = make_layout(j.T, true);
	resolve_capacity(scratch, L, j);
}

__global__ __launch_bounds__(64) void decode_superblocks_wide(DecodeArgs a, uint8_t* __restrict__ scratch, uint64_t stride)
{
	const uint32_t T = a.T;
	for (uint64_t e = blockIdx.x; e < a.nsb; e += gridDim.x) {
		const uint64_t s = a.sb_ids ? a.sb_ids[e] : e;
		const uint64_t p = a.sb_off[e];
		if (p > a.size || a.size - p < 4) { // (written without sums: an index entry may hold anything)
			if (threadIdx.x == 0)
				atomicOr(a.status, DECODE_STATUS_TRUNCATED);
			continue;
		}
		const uint32_t code = a.frame[p];
		const uint32_t csize = (uint32_t)a.frame[p + 1] | ((uint32_t)a.frame[p + 2] << 8) | ((uint32_t)a.frame[p + 3] << 16);
		const uint64_t begin = s * (uint64_t)a.sb_bytes;
		const uint32_t dsize = (uint32_t)((a.total_bytes - begin) < a.sb_bytes ? (a.total_bytes - begin) : a.sb_bytes);
		if (a.size - p - 4 < csize) { // stenos.cpp:1133-1134
			if (threadIdx.x == 0)
				atomicOr(a.status, DECODE_STATUS_TRUNCATED);
			continue;
		}
		const uint8_t* payload = a.frame + p + 4;
		uint8_t* out = a.dst + begin;
		if (code == 1) {
			const DecLayout L = make_dec_layout(T);
			if (decode_superblock(scratch + blockIdx.x * stride, L, T, payload, csize, out, dsize) == DEC_ERROR && threadIdx.x == 0)
				atomicOr(a.status, DECODE_STATUS_INVALID);
		}
		else if (code == 6) { // stenos.cpp:741-746
			if (csize != dsize) {
				if (threadIdx.x == 0)
					atomicOr(a.status, DECODE_STATUS_INVALID);
			}
			else
				copy_g2g_wide(out, payload, csize);
		}
		else if (threadIdx.x == 0)
			atomicOr(a.status, code >= 2 && code <= 5 ? DECODE_STATUS_HOST_CODES : DECODE_STATUS_INVALID); // zstd based codes are finished by the host
	}
}

// workgroups that share `bytes` of scratch
uint32_t groups_for(uint64_t units, uint64_t bytes, uint64_t stride)
{
	uint64_t g = stride ? bytes / stride : 0;
	g = g > units ? units : g;
	g = g > 4096 ? 4096 : g;
	return (uint32_t)g;
}

} // namespace

size_t stenos_kw_scratch_stride(uint32_t T)
{
	const size_t e = make_layout(T, true).total, d = make_dec_layout(T).total;
	return ((e > d ? e : d) + 255) & ~(size_t)255;
}

hipError_t stenos_kw_launch_encode(const FrameJob& j, uint64_t b_begin, uint64_t b_end, hipStream_t stream)
{
	const uint64_t stride = stenos_kw_scratch_stride(j.T);
	const uint32_t grid = groups_for(b_end - b_begin, j.wide_scratch_bytes, stride);
	if (!grid)
		return hipErrorInvalidValue;
	hipLaunchKernelGGL(encode_blocks_wide, dim3(grid), dim3(64), 0, stream, j.src, b_begin, b_end, j.nfull, j.tail_bytes, j.T, j.slots, j.slot_stride, j.bsize, j.binfo,
			   j.bneed, j.wide_scratch, stride);
	return hipGetLastError();
}

hipError_t stenos_kw_launch_plan(const FrameJob& j, uint64_t s_begin, uint64_t s_end, hipStream_t stream)
{
	const uint64_t stride = stenos_kw_scratch_stride(j.T);
	const uint32_t grid = groups_for(s_end - s_begin, j.wide_scratch_bytes, stride);
	if (!grid)
		return hipErrorInvalidValue;
	hipLaunchKernelGGL(plan_superblocks_wide, dim3(grid), dim3(64), 0, stream, j, s_begin, s_end, j.wide_scratch, stride);
	return hipGetLastError();
}

hipError_t stenos_kw_launch_resolve(const FrameJob& j, hipStream_t stream)
{
	if (j.wide_scratch_bytes < stenos_kw_scratch_stride(j.T))
		return hipErrorInvalidValue;
	hipLaunchKernelGGL(resolve_frame_wide, dim3(1), dim3(64), 0, stream, j, j.wide_scratch);
	return hipGetLastError();
}

hipError_t stenos_kw_launch_decode(const DecodeArgs& a, hipStream_t stream)
{
	const uint64_t stride = stenos_kw_scratch_stride(a.T);
	const uint32_t grid = groups_for(a.nsb, a.wide_scratch_bytes, stride);
	if (!grid)
		return hipErrorInvalidValue;
	hipLaunchKernelGGL(decode_superblocks_wide, dim3(grid), dim3(64), 0, stream, a, a.wide_scratch, stride);
	return hipGetLastError();
}
