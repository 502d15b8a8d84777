// byte_kernels.hip -- whole-buffer byte kernels of the Stenos path, HBM-bound, no LDS reuse beyond a tile:
//   shuffle / unshuffle   dest[j*N + i] = src[i*T + j] and back, leftover bytes copied
//                         (reference stenos/internal/shuffle.cpp:82-103, semantics shuffle-generic.h:33-125)
//   delta / delta_inv     byte delta in one stream up to 2048 bytes, else four quarter streams + tail
//                         (reference stenos/internal/delta.cpp:30-71, 230-268)
// They serve the strategy codes 3 (TRANSPOSED_ZSTD) and 4 (TRANSPOSED_DELTA_ZSTD) of
// decompress_generic_superblock (stenos.cpp:700-725) and, later, the level >= 2 producers (:513, 646).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

namespace {

constexpr uint32_t TILE_THREADS = 256;

// tile of E elements per workgroup, E a multiple of 4 * TILE_THREADS... chosen by the launcher so that E*T <= 48 KiB
extern __shared__ __attribute__((aligned(16))) uint8_t tile[];

__global__ __launch_bounds__(TILE_THREADS) void shuffle_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, uint32_t T, uint64_t n,
								uint32_t E, uint64_t bytes)
{
	const uint64_t e0 = (uint64_t)blockIdx.x * E;
	const uint32_t cnt = (uint32_t)((n - e0) < E ? (n - e0) : E); // elements of this tile
	const uint32_t tb = cnt * T;
	const uint8_t* s = src + e0 * T;
	// coalesced load of the element-major tile
	if ((((uintptr_t)s) & 15u) == 0) {
		for (uint32_t o = threadIdx.x * 16; o + 16 <= tb; o += TILE_THREADS * 16)
			*(uint4*)(tile + o) = *(const uint4*)(s + o);
		for (uint32_t o = (tb & ~15u) + threadIdx.x; o < tb; o += TILE_THREADS)
			tile[o] = s[o];
	}
	else
		for (uint32_t o = threadIdx.x; o < tb; o += TILE_THREADS)
			tile[o] = s[o];
	__syncthreads();
	// plane-major stores: thread handles 4 consecutive elements of one plane
	for (uint32_t j = 0; j < T; ++j) {
		uint8_t* d = dst + (uint64_t)j * n + e0;
		const bool aligned = (((uintptr_t)d) & 3u) == 0;
		for (uint32_t e = threadIdx.x * 4; e < cnt; e += TILE_THREADS * 4) {
			if (e + 4 <= cnt && aligned) {
				uint32_t w = tile[e * T + j] | (tile[(e + 1) * T + j] << 8) | (tile[(e + 2) * T + j] << 16) | ((uint32_t)tile[(e + 3) * T + j] << 24);
				*(uint32_t*)(d + e) = w;
			}
			else
				for (uint32_t k = 0; k < 4 && e + k < cnt; ++k)
					d[e + k] = tile[(e + k) * T + j];
		}
	}
	// leftover bytes (bytes % T) are copied verbatim by the first workgroup
	if (blockIdx.x == 0) {
		const uint64_t rem = bytes - n * T;
		for (uint64_t o = threadIdx.x; o < rem; o += TILE_THREADS)
			dst[n * T + o] = src[n * T + o];
	}
}

__global__ __launch_bounds__(TILE_THREADS) void unshuffle_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, uint32_t T, uint64_t n,
								  uint32_t E, uint64_t bytes)
{
	const uint64_t e0 = (uint64_t)blockIdx.x * E;
	const uint32_t cnt = (uint32_t)((n - e0) < E ? (n - e0) : E);
	const uint32_t tb = cnt * T;
	for (uint32_t j = 0; j < T; ++j) {
		const uint8_t* s = src + (uint64_t)j * n + e0;
		const bool aligned = (((uintptr_t)s) & 3u) == 0;
		for (uint32_t e = threadIdx.x * 4; e < cnt; e += TILE_THREADS * 4) {
			if (e + 4 <= cnt && aligned) {
				uint32_t w = *(const uint32_t*)(s + e);
				tile[e * T + j] = (uint8_t)w;
				tile[(e + 1) * T + j] = (uint8_t)(w >> 8);
				tile[(e + 2) * T + j] = (uint8_t)(w >> 16);
				tile[(e + 3) * T + j] = (uint8_t)(w >> 24);
			}
			else
				for (uint32_t k = 0; k < 4 && e + k < cnt; ++k)
					tile[(e + k) * T + j] = s[e + k];
		}
	}
	__syncthreads();
	uint8_t* d = dst + e0 * T;
	if ((((uintptr_t)d) & 15u) == 0) {
		for (uint32_t o = threadIdx.x * 16; o + 16 <= tb; o += TILE_THREADS * 16)
			*(uint4*)(d + o) = *(const uint4*)(tile + o);
		for (uint32_t o = (tb & ~15u) + threadIdx.x; o < tb; o += TILE_THREADS)
			d[o] = tile[o];
	}
	else
		for (uint32_t o = threadIdx.x; o < tb; o += TILE_THREADS)
			d[o] = tile[o];
	if (blockIdx.x == 0) {
		const uint64_t rem = bytes - n * T;
		for (uint64_t o = threadIdx.x; o < rem; o += TILE_THREADS)
			dst[n * T + o] = src[n * T + o];
	}
}

// stream layout of delta.cpp: one stream when bytes <= 2048, else quarters of bytes/4 plus a tail that
// continues the last quarter
__device__ __forceinline__ bool delta_is_start(uint64_t i, uint64_t bytes, uint64_t q)
{
	if (i == 0)
		return true;
	if (bytes <= 2048)
		return false;
	return i == q || i == 2 * q || i == 3 * q;
}

__global__ __launch_bounds__(256) void delta_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, uint64_t bytes)
{
	const uint64_t q = bytes / 4;
	const uint64_t base = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 16;
	if (base >= bytes)
		return;
	uint8_t prev = base ? src[base - 1] : 0;
	const uint32_t cnt = (uint32_t)((bytes - base) < 16 ? (bytes - base) : 16);
	uint8_t v[16];
	if (cnt == 16 && (((uintptr_t)(src + base)) & 15u) == 0)
		*(uint4*)v = *(const uint4*)(src + base);
	else
		for (uint32_t k = 0; k < cnt; ++k)
			v[k] = src[base + k];
	uint8_t o[16];
	for (uint32_t k = 0; k < cnt; ++k) {
		o[k] = delta_is_start(base + k, bytes, q) ? v[k] : (uint8_t)(v[k] - prev);
		prev = v[k];
	}
	if (cnt == 16 && (((uintptr_t)(dst + base)) & 15u) == 0)
		*(uint4*)(dst + base) = *(const uint4*)o;
	else
		for (uint32_t k = 0; k < cnt; ++k)
			dst[base + k] = o[k];
}

// One workgroup per stream: chunked inclusive byte prefix sum with a running carry.
__global__ __launch_bounds__(256) void delta_inv_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, uint64_t bytes)
{
	__shared__ uint32_t sums[256];
	__shared__ uint32_t carry_s;
	const uint64_t q = bytes / 4;
	uint64_t begin, end;
	if (bytes <= 2048) {
		begin = 0;
		end = blockIdx.x == 0 ? bytes : 0;
	}
	else {
		begin = blockIdx.x * q;
		end = blockIdx.x == 3 ? bytes : begin + q; // the tail (bytes % 4) continues the last quarter
	}
	if (threadIdx.x == 0)
		carry_s = 0;
	__syncthreads();
	for (uint64_t c = begin; c < end; c += 256 * 16) {
		const uint64_t base = c + (uint64_t)threadIdx.x * 16;
		const uint32_t cnt = base < end ? (uint32_t)((end - base) < 16 ? (end - base) : 16) : 0;
		uint8_t v[16];
		uint32_t acc = 0;
		for (uint32_t k = 0; k < cnt; ++k) {
			acc = (acc + src[base + k]) & 0xFF;
			v[k] = (uint8_t)acc;
		}
		sums[threadIdx.x] = acc;
		__syncthreads();
		for (uint32_t d = 1; d < 256; d <<= 1) {
			uint32_t t = threadIdx.x >= d ? sums[threadIdx.x - d] : 0;
			__syncthreads();
			sums[threadIdx.x] = (sums[threadIdx.x] + t) & 0xFF;
			__syncthreads();
		}
		const uint32_t before = (carry_s + (threadIdx.x ? sums[threadIdx.x - 1] : 0)) & 0xFF;
		for (uint32_t k = 0; k < cnt; ++k)
			dst[base + k] = (uint8_t)(v[k] + before);
		__syncthreads();
		if (threadIdx.x == 255)
			carry_s = (carry_s + sums[255]) & 0xFF;
		__syncthreads();
	}
}

// All superblocks of a buffer at once: superblock s = bytes [s*sb, min((s+1)*sb, total)) is shuffled on its own
// (stenos.cpp:513), tiles_per_sb workgroups each.
__global__ __launch_bounds__(TILE_THREADS) void shuffle_superblocks_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, uint32_t T,
									    uint64_t sb, uint64_t total, uint32_t E, uint32_t tiles_per_sb)
{
	const uint64_t s = blockIdx.x / tiles_per_sb;
	const uint32_t tix = blockIdx.x % tiles_per_sb;
	const uint64_t begin = s * sb;
	const uint64_t bytes = (total - begin) < sb ? (total - begin) : sb;
	const uint64_t n = bytes / T;
	const uint64_t e0 = (uint64_t)tix * E;
	const uint8_t* sp = src + begin;
	uint8_t* dp = dst + begin;
	if (tix == 0) { // leftover bytes of the superblock
		const uint64_t rem = bytes - n * T;
		for (uint64_t o = threadIdx.x; o < rem; o += TILE_THREADS)
			dp[n * T + o] = sp[n * T + o];
	}
	if (e0 >= n)
		return;
	const uint32_t cnt = (uint32_t)((n - e0) < E ? (n - e0) : E);
	const uint32_t tb = cnt * T;
	const uint8_t* t0 = sp + e0 * T;
	for (uint32_t o = threadIdx.x; o < tb; o += TILE_THREADS)
		tile[o] = t0[o];
	__syncthreads();
	for (uint32_t j = 0; j < T; ++j) {
		uint8_t* d = dp + (uint64_t)j * n + e0;
		for (uint32_t e = threadIdx.x; e < cnt; e += TILE_THREADS)
			d[e] = tile[e * T + j];
	}
}

// Byte delta (delta.cpp:30-71) of the `step` bytes around the middle of every byte plane of every shuffled
// superblock: the input of guess_transposed_lz_ratio (stenos.cpp:388-392).  One workgroup per (superblock, plane);
// out receives the T pieces of superblock s back to back at s*sb.
__global__ __launch_bounds__(256) void delta_middles_kernel(const uint8_t* __restrict__ shuffled, uint8_t* __restrict__ out, uint32_t T, uint64_t sb,
							     uint64_t total, uint32_t level, uint32_t with_delta)
{
	const uint64_t s = blockIdx.x / T;
	const uint32_t i = blockIdx.x % T;
	const uint64_t begin = s * sb;
	const uint64_t bytes = (total - begin) < sb ? (total - begin) : sb;
	const uint64_t elements = bytes / T;
	uint64_t step = elements / (16 / (level - 1));
	if (step < 64)
		step = elements;
	const uint8_t* in = shuffled + begin + i * elements + (elements - step) / 2;
	uint8_t* o = out + begin + i * step;
	const uint64_t q = step / 4;
	for (uint64_t k = threadIdx.x; k < step; k += 256) {
		const bool start = !with_delta || delta_is_start(k, step, q);
		o[k] = start ? in[k] : (uint8_t)(in[k] - in[k - 1]);
	}
}

uint32_t tile_elements(uint32_t T)
{
	uint32_t e = (48u * 1024u) / T;
	e &= ~1023u; // whole number of 4-element groups per thread round
	if (e == 0)
		e = 256;
	return e > 4096 ? 4096 : e;
}

} // namespace

hipError_t stenos_k_launch_shuffle(const uint8_t* src, uint8_t* dst, uint32_t T, uint64_t bytes, bool inverse, hipStream_t stream)
{
	if (bytes == 0)
		return hipSuccess;
	if (T == 1)
		return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, stream); // shuffle.cpp:82-103
	const uint64_t n = bytes / T;
	const uint32_t E = tile_elements(T);
	const uint32_t grid = (uint32_t)((n + E - 1) / E);
	const size_t lds = (size_t)E * T;
	if (n == 0)
		return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, stream);
	if (inverse) {
		hipError_t e = hipFuncSetAttribute((const void*)unshuffle_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
		if (e != hipSuccess)
			return e;
		hipLaunchKernelGGL(unshuffle_kernel, dim3(grid), dim3(TILE_THREADS), lds, stream, src, dst, T, n, E, bytes);
	}
	else {
		hipError_t e = hipFuncSetAttribute((const void*)shuffle_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
		if (e != hipSuccess)
			return e;
		hipLaunchKernelGGL(shuffle_kernel, dim3(grid), dim3(TILE_THREADS), lds, stream, src, dst, T, n, E, bytes);
	}
	return hipGetLastError();
}

// Levels >= 2, device destinations: the superblocks of a batch come out of the host's zstd in slots of `stride` bytes (piece k
// at src + k * stride, size[k] bytes, [code][csize:3] header included); each goes to dst + off[k] -- the frame is laid out
// on the device instead of by the host's threads.  One workgroup per piece; the source slots are 16-byte aligned, the
// destination is wherever the frame puts the piece (unaligned 16-byte stores).
__global__ __launch_bounds__(256) void gather_pieces_kernel(const uint8_t* __restrict__ src, uint64_t stride, const uint64_t* __restrict__ off,
							     const uint64_t* __restrict__ size, uint8_t* __restrict__ dst)
{
	typedef uint4 __attribute__((aligned(1))) uint4_u;
	const uint8_t* from = src + (uint64_t)blockIdx.x * stride;
	uint8_t* to = dst + off[blockIdx.x];
	const uint64_t n = size[blockIdx.x], groups = n >> 4;
	for (uint64_t g = threadIdx.x; g < groups; g += 256)
		*(uint4_u*)(to + g * 16) = *(const uint4*)(from + g * 16);
	for (uint64_t b = groups * 16 + threadIdx.x; b < n; b += 256)
		to[b] = from[b];
}
hipError_t stenos_k_launch_gather_pieces(const uint8_t* src, uint64_t stride, const uint64_t* off, const uint64_t* size, uint32_t count, uint8_t* dst, hipStream_t stream)
{
	if (count == 0)
		return hipSuccess;
	hipLaunchKernelGGL(gather_pieces_kernel, dim3(count), dim3(256), 0, stream, src, stride, off, size, dst);
	return hipGetLastError();
}

hipError_t stenos_k_launch_delta(const uint8_t* src, uint8_t* dst, uint64_t bytes, bool inverse, hipStream_t stream)
{
	if (bytes == 0)
		return hipSuccess;
	if (inverse)
		hipLaunchKernelGGL(delta_inv_kernel, dim3(bytes <= 2048 ? 1 : 4), dim3(256), 0, stream, src, dst, bytes);
	else
		hipLaunchKernelGGL(delta_kernel, dim3((uint32_t)((bytes + 4095) / 4096)), dim3(256), 0, stream, src, dst, bytes);
	return hipGetLastError();
}

hipError_t stenos_k_launch_shuffle_superblocks(const uint8_t* src, uint8_t* dst, uint32_t T, uint64_t sb, uint64_t total, hipStream_t stream)
{
	if (total == 0)
		return hipSuccess;
	if (T == 1)
		return hipMemcpyAsync(dst, src, total, hipMemcpyDeviceToDevice, stream);
	const uint32_t E = tile_elements(T);
	const uint64_t nsb = (total + sb - 1) / sb;
	const uint32_t tiles = (uint32_t)((sb / T + E - 1) / E);
	const size_t lds = (size_t)E * T;
	hipError_t e = hipFuncSetAttribute((const void*)shuffle_superblocks_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
	if (e != hipSuccess)
		return e;
	hipLaunchKernelGGL(shuffle_superblocks_kernel, dim3((uint32_t)(nsb * tiles)), dim3(TILE_THREADS), lds, stream, src, dst, T, sb, total, E, tiles);
	return hipGetLastError();
}

hipError_t stenos_k_launch_delta_middles(const uint8_t* shuffled, uint8_t* out, uint32_t T, uint64_t sb, uint64_t total, uint32_t level, bool with_delta,
					 hipStream_t stream)
{
	if (total == 0)
		return hipSuccess;
	const uint64_t nsb = (total + sb - 1) / sb;
	hipLaunchKernelGGL(delta_middles_kernel, dim3((uint32_t)(nsb * T)), dim3(256), 0, stream, shuffled, out, T, sb, total, level, with_delta ? 1u : 0u);
	return hipGetLastError();
}
