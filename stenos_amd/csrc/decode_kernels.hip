// decode_kernels.hip -- gfx950 kernels of the decode pipeline and their launchers.
//   (the [code][csize:3] chain of a frame that comes without an index is walked by walk_kernels.hip)
//   decode_superblocks one wavefront per superblock (block_compress.h:2088-2175)
//
// A translation unit of its own because it is compiled with -mllvm -structurizecfg-skip-uniform-regions (csrc/Makefile):
// decode_superblocks has no divergent branch at all (wavevec.h, "Predicated memory accesses without a branch"), and with
// that option the compiler leaves its wave-uniform control flow -- the dispatch on block markers, plane types and row
// kinds that makes up most of a decoder -- as plain scalar compares and branches instead of rebuilding it around flags
// held in lane masks: 134 -> 96 scalar instructions per 1 KiB block of the int32 headline, int16 random walk decode
// 2.21 -> 2.03 ms per 4 GiB.  The encoders are slightly slower with it and keep the default.
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "kernels.h"

using namespace codec;
using namespace wv;

namespace {

extern __shared__ __attribute__((aligned(16))) uint8_t g_lds[];

// Waves per SIMD the kernel for bytesoftype TT is compiled for.  bytesoftype 8: a wave's window and image take 7 KB of LDS,
// so a CU holds 23 of them (five to six per SIMD) whatever is said here; saying six gives the kernel 80 vector registers.
#ifndef STENOS_DECODE_OCCUPANCY_T8
#define STENOS_DECODE_OCCUPANCY_T8 8
#endif
constexpr uint32_t decode_occupancy(uint32_t TT) { return TT == 8 ? STENOS_DECODE_OCCUPANCY_T8 : 8; }
template <uint32_t TT>
__global__ __launch_bounds__(64, decode_occupancy(TT)) void decode_superblocks(DecodeArgs a)
{
	const uint32_t T = TT ? TT : a.T;
	const uint64_t s = a.sb_ids ? a.sb_ids[blockIdx.x] : blockIdx.x;
	const U32 lane = lane_id();
	const uint64_t p = a.sb_off[blockIdx.x];
	if (p > a.size || a.size - p < 4) { // (written without sums: an index entry may hold anything)
		status_or(a.status, DECODE_STATUS_TRUNCATED);
		return;
	}
	const uint32_t code = a.frame[p];
	const uint32_t csize = (uint32_t)a.frame[p + 1] | ((uint32_t)a.frame[p + 2] << 8) | ((uint32_t)a.frame[p + 3] << 16);
	const uint64_t begin = s * (uint64_t)a.sb_bytes;
	const uint32_t dsize = (uint32_t)((a.total_bytes - begin) < a.sb_bytes ? (a.total_bytes - begin) : a.sb_bytes);
	if (a.size - p - 4 < csize) { // stenos.cpp:1133-1134
		status_or(a.status, DECODE_STATUS_TRUNCATED);
		return;
	}
	const uint8_t* payload = a.frame + p + 4;
	uint8_t* out = a.dst + begin;
	if (code == 1) {
		const DecLayout L = make_dec_layout(T);
		uint32_t r = decode_superblock(g_lds, L, T, payload, csize, out, dsize, TT != 0);
		if (r == DEC_ERROR)
			status_or(a.status, DECODE_STATUS_INVALID);
	}
	else if (code == 6) { // stenos.cpp:741-746
		if (csize != dsize) {
			status_or(a.status, DECODE_STATUS_INVALID);
			return;
		}
		copy_g2g_wide<COPY_ROUNDS>(out, payload, csize); // (this kernel has registers to spare: more loads in flight per trip)
	}
	else if (code >= 2 && code <= 5) { // zstd based codes are finished by the host
		status_or(a.status, DECODE_STATUS_HOST_CODES);
	}
	else
		status_or(a.status, DECODE_STATUS_INVALID);
	(void)lane;
}

} // namespace

template <uint32_t TT>
static hipError_t launch_decode_t(const DecodeArgs& a, hipStream_t stream)
{
	const size_t lds = stenos_k_decode_lds_bytes(a.T);
	hipError_t e = hipFuncSetAttribute((const void*)decode_superblocks<TT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
	if (e != hipSuccess)
		return e;
	hipLaunchKernelGGL(decode_superblocks<TT>, dim3((uint32_t)a.nsb), dim3(64), lds, stream, a);
	return hipGetLastError();
}

hipError_t stenos_k_launch_decode(const DecodeArgs& a, hipStream_t stream)
{
	if (a.T > STENOS_K_LDS_MAX_T)
		return stenos_kw_launch_decode(a, stream);
	switch (a.T) {
		case 2: return launch_decode_t<2>(a, stream);
		case 4: return launch_decode_t<4>(a, stream);
		case 8: return launch_decode_t<8>(a, stream);
		default: return launch_decode_t<0>(a, stream);
	}
}
