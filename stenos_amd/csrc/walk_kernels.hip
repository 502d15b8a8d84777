// walk_kernels.hip -- the superblock index of a device-resident frame (stenos.cpp:1126-1134, 1166-1182), found by
// segments of the frame in parallel and proven equal to the serial walk, which stays as the fallback (walk.h).
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "walk.h"

using namespace walk;

namespace {

// Serial walk of the superblock chain by one lane: off[s] = byte offset of superblock s's header.  only_if: run only when
// *only_if is non-zero (the fallback of the parallel walk); NULL: always.
__global__ void walk_superblocks(const uint8_t* __restrict__ frame, uint64_t size, uint64_t first, uint64_t nsb, uint64_t* __restrict__ off,
				 uint32_t* __restrict__ status, const uint32_t* __restrict__ only_if)
{
	if (threadIdx.x != 0 || blockIdx.x != 0)
		return;
	if (only_if && *only_if == 0)
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

struct SharedAdd {
	__device__ uint32_t operator()(uint32_t* c) const { return atomicAdd(c, 1u); }
};

// phase A: one workgroup per segment -- four wavefronts look through the window, the first one follows the roots
constexpr uint32_t SCAN_THREADS = 256;
__global__ __launch_bounds__(SCAN_THREADS) void walk_speculate(Plan P, const uint8_t* __restrict__ frame, Segment* __restrict__ seg, uint32_t* __restrict__ flags)
{
	const uint32_t k = blockIdx.x, lane = threadIdx.x;
	if (k == 0) {
		if (lane == 0) {
			*flags = 0;
			follow_first(P, frame, &seg[0]);
		}
		return;
	}
	if (k + 1 >= P.nseg) // the last segment is walked in phase B
		return;
	__shared__ uint64_t roots[LANES];
	__shared__ uint32_t nroots;
	if (lane == 0)
		nroots = 0;
	__syncthreads();
	const uint64_t begin = seg_begin(P, k), wend = begin + P.window;
	for (uint64_t base = begin + lane * 16u; base < wend; base += SCAN_THREADS * 16u)
		scan_window16(P, frame, k, base, roots, &nroots, SharedAdd());
	__syncthreads();
	if (lane >= LANES)
		return;
	const uint32_t n = nroots;
	uint64_t exit = 0;
	uint32_t hops = 0;
	bool alive = false;
	if (n <= LANES && lane < n)
		alive = follow_root(P, frame, k, roots[lane], &exit, &hops);
	const uint64_t live = __ballot(alive);
	const uint32_t nlive = (uint32_t)__popcll(live);
	bool ok = nlive >= 1 && nlive <= MAX_ROOTS;
	if (ok) { // survivors must have merged: one exit
		const int lead = __ffsll((unsigned long long)live) - 1;
		const uint64_t e0 = __shfl(exit, lead);
		ok = __ballot(alive && exit != e0) == 0;
	}
	if (ok) {
		if (alive) {
			const uint32_t i = (uint32_t)__popcll(live & ((1ull << lane) - 1ull));
			seg[k].root[i] = roots[lane];
			seg[k].hops[i] = hops;
			if (i == 0) {
				seg[k].exit = exit;
				seg[k].count = 0;
				seg[k].nroots = nlive;
				seg[k].state = SEG_OK;
			}
		}
	}
	else if (lane == 0)
		seg[k].state = SEG_UNRESOLVED;
}

// phase B: one lane per segment
__global__ __launch_bounds__(64) void walk_verify(Plan P, const uint8_t* __restrict__ frame, Segment* __restrict__ seg, uint32_t* __restrict__ flags)
{
	const uint32_t k = blockIdx.x * 64u + threadIdx.x;
	if (k >= P.nseg)
		return;
	if (!verify_segment(P, frame, k, seg))
		atomicOr(flags, WALK_FAILED);
}

// phase C: one wavefront per segment (64 lanes sum the counts in front, lane 0 writes)
__global__ __launch_bounds__(64) void walk_write(Plan P, const uint8_t* __restrict__ frame, const Segment* __restrict__ seg, const uint32_t* __restrict__ flags,
						 uint64_t* __restrict__ off, uint32_t* __restrict__ status)
{
	if (*flags)
		return;
	const uint32_t k = blockIdx.x, lane = threadIdx.x;
	uint64_t sum = 0;
	for (uint32_t j = lane; j < k; j += LANES)
		sum += seg[j].count;
	for (uint32_t o = 32; o; o >>= 1)
		sum += __shfl_xor(sum, (int)o);
	if (lane == 0) {
		const uint32_t bits = write_segment(P, frame, k, seg, sum, off, DECODE_STATUS_TRUNCATED);
		if (bits)
			atomicOr(status, bits);
	}
}

} // namespace

hipError_t stenos_k_launch_walk(const uint8_t* frame, uint64_t size, uint64_t first, uint64_t nsb, uint32_t sb_bytes, uint64_t* off, uint32_t* status,
				void* scratch, hipStream_t stream)
{
	const Plan P = make_plan(first, size, nsb, sb_bytes);
	if (P.nseg == 0 || !scratch) {
		hipLaunchKernelGGL(walk_superblocks, dim3(1), dim3(64), 0, stream, frame, size, first, nsb, off, status, (const uint32_t*)nullptr);
		return hipGetLastError();
	}
	uint32_t* flags = (uint32_t*)scratch;
	Segment* seg = (Segment*)((uint8_t*)scratch + 64);
	hipLaunchKernelGGL(walk_speculate, dim3(P.nseg), dim3(SCAN_THREADS), 0, stream, P, frame, seg, flags);
	hipLaunchKernelGGL(walk_verify, dim3((P.nseg + 63) / 64), dim3(64), 0, stream, P, frame, seg, flags);
	hipLaunchKernelGGL(walk_write, dim3(P.nseg), dim3(64), 0, stream, P, frame, (const Segment*)seg, (const uint32_t*)flags, off, status);
	hipLaunchKernelGGL(walk_superblocks, dim3(1), dim3(64), 0, stream, frame, size, first, nsb, off, status, (const uint32_t*)flags);
	return hipGetLastError();
}
size_t stenos_k_walk_scratch_bytes() { return 64 + (size_t)MAX_SEGMENTS * sizeof(Segment); }
