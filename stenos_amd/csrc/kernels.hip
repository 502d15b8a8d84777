// kernels.hip -- gfx950 kernels of the level-1 block codec and their launchers.
//
// Encode pipeline (all on one stream, no host round trip):
//   encode_blocks      one wavefront per 256-element block: HBM -> LDS -> encoded image -> 16-byte
//                      aligned slot in the workspace, size to bsize[]            (block_compress.h:1152-1298)
//   plan_superblocks   one wavefront per superblock: payload size, BLOCK vs COPY decision
//                      (stenos.cpp:606-615), offsets of its blocks, capacity requirement
//   scan_superblocks   exclusive scan of the superblock sizes -> byte offset of every superblock header
//   resolve_frame      one wavefront: exact replay of the reference's capacity rules for the superblocks
//                      the plan flagged (normally none or the last; pipeline.h)
//   pack_frame         four wavefronts per superblock, destination-ordered gather of the block slots,
//                      frame header (stenos.cpp:862-874)
// The decode pipeline lives in decode_kernels.hip (a translation unit with compiler options of its own).
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "kernels.h"

using namespace codec;
using namespace wv;

namespace {

extern __shared__ __attribute__((aligned(16))) uint8_t g_lds[];

// TT: bytesoftype known at compile time (2, 4, 8: loops unroll, plane words stay in registers) or 0 = runtime value
template <uint32_t TT>
__global__ __launch_bounds__(64) void encode_blocks(const uint8_t* __restrict__ src, uint64_t b_begin, uint64_t b_end, uint64_t nfull, uint32_t tail_bytes, uint32_t Trt,
						    uint8_t* __restrict__ slots, uint32_t slot_stride, uint32_t* __restrict__ bsize,
						    uint32_t* __restrict__ binfo, uint32_t* __restrict__ bneed)
{
	const uint32_t T = TT ? TT : Trt;
	const Layout L = make_layout(T, true);
	for (uint64_t b = b_begin + blockIdx.x; b < b_end; b += gridDim.x) {
		BlockInfo r;
		if (b < nfull)
			r = encode_block_job(g_lds, L, T, src + b * (uint64_t)(256 * T), slots + b * (uint64_t)slot_stride, true);
		else
			r = encode_tail_job(g_lds, L, T, src + nfull * (uint64_t)(256 * T), tail_bytes, slots + nfull * (uint64_t)slot_stride);
		if (threadIdx.x == 0) {
			bsize[b] = r.size;
			binfo[b] = r.info;
			bneed[b] = r.need;
		}
	}
}

// Start of a compression job: misc = { total = first_off, [8,12) untouched, status = 0, first_flagged = none,
// [20,24) untouched, scan carry = first_off } and two word ranges cleared (the fused path's tickets / sizes and
// the superblock offsets it polls).  One launch instead of five small copies and fills.
__global__ __launch_bounds__(256) void init_job(uint8_t* __restrict__ misc, uint64_t first_off, uint64_t* __restrict__ z1, uint64_t n1,
						 uint64_t* __restrict__ z2, uint64_t n2)
{
	const uint64_t i = blockIdx.x * 256ull + threadIdx.x, step = gridDim.x * 256ull;
	if (i == 0) {
		*(uint64_t*)misc = first_off;
		*(uint32_t*)(misc + 12) = 0u;
		*(uint32_t*)(misc + 16) = 0xFFFFFFFFu;
		*(uint32_t*)(misc + 20) = 0u;
		*(uint64_t*)(misc + 24) = first_off;
	}
	for (uint64_t k = i; k < n1; k += step)
		z1[k] = 0;
	for (uint64_t k = i; k < n2; k += step)
		z2[k] = 0;
}

// ---- fused path -----------------------------------------------------------------------------------------
// Frame offsets of the superblocks are produced while the encoders run.  Every workgroup publishes the bytes its
// superblock takes (size[s], non-zero) as soon as the blocks are encoded; one scanner wavefront (workgroup 0) follows
// the sizes in superblock order and publishes the offsets (sb_off[s], non-zero).  An encoder workgroup stores
// superblock k only after it has encoded superblock k+1, so the offset is normally there when it asks for it.
//
// Progress: superblock numbers are tickets taken when the work on them starts, and between taking a ticket and
// publishing its size a workgroup waits for nothing; the scanner is the first workgroup of the grid.  So every wait
// ends.  A poll counter bounds them anyway: a wait that would hang the device is reported as an error instead.
constexpr uint32_t CHAIN_SPIN_LIMIT = 1u << 22;
constexpr uint64_t CHAIN_FAILED = ~0ull;
// Waves per SIMD the kernel for bytesoftype TT is compiled for (and, times four SIMDs, the workgroups that stay resident per
// CU).  bytesoftype 8: its four waves need 30 KB of LDS, so five workgroups are all a CU holds anyway -- saying so gives the
// encoder 96 vector registers instead of 64 (double sine: 8.5 -> 8.2 ms per 8 GiB).
constexpr uint32_t fused_occupancy(uint32_t TT) { return TT == 8 ? 5u : 8u; }
constexpr bool FUSED_SPECULATE = true; // raw bytes of measured superblocks go straight to where a copy behind copies stands

__device__ inline void chain_put(uint64_t* p, uint64_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline uint64_t chain_get(const uint64_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// one wavefront: sizes -> offsets, in order, up to 256 superblocks per round (four per lane, all loads in flight
// together: the encoders finish a superblock every 0.1 us, a round trip to the sizes takes a few us)
__device__ void chain_scanner(const FrameJob& j, uint64_t nsb, const uint64_t* size, uint64_t* carry)
{
	constexpr uint32_t PER = 4;
	const uint32_t lane = threadIdx.x & 63u;
	__builtin_amdgcn_s_setprio(3);
	uint64_t running = j.header_bytes, base = 0;
	uint32_t spins = 0;
	while (base < nsb) {
		uint64_t d[PER];
		for (uint32_t q = 0; q < PER; ++q) {
			const uint64_t idx = base + lane * PER + q;
			d[q] = idx < nsb ? chain_get(size + idx) : 0;
		}
		uint32_t mine = 0; // leading sizes of this lane that are known
		while (mine < PER && d[mine])
			++mine;
		const uint64_t partial = __ballot(mine < PER);
		const uint32_t first = partial ? (uint32_t)__builtin_ctzll(partial) : 64u; // lanes before it are complete
		const uint32_t ready = first == 64u ? 64u * PER : first * PER + (uint32_t)__shfl((int)mine, (int)first);
		if (ready == 0) {
			if (++spins > CHAIN_SPIN_LIMIT) {
				if (lane == 0)
					atomicOr(j.status, ENCODE_STATUS_CHAIN_TIMEOUT);
				return;
			}
			__builtin_amdgcn_s_sleep(2);
			continue;
		}
		spins = 0;
		const uint32_t take = lane < first ? PER : (lane == first ? mine : 0u); // entries of this lane inside the ready prefix
		uint64_t sum = 0;
		for (uint32_t q = 0; q < PER; ++q)
			sum += q < take ? d[q] : 0;
		uint64_t incl = sum;
		for (uint32_t o = 1; o < 64; o <<= 1) {
			const uint64_t up = __shfl_up(incl, o);
			if (lane >= o)
				incl += up;
		}
		uint64_t off = running + incl - sum;
		for (uint32_t q = 0; q < PER; ++q)
			if (q < take) {
				chain_put(j.sb_off + base + lane * PER + q, off);
				off += d[q];
			}
		running += __shfl(incl, 63);
		base += ready;
	}
	if (lane == 0) { // what scan_superblocks leaves behind for the ranges that follow
		*carry = running;
		j.sb_off[nsb] = running;
		*j.total = running;
	}
}

// offset of superblock s once the scanner has published it (every lane polls the same word)
__device__ uint64_t chain_wait(const FrameJob& j, uint64_t s)
{
	for (uint32_t spins = 0; spins <= CHAIN_SPIN_LIMIT; ++spins) {
		const uint64_t v = chain_get(j.sb_off + s);
		// every lane read the same word: say so, or everything derived from the offset lives in vector registers
		const uint64_t off = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)v) |
				     ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(v >> 32)) << 32);
		if (off)
			return off;
		__builtin_amdgcn_s_sleep(4);
	}
	if ((threadIdx.x & 63u) == 0)
		atomicOr(j.status, ENCODE_STATUS_CHAIN_TIMEOUT);
	return CHAIN_FAILED;
}

// Workgroup 0: the scanner.  Every other workgroup: FUSED_WAVES wavefronts that take superblocks one after the other
// until none is left -- encode (each wave a run of consecutive blocks into its staging stream), publish the size, then
// store the previous superblock at its offset (pipeline.h, fused_store): by then the scanner has normally passed it.
// A workgroup owns two staging buffers and alternates between them.
template <uint32_t TT>
__global__ __launch_bounds__(64 * FUSED_WAVES, fused_occupancy(TT)) void encode_superblocks(FrameJob j, uint64_t nsb, uint8_t* __restrict__ stage, uint32_t run_cap,
									uint64_t* __restrict__ size, uint32_t* __restrict__ ticket, uint64_t* __restrict__ carry)
{
	if (blockIdx.x == 0) {
		if (threadIdx.x < 64)
			chain_scanner(j, nsb, size, carry);
		return;
	}
	const uint32_t T = TT ? TT : j.T;
	const Layout L = make_layout(T, true);
	const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	volatile uint32_t* shared = (volatile uint32_t*)(g_lds + FUSED_WAVES * L.total); // [parity] ticket, [2] a wait has failed, [8 + 4*parity ..] run sizes
	uint8_t* const stage_w = stage + (((uint64_t)(blockIdx.x - 1) * 2) * FUSED_WAVES + w) * run_cap; // + parity * FUSED_WAVES * run_cap
	uint32_t b0, b1;
	fused_run_range(j.bps, w, &b0, &b1);
	uint64_t prev = CHAIN_FAILED; // superblock that is encoded but not stored yet
	bool prev_spec = false;       // ... and its raw bytes were put where a copy behind copies goes
	uint32_t prev_run[FUSED_WAVES];
	bool guess_copy = false; // the workgroup's last superblock ended up as a copy
	bool group_hint = true;  // the wave's last blocks had the shape that groups of four blocks want (superblock_codec.h)
	if (threadIdx.x == 0)
		shared[2] = 0;
	for (uint32_t it = 0;; ++it) {
		const uint32_t parity = it & 1u;
		if (threadIdx.x == 0)
			shared[parity] = atomicAdd(ticket, 1u);
		__syncthreads();
		if (shared[2]) // a wavefront of this workgroup gave up waiting: all leave together (the error is in j.status)
			return;
		// readfirstlane yields an int: go through uint32_t or values beyond 2^31 get sign-extended
		const uint64_t s = (uint32_t)__builtin_amdgcn_readfirstlane(shared[parity]);
		const bool work = s < nsb;
		volatile uint32_t* runs = shared + 8 + FUSED_WAVES * parity;
		uint32_t run_size[FUSED_WAVES];
		bool spec_now = false;
		if (work) {
			// A superblock whose block stream comes out larger than its input is stored as a copy (stenos.cpp:609-610) and
			// the stream is thrown away: after such a superblock the next one is only measured (nothing written, nothing
			// staged) and encoded for real only if the guess was wrong.  Incompressible data is incompressible throughout.
			const uint8_t* from = j.src + (s * j.bps + b0) * (uint64_t)(256 * T);
			uint8_t* to = stage_w + (uint64_t)parity * FUSED_WAVES * run_cap;
			for (uint32_t attempt = 0;; ++attempt) {
				const bool measure = guess_copy && attempt == 0;
				// Speculative copy.  A superblock that is only measured is expected to be stored as a copy, and if every superblock
				// before it is one too its place in the frame is known: header + s * (superblock + 4).  Its raw bytes go there
				// while they pass through the registers of the measuring pass; when the offset arrives and is that place, the copy
				// (a second read of the input, a second pass over 128 KiB) is not needed.  A wrong guess costs nothing but the
				// stores: the bytes land at or behind the superblock's real place, inside the frame's worst case, and whoever owns
				// those bytes writes them later -- later superblocks learn their offsets only after this one has published its
				// size, which it does after its stores -- write-through: the XCDs' L2s are not coherent -- have completed.
				uint8_t* const spec_to = measure && FUSED_SPECULATE ? j.dst + j.header_bytes + s * (uint64_t)(j.sb_bytes + 4) + 4 + (uint64_t)b0 * (256 * T) : nullptr;
				const uint32_t n = encode_run(g_lds + w * L.total, L, T, from, b1 - b0, measure ? nullptr : to, true, NoPassHook(), spec_to, &group_hint);
				if (spec_to)
					gst_through_wait(); // (write-through stores: in memory before the size is published)
				if ((threadIdx.x & 63u) == 0)
					runs[w] = n;
				__syncthreads();
				for (uint32_t k = 0; k < FUSED_WAVES; ++k)
					run_size[k] = (uint32_t)__builtin_amdgcn_readfirstlane(runs[k]);
				uint32_t code;
				const uint32_t bytes = fused_superblock_size(j, run_size, &code);
				if (measure && code != 6) { // it does compress: once more, with the bytes
					__syncthreads(); // (everyone has read the sizes before they are written again)
					continue;
				}
				guess_copy = code == 6;
				spec_now = measure && FUSED_SPECULATE; // (its raw bytes stand at the speculated place; code is 6 here)
				if (threadIdx.x == 0)
					chain_put(size + s, bytes);
				break;
			}
		}
		else
			__syncthreads();
		if (prev != CHAIN_FAILED) {
			const uint64_t off = chain_wait(j, prev);
			if (off == CHAIN_FAILED)
				shared[2] = 1;
			else
				fused_store(j, prev, w, off, prev_run, stage_w + (uint64_t)(parity ^ 1u) * FUSED_WAVES * run_cap,
					    prev_spec && off == j.header_bytes + prev * (uint64_t)(j.sb_bytes + 4));
		}
		if (!work)
			return;
		prev = s;
		prev_spec = spec_now;
		for (uint32_t k = 0; k < FUSED_WAVES; ++k)
			prev_run[k] = run_size[k];
	}
}

// One wavefront per superblock.
__global__ __launch_bounds__(64) void plan_superblocks(FrameJob j, uint64_t s_begin)
{
	const Layout L = make_layout(j.T, true);
	plan_superblock(g_lds, L, j, s_begin + blockIdx.x);
}

// Exclusive scan of (csize + 4) over superblocks [s_begin, s_begin + n) by one workgroup of 1024 threads.
// *carry holds the frame offset of superblock s_begin on entry and of s_begin + n on exit, so consecutive
// ranges chain on the device.
__global__ __launch_bounds__(1024) void scan_superblocks(const uint32_t* __restrict__ csize, uint64_t s_begin, uint64_t n, uint64_t* __restrict__ carry,
							 uint64_t* __restrict__ off, uint64_t* __restrict__ total)
{
	__shared__ uint64_t partial[1024];
	const uint32_t tid = threadIdx.x;
	const uint64_t per = (n + 1023) / 1024;
	const uint64_t lo = tid * per < n ? tid * per : n, hi = lo + per < n ? lo + per : n;
	const uint64_t start = *carry;
	uint64_t sum = 0;
	for (uint64_t i = lo; i < hi; ++i)
		sum += (uint64_t)csize[s_begin + i] + 4;
	partial[tid] = sum;
	__syncthreads();
	for (uint32_t d = 1; d < 1024; d <<= 1) { // Hillis-Steele inclusive scan
		uint64_t v = tid >= d ? partial[tid - d] : 0;
		__syncthreads();
		partial[tid] += v;
		__syncthreads();
	}
	uint64_t base = start + partial[tid] - sum;
	for (uint64_t i = lo; i < hi; ++i) {
		off[s_begin + i] = base;
		base += (uint64_t)csize[s_begin + i] + 4;
	}
	__syncthreads(); // every thread has read *carry
	if (tid == 1023) {
		off[s_begin + n] = start + partial[1023];
		*total = start + partial[1023];
		*carry = start + partial[1023];
	}
}

// levels >= 2: the host has decided which superblocks keep their block stream (keep[s] != 0); the others leave no
// payload behind (code 0, size 0: pack_frame then writes their 4-byte header only)
__global__ __launch_bounds__(256) void keep_superblocks(const uint8_t* __restrict__ keep, uint8_t* __restrict__ code, uint32_t* __restrict__ csize, uint64_t nsb)
{
	const uint64_t s = blockIdx.x * 256ull + threadIdx.x;
	if (s < nsb && !keep[s]) {
		code[s] = 0;
		csize[s] = 0;
	}
}

// One wavefront replays the capacity rules where the parallel plan could not clear them.
__global__ __launch_bounds__(64) void resolve_frame(FrameJob j)
{
	const Layout L = make_layout(j.T, true);
	resolve_capacity(g_lds, L, j);
}

__global__ __launch_bounds__(64) void pack_frame(FrameJob j, uint64_t s_begin) { pack_superblock(g_lds, j, s_begin + blockIdx.x / PACK_WAVES, blockIdx.x % PACK_WAVES); }

} // namespace

// ---------------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------------

// number of CUs of the current device
uint32_t stenos_k_cu_count()
{
	static int cus = 0;
	if (!cus) {
		int dev = 0;
		hipDeviceProp_t prop;
		if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
			cus = prop.multiProcessorCount;
		if (cus <= 0)
			cus = 256;
	}
	return (uint32_t)cus;
}
size_t stenos_k_encode_lds_bytes(uint32_t T) { return make_layout(T, true).total; }
size_t stenos_k_decode_lds_bytes(uint32_t T) { return make_dec_layout(T).total; }
uint32_t stenos_k_slot_stride(uint32_t T) { return out_capacity(T); }

template <uint32_t TT>
static hipError_t launch_encode_t(const FrameJob& j, uint64_t b_begin, uint64_t b_end, hipStream_t stream)
{
	const size_t lds = stenos_k_encode_lds_bytes(j.T);
	hipError_t e = hipFuncSetAttribute((const void*)encode_blocks<TT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
	if (e != hipSuccess)
		return e;
	// One workgroup per block measured faster than a persistent grid on MI355X (10.4 ms vs 12.5-14.8 ms for
	// 8 GiB of int32): the dispatcher staggers the waves, a resident grid runs them in phase.
	const uint32_t grid = (uint32_t)(b_end - b_begin);
	hipLaunchKernelGGL(encode_blocks<TT>, dim3(grid), dim3(64), lds, stream, j.src, b_begin, b_end, j.nfull, j.tail_bytes, j.T, j.slots, j.slot_stride,
			   j.bsize, j.binfo, j.bneed);
	return hipGetLastError();
}

hipError_t stenos_k_launch_init(uint8_t* misc, uint64_t first_off, uint64_t* z1, uint64_t n1, uint64_t* z2, uint64_t n2, hipStream_t stream)
{
	const uint64_t words = n1 > n2 ? n1 : n2;
	uint32_t grid = (uint32_t)((words + 255) / 256);
	grid = grid < 1 ? 1 : (grid > 1024 ? 1024 : grid);
	hipLaunchKernelGGL(init_job, dim3(grid), dim3(256), 0, stream, misc, first_off, z1, n1, z2, n2);
	return hipGetLastError();
}

template <uint32_t TT>
static hipError_t launch_fused_t(const FrameJob& j, uint64_t nsb, uint8_t* stage, uint64_t* desc, uint32_t* ticket, uint64_t* carry, hipStream_t stream)
{
	const size_t lds = FUSED_WAVES * stenos_k_encode_lds_bytes(j.T) + 32 + 8 * FUSED_WAVES;
	hipError_t e = hipFuncSetAttribute((const void*)encode_superblocks<TT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
	if (e != hipSuccess)
		return e;
	const uint32_t grid = stenos_k_fused_groups(nsb, j.T) + 1; // + the scanner
	hipLaunchKernelGGL((encode_superblocks<TT>), dim3(grid), dim3(64 * FUSED_WAVES), lds, stream, j, nsb, stage, fused_run_capacity(j.bps, j.T), desc, ticket, carry);
	return hipGetLastError();
}

// Superblocks [0, nsb) of the job, all of them bps full blocks with room for any encoding.  desc: nsb zeroed words,
// ticket: one zeroed word, j.sb_off[0, nsb] zeroed, stage: stenos_k_fused_stage_bytes(); *carry receives the frame
// offset behind them.
hipError_t stenos_k_launch_encode_fused(const FrameJob& j, uint64_t nsb, uint8_t* stage, uint64_t* desc, uint32_t* ticket, uint64_t* carry, hipStream_t stream)
{
	if (nsb == 0)
		return hipSuccess;
	switch (j.T) {
		case 2: return launch_fused_t<2>(j, nsb, stage, desc, ticket, carry, stream);
		case 4: return launch_fused_t<4>(j, nsb, stage, desc, ticket, carry, stream);
		case 8: return launch_fused_t<8>(j, nsb, stage, desc, ticket, carry, stream);
		default: return launch_fused_t<0>(j, nsb, stage, desc, ticket, carry, stream);
	}
}
// the workgroup's scratch must fit the 160 KiB of a CU (bytesoftype up to about 40)
bool stenos_k_fused_supported(uint32_t T) { return T <= STENOS_K_LDS_MAX_T && FUSED_WAVES * stenos_k_encode_lds_bytes(T) + 32 + 8 * FUSED_WAVES <= 160u * 1024u; }
// encoder workgroups of the fused kernel: as many as stay resident (they take superblocks until none is left)
uint32_t stenos_k_fused_groups(uint64_t nsb, uint32_t T)
{
	const uint64_t resident = (uint64_t)stenos_k_cu_count() * (fused_occupancy(T) * 4 / FUSED_WAVES); // waves per SIMD x four SIMDs
	return (uint32_t)(nsb < resident ? nsb : resident);
}
// two staging buffers per workgroup
size_t stenos_k_fused_stage_bytes(uint32_t T, uint32_t bps, uint64_t nsb) { return (size_t)stenos_k_fused_groups(nsb, T) * 2 * FUSED_WAVES * fused_run_capacity(bps, T) + 64; }

// blocks [b_begin, b_end) of the job (the tail block has index nfull)
hipError_t stenos_k_launch_encode(const FrameJob& j, uint64_t b_begin, uint64_t b_end, hipStream_t stream)
{
	if (b_end <= b_begin)
		return hipSuccess;
	if (j.T > STENOS_K_LDS_MAX_T)
		return stenos_kw_launch_encode(j, b_begin, b_end, stream);
	switch (j.T) {
		case 2: return launch_encode_t<2>(j, b_begin, b_end, stream);
		case 4: return launch_encode_t<4>(j, b_begin, b_end, stream);
		case 8: return launch_encode_t<8>(j, b_begin, b_end, stream);
		default: return launch_encode_t<0>(j, b_begin, b_end, stream);
	}
}

// superblocks [s_begin, s_end)
hipError_t stenos_k_launch_plan(const FrameJob& j, uint64_t s_begin, uint64_t s_end, hipStream_t stream)
{
	if (s_end <= s_begin)
		return hipSuccess;
	if (j.T > STENOS_K_LDS_MAX_T)
		return stenos_kw_launch_plan(j, s_begin, s_end, stream);
	// the replay inside the plan (fixed-capacity mode) re-encodes blocks and needs the encoder's LDS
	const size_t lds = j.fixed_capacity ? stenos_k_encode_lds_bytes(j.T) : 0;
	if (lds) {
		hipError_t e = hipFuncSetAttribute((const void*)plan_superblocks, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
		if (e != hipSuccess)
			return e;
	}
	hipLaunchKernelGGL(plan_superblocks, dim3((uint32_t)(s_end - s_begin)), dim3(64), lds, stream, j, s_begin);
	return hipGetLastError();
}

hipError_t stenos_k_launch_scan(const FrameJob& j, uint64_t s_begin, uint64_t s_end, uint64_t* carry, hipStream_t stream)
{
	if (s_end <= s_begin)
		return hipSuccess;
	hipLaunchKernelGGL(scan_superblocks, dim3(1), dim3(1024), 0, stream, j.sb_csize, s_begin, s_end - s_begin, carry, j.sb_off, j.total);
	return hipGetLastError();
}

hipError_t stenos_k_launch_keep_superblocks(const uint8_t* keep, uint8_t* code, uint32_t* csize, uint64_t nsb, hipStream_t stream)
{
	if (nsb == 0)
		return hipSuccess;
	hipLaunchKernelGGL(keep_superblocks, dim3((uint32_t)((nsb + 255) / 256)), dim3(256), 0, stream, keep, code, csize, nsb);
	return hipGetLastError();
}

hipError_t stenos_k_launch_resolve(const FrameJob& j, hipStream_t stream)
{
	if (j.T > STENOS_K_LDS_MAX_T)
		return stenos_kw_launch_resolve(j, stream);
	const size_t lds = stenos_k_encode_lds_bytes(j.T);
	hipError_t e = hipFuncSetAttribute((const void*)resolve_frame, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
	if (e != hipSuccess)
		return e;
	hipLaunchKernelGGL(resolve_frame, dim3(1), dim3(64), lds, stream, j);
	return hipGetLastError();
}

hipError_t stenos_k_launch_pack(const FrameJob& j, uint64_t s_begin, uint64_t s_end, hipStream_t stream)
{
	if (s_end <= s_begin)
		return hipSuccess;
	hipLaunchKernelGGL(pack_frame, dim3((uint32_t)((s_end - s_begin) * PACK_WAVES)), dim3(64), pack_lds_bytes(j.bps), stream, j, s_begin);
	return hipGetLastError();
}
