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
// Decode pipeline:
//   walk_superblocks   (only without an index) serial walk of the [code][csize:3] chain (stenos.cpp:1129-1134)
//   decode_superblocks one wavefront per superblock (block_compress.h:2088-2175)
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
						    uint32_t* __restrict__ binfo)
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
#ifndef STENOS_FUSED_TICKETS
#define STENOS_FUSED_TICKETS 1
#endif
constexpr uint32_t FUSED_TICKETS = STENOS_FUSED_TICKETS; // superblocks per encoder workgroup
#ifndef STENOS_FUSED_OCCUPANCY
#define STENOS_FUSED_OCCUPANCY 8
#endif
constexpr uint32_t FUSED_OCCUPANCY = STENOS_FUSED_OCCUPANCY; // waves per SIMD the register allocation of the fused kernel aims at

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
#ifdef STENOS_EXP_STATS
	uint64_t rounds = 0, empty = 0;
#endif
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
#ifdef STENOS_EXP_STATS
			++empty;
#endif
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
#ifdef STENOS_EXP_STATS
		++rounds;
#endif
	}
	if (lane == 0) { // what scan_superblocks leaves behind for the ranges that follow
#ifdef STENOS_EXP_STATS
		j.sb_off[nsb + 1] = rounds | (empty << 32);
#endif
		*carry = running;
		j.sb_off[nsb] = running;
		*j.total = running;
	}
}

// offset of superblock s once the scanner has published it (every lane polls the same word)
__device__ uint64_t chain_wait(const FrameJob& j, uint64_t s)
{
#ifdef STENOS_EXP_NOWAIT
	return j.header_bytes + s * 60000ull;
#endif
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

// Workgroup 0: the scanner.  Every other workgroup: FUSED_WAVES wavefronts that take FUSED_TICKETS superblocks one
// after the other -- encode (each wave a run of consecutive blocks into its staging stream), publish the size,
// then store the previous superblock at its offset (pipeline.h, fused_store).
template <uint32_t TT>
__global__ __launch_bounds__(64 * FUSED_WAVES, FUSED_OCCUPANCY) void encode_superblocks(FrameJob j, uint64_t nsb, uint8_t* __restrict__ stage, uint32_t run_cap,
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
	volatile uint32_t* shared = (volatile uint32_t*)(g_lds + FUSED_WAVES * L.total); // [0] ticket, [8 + 4*parity ..] run sizes
	uint32_t b0, b1;
	fused_run_range(j.bps, w, &b0, &b1);
	uint64_t prev = CHAIN_FAILED; // superblock that is encoded but not stored yet
	uint32_t prev_run[FUSED_WAVES];
	for (uint32_t it = 0; it < FUSED_TICKETS; ++it) {
		if (threadIdx.x == 0)
			shared[0] = atomicAdd(ticket, 1u);
		__syncthreads();
		// readfirstlane yields an int: go through uint32_t or values beyond 2^31 get sign-extended
		const uint64_t s = (uint32_t)__builtin_amdgcn_readfirstlane(shared[0]);
		const bool work = s < nsb;
		volatile uint32_t* runs = shared + 8 + 4 * (it & 1u);
		if (work) {
			uint8_t* stage_w = stage + (s * FUSED_WAVES + w) * (uint64_t)run_cap;
#ifdef STENOS_EXP_STATS
			const uint64_t te = __builtin_readcyclecounter();
#endif
			const uint32_t n = encode_run(g_lds + w * L.total, L, T, j.src + (s * j.bps + b0) * (uint64_t)(256 * T), b1 - b0, stage_w);
			if ((threadIdx.x & 63u) == 0)
				runs[w] = n;
#ifdef STENOS_EXP_STATS
			if (threadIdx.x == 0)
				atomicAdd((unsigned long long*)(j.sb_off + nsb + 6), (unsigned long long)(__builtin_readcyclecounter() - te));
#endif
		}
		__syncthreads();
		uint32_t run_size[FUSED_WAVES];
		if (work) {
			for (uint32_t k = 0; k < FUSED_WAVES; ++k)
				run_size[k] = (uint32_t)__builtin_amdgcn_readfirstlane(runs[k]);
			uint32_t code;
			const uint32_t bytes = fused_superblock_size(j, run_size, &code);
			if (threadIdx.x == 0)
				chain_put(size + s, bytes);
		}
		if (prev != CHAIN_FAILED) {
#ifdef STENOS_EXP_STATS
			const uint64_t t0 = __builtin_readcyclecounter();
#endif
			const uint64_t off = chain_wait(j, prev);
#ifdef STENOS_EXP_STATS
			const uint64_t t1 = __builtin_readcyclecounter();
#endif
			if (off == CHAIN_FAILED)
				return;
			fused_store(j, prev, w, off, prev_run, stage + (prev * FUSED_WAVES + w) * (uint64_t)run_cap);
#ifdef STENOS_EXP_STATS
			if (threadIdx.x == 0) {
				atomicAdd((unsigned long long*)(j.sb_off + nsb + 2), (unsigned long long)(t1 - t0));
				atomicAdd((unsigned long long*)(j.sb_off + nsb + 3), (unsigned long long)(__builtin_readcyclecounter() - t1));
			}
#endif
		}
		prev = work ? s : CHAIN_FAILED;
		for (uint32_t k = 0; k < FUSED_WAVES; ++k)
			prev_run[k] = run_size[k];
		if (!work)
			break;
	}
	if (prev != CHAIN_FAILED) {
#ifdef STENOS_EXP_STATS
		const uint64_t t0 = __builtin_readcyclecounter();
#endif
		const uint64_t off = chain_wait(j, prev);
#ifdef STENOS_EXP_STATS
		const uint64_t t1 = __builtin_readcyclecounter();
#endif
		if (off != CHAIN_FAILED)
			fused_store(j, prev, w, off, prev_run, stage + (prev * FUSED_WAVES + w) * (uint64_t)run_cap);
#ifdef STENOS_EXP_STATS
		if (threadIdx.x == 0) {
			atomicAdd((unsigned long long*)(j.sb_off + nsb + 4), (unsigned long long)(t1 - t0));
			atomicAdd((unsigned long long*)(j.sb_off + nsb + 5), (unsigned long long)(__builtin_readcyclecounter() - t1));
		}
#endif
	}
}

// ---- streaming path (bytesoftype 2 and 4) ------------------------------------------------------------------
// Workgroup 0: the scanner.  Every other wavefront takes units (pipeline.h) in frame order, each on its own: it encodes a
// unit into one of its two LDS images, publishes the bytes it takes (agg[unit], non-zero) and counts itself in done[s]
// (units published, high half; their bytes, low half).  The unit that completes the count hands the superblock's bytes
// in the frame to the scanner (size[s]).  Once all units of its superblock are counted -- one word to poll -- a wavefront
// reads their words, which gives it the bytes in front of its own and the superblock's total (BLOCK or COPY,
// stenos.cpp:609-610), and with the superblock's frame offset from the scanner (sb_off[s]) the image goes straight to its
// place.  Nothing is staged in HBM.  A wavefront stores a unit after it has encoded the next one into its other image,
// by which time what it needs to know has normally arrived; it polls in earnest only when both images are full.
//
// Progress.  Tickets are drawn from `shards` counters (a single one saturates near 90 tickets per microsecond): counter
// c hands out the units c, c + shards, c + 2 * shards, ... to the wavefronts whose number (among the encoders) is c
// modulo shards.  A wavefront never waits while it holds a ticket whose unit it has not published: between drawing a
// ticket and publishing the unit it only probes, once, whether its older unit can be stored.  It waits for units of its
// own superblock (at most 63 ahead of its own) and for superblocks before it.  Take the smallest unit nobody has drawn:
// the wavefronts of its counter all hold smaller units of that counter, the holder of the smallest of those waits only
// for units below the undrawn one, which are all drawn and get published without waiting, so it finishes and draws
// the next.  Hence every wait ends as long as every counter has a running wavefront, which the grid size guarantees;
// a poll counter bounds the waits anyway and reports a stall as an error.
constexpr uint32_t STREAM_WAVES = 4;
constexpr uint32_t STREAM_MAX_SHARDS = 128;
#ifndef STENOS_STREAM_OCCUPANCY
#define STENOS_STREAM_OCCUPANCY 7
#endif
constexpr uint32_t AGG_READY = 0x80000000u;

__device__ inline void agg_put(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline uint32_t agg_get(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline uint64_t uniform64(uint64_t v)
{
	return (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)v) | ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(v >> 32)) << 32);
}

template <uint32_t TT>
__global__ __launch_bounds__(64 * STREAM_WAVES, STENOS_STREAM_OCCUPANCY) void encode_stream(FrameJob j, uint64_t nsb, uint32_t* __restrict__ agg, uint64_t* __restrict__ done,
											   uint64_t* __restrict__ size, uint32_t* __restrict__ tickets, uint64_t* __restrict__ carry,
											   uint32_t shards)
{
	if (blockIdx.x == 0) {
		if (threadIdx.x < 64)
			chain_scanner(j, nsb, size, carry);
		return;
	}
	constexpr uint32_t T = TT;
	const Layout L = make_unit_layout(T);
	const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const uint32_t lane = threadIdx.x & 63u;
	Lds lds = g_lds + w * L.total;
	constexpr uint32_t ups = 64; // units per superblock (pipeline.h, stream_supported)
	const uint64_t nunits = nsb * ups;
	const uint32_t shard = ((blockIdx.x - 1) * STREAM_WAVES + w) % shards;
	uint32_t* const my_tickets = tickets + shard * 32u; // a cache line per counter

	// the unit that is encoded and published but not stored yet
	uint64_t pend_unit = ~0ull;
	uint32_t pend_n = 0, pend_image = 0;
	// What storing the pending unit takes, requested in one go: lane 0 the count of its superblock, lane 1 the superblock's
	// frame offset; every lane the word of one unit of the superblock.
	struct Probe {
		uint64_t word;
		uint32_t unit_word;
	};
	auto probe = [&]() -> Probe {
		const uint64_t s = pend_unit / ups;
		Probe p;
		p.word = chain_get((lane == 1 ? j.sb_off : done) + s);
		p.unit_word = agg_get(agg + s * ups + lane);
		return p;
	};
	auto store_if_ready = [&](const Probe& p) -> bool {
		const uint64_t s = pend_unit / ups;
		const uint32_t i = (uint32_t)(pend_unit % ups);
		const uint32_t counted = readlane((uint32_t)(p.word >> 32), 0);
		const uint64_t off = (uint64_t)readlane((uint32_t)p.word, 1) | ((uint64_t)readlane((uint32_t)(p.word >> 32), 1) << 32);
		// (a word that has not arrived although its unit is counted shows by its missing flag)
		if (counted != ups || off == 0 || __builtin_amdgcn_ballot_w64(p.unit_word == 0) != 0)
			return false;
		const uint32_t bytes = p.unit_word & ~AGG_READY;
		const uint32_t incl = wave_incl_scan(bytes);
		unit_store(j, lds, unit_image(L, T, pend_image), s, i, off, readlane(incl - bytes, i), readlane(incl, 63), pend_n);
		pend_unit = ~0ull;
		return true;
	};
	// A ticket may be drawn one unit ahead (its latency then hides behind the encoding) when the units a wavefront can wait
	// for, those below the end of its pending unit's superblock, lie below the unit drawn ahead: 2 * shards >= ups.
	const bool ahead = shards * 2 >= ups;
	auto draw = [&]() -> uint32_t { return lane == 0 ? atomicAdd(my_tickets, 1u) : 0u; };
	auto unit_of = [&](uint32_t t) -> uint64_t { return (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane(t) * shards + shard; };
#ifdef STENOS_EXP_STATS
	uint64_t st_enc = 0, st_pub = 0, st_poll = 0, st_store = 0, st_polls = 0, st_units = 0, st_blocked = 0, st_t0 = 0;
#define ST_BEGIN st_t0 = __builtin_readcyclecounter()
#define ST_END(x) x += __builtin_readcyclecounter() - st_t0
#else
#define ST_BEGIN
#define ST_END(x)
#endif
	uint64_t unit = unit_of(draw());
	while (unit < nunits) {
		uint32_t t_next = 0;
		if (ahead)
			t_next = draw();
		Probe p;
		p.word = 0;
		p.unit_word = 0;
		if (pend_unit != ~0ull)
			p = probe(); // answered while the unit is encoded
		const uint32_t image = pend_unit != ~0ull ? pend_image ^ 1u : 0u;
		ST_BEGIN;
		const uint32_t n = encode_unit(lds, unit_image(L, T, image), T, j.src + unit * (uint64_t)UNIT_BYTES, unit_blocks(T));
		ST_END(st_enc);
		ST_BEGIN;
		if (lane == 0) {
			agg_put(agg + unit, n | AGG_READY);
			// The unit that completes its superblock's count hands the superblock's bytes in the frame to the scanner, right
			// here, where it waits for nothing: the chain of offsets must not run through anybody's waiting.
			const uint64_t before = __hip_atomic_fetch_add(done + unit / ups, (1ull << 32) | n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			if ((uint32_t)(before >> 32) == ups - 1) {
				const uint32_t total = (uint32_t)before + n;
				chain_put(size + unit / ups, 4ull + (total > j.sb_bytes ? j.sb_bytes : total));
			}
		}
		ST_END(st_pub);
#ifdef STENOS_EXP_STATS
		++st_units;
#endif
		ST_BEGIN;
		const bool stored_at_once = pend_unit == ~0ull || store_if_ready(p);
		ST_END(st_store);
		if (!stored_at_once) { // both images are full: poll
#ifdef STENOS_EXP_STATS
			++st_blocked;
			const uint64_t tp = __builtin_readcyclecounter();
#endif
			uint32_t spins = 0;
			for (;;) {
				__builtin_amdgcn_s_sleep(4);
#ifdef STENOS_EXP_STATS
				++st_polls;
#endif
				if (store_if_ready(probe())) {
#ifdef STENOS_EXP_STATS
					st_poll += __builtin_readcyclecounter() - tp;
#endif
					break;
				}
				if (++spins > CHAIN_SPIN_LIMIT) {
					if (lane == 0)
						atomicOr(j.status, ENCODE_STATUS_CHAIN_TIMEOUT);
					return;
				}
			}
		}
		pend_unit = unit;
		pend_n = n;
		pend_image = image;
		unit = unit_of(ahead ? t_next : draw());
	}
	for (uint32_t spins = 0; pend_unit != ~0ull; ++spins) {
		if (store_if_ready(probe()))
			break;
		if (spins > CHAIN_SPIN_LIMIT) {
			if (lane == 0)
				atomicOr(j.status, ENCODE_STATUS_CHAIN_TIMEOUT);
			return;
		}
		__builtin_amdgcn_s_sleep(4);
	}
#ifdef STENOS_EXP_STATS
	if (lane == 0) {
		unsigned long long* st = (unsigned long long*)(j.sb_off + nsb + 2);
		atomicAdd(st + 0, st_enc);
		atomicAdd(st + 1, st_pub);
		atomicAdd(st + 2, st_poll);
		atomicAdd(st + 3, st_store);
		atomicAdd(st + 4, st_polls);
		atomicAdd(st + 5, st_units);
		atomicAdd(st + 6, st_blocked);
	}
#endif
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

// One wavefront replays the capacity rules where the parallel plan could not clear them.
__global__ __launch_bounds__(64) void resolve_frame(FrameJob j)
{
	const Layout L = make_layout(j.T, true);
	resolve_capacity(g_lds, L, j);
}

__global__ __launch_bounds__(64) void pack_frame(FrameJob j, uint64_t s_begin) { pack_superblock(g_lds, j, s_begin + blockIdx.x / PACK_WAVES, blockIdx.x % PACK_WAVES); }

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

template <uint32_t TT>
__global__ __launch_bounds__(64, 8) void decode_superblocks(DecodeArgs a)
{
	const uint32_t T = TT ? TT : a.T;
	const uint64_t s = a.sb_ids ? a.sb_ids[blockIdx.x] : blockIdx.x;
	const U32 lane = lane_id();
	const uint64_t p = a.sb_off[blockIdx.x];
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
		const DecLayout L = make_dec_layout(T);
		uint32_t r = decode_superblock(g_lds, L, T, payload, csize, out, dsize);
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

// number of CUs of the current device and how many one-wave workgroups with `lds` bytes each stay resident on one
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
uint32_t stenos_k_waves_per_cu(size_t lds)
{
	const char* env = getenv("STENOS_WAVES_PER_CU");
	if (env && atoi(env) > 0)
		return (uint32_t)atoi(env);
	size_t by_lds = lds ? (160u * 1024u) / lds : 32;
	return (uint32_t)(by_lds > 32 ? 32 : (by_lds < 1 ? 1 : by_lds));
}

size_t stenos_k_encode_lds_bytes(uint32_t T) { return make_layout(T, true).total; }
size_t stenos_k_decode_lds_bytes(uint32_t T) { return make_dec_layout(T).total; }
uint32_t stenos_k_slot_stride(uint32_t T) { return out_capacity(T); }

template <uint32_t TT>
static hipError_t launch_encode_t(const FrameJob& j, uint64_t b_begin, uint64_t b_end, hipStream_t stream)
{
	size_t lds = stenos_k_encode_lds_bytes(j.T);
	if (getenv("STENOS_EXP_SMALL_LDS"))
		lds = (size_t)atoi(getenv("STENOS_EXP_SMALL_LDS"));
	hipError_t e = hipFuncSetAttribute((const void*)encode_blocks<TT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
	if (e != hipSuccess)
		return e;
	// One workgroup per block measured faster than a persistent grid on MI355X (10.4 ms vs 12.5-14.8 ms for
	// 8 GiB of int32): the dispatcher staggers the waves, a resident grid runs them in phase.  The
	// grid-stride form stays available for experiments through STENOS_WAVES_PER_CU.
	const uint64_t nblocks = b_end - b_begin;
	uint32_t grid = (uint32_t)nblocks;
	if (getenv("STENOS_WAVES_PER_CU")) {
		const uint64_t resident = (uint64_t)stenos_k_cu_count() * stenos_k_waves_per_cu(lds);
		grid = (uint32_t)(nblocks < resident ? nblocks : resident);
	}
	hipLaunchKernelGGL(encode_blocks<TT>, dim3(grid), dim3(64), lds, stream, j.src, b_begin, b_end, j.nfull, j.tail_bytes, j.T, j.slots, j.slot_stride,
			   j.bsize, j.binfo);
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
	const size_t lds = FUSED_WAVES * stenos_k_encode_lds_bytes(j.T) + 64;
	hipError_t e = hipFuncSetAttribute((const void*)encode_superblocks<TT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
	if (e != hipSuccess)
		return e;
	const uint32_t grid = (uint32_t)((nsb + FUSED_TICKETS - 1) / FUSED_TICKETS) + 1; // + the scanner
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
// Superblocks [0, nsb) of the job (stream_supported), all of them full blocks with room for any encoding.  agg: nsb * 64
// zeroed words, done and size: nsb zeroed words each, tickets: STREAM_MAX_SHARDS * 32 zeroed words, j.sb_off[0, nsb] zeroed; *carry
// receives the frame offset behind them.
template <uint32_t TT>
static hipError_t launch_stream_t(const FrameJob& j, uint64_t nsb, uint32_t* agg, uint64_t* done, uint64_t* size, uint32_t* tickets, uint64_t* carry, hipStream_t stream)
{
	const size_t lds = STREAM_WAVES * make_unit_layout(TT).total + 64;
	hipError_t e = hipFuncSetAttribute((const void*)encode_stream<TT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
	if (e != hipSuccess)
		return e;
	const uint64_t groups_wanted = (nsb * 64 + STREAM_WAVES - 1) / STREAM_WAVES;
	const uint64_t resident = (uint64_t)stenos_k_cu_count() * 8;
	const uint32_t groups = (uint32_t)(groups_wanted < resident ? groups_wanted : resident);
	const uint32_t shards = groups * STREAM_WAVES < STREAM_MAX_SHARDS ? groups * STREAM_WAVES : STREAM_MAX_SHARDS;
	hipLaunchKernelGGL(encode_stream<TT>, dim3(groups + 1), dim3(64 * STREAM_WAVES), lds, stream, j, nsb, agg, done, size, tickets, carry, shards);
	return hipGetLastError();
}
hipError_t stenos_k_launch_encode_stream(const FrameJob& j, uint64_t nsb, uint32_t* agg, uint64_t* done, uint64_t* size, uint32_t* tickets, uint64_t* carry,
					 hipStream_t stream)
{
	if (nsb == 0)
		return hipSuccess;
	return j.T == 2 ? launch_stream_t<2>(j, nsb, agg, done, size, tickets, carry, stream) : launch_stream_t<4>(j, nsb, agg, done, size, tickets, carry, stream);
}
bool stenos_k_stream_supported(uint32_t T, uint32_t bps) { return stream_supported(bps, T); }
size_t stenos_k_stream_words(uint64_t nsb) { return (size_t)nsb * 64 + STREAM_MAX_SHARDS * 32; } // agg + tickets, 32-bit words

// the workgroup's scratch must fit the 160 KiB of a CU (bytesoftype up to about 40)
bool stenos_k_fused_supported(uint32_t T) { return FUSED_WAVES * stenos_k_encode_lds_bytes(T) + 64 <= 160u * 1024u; }
size_t stenos_k_fused_stage_bytes(uint32_t T, uint32_t bps, uint64_t nsb) { return (size_t)nsb * FUSED_WAVES * fused_run_capacity(bps, T) + 64; }

// blocks [b_begin, b_end) of the job (the tail block has index nfull)
hipError_t stenos_k_launch_encode(const FrameJob& j, uint64_t b_begin, uint64_t b_end, hipStream_t stream)
{
	if (b_end <= b_begin)
		return hipSuccess;
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

hipError_t stenos_k_launch_resolve(const FrameJob& j, hipStream_t stream)
{
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

hipError_t stenos_k_launch_walk(const uint8_t* frame, uint64_t size, uint64_t first, uint64_t nsb, uint64_t* off, uint32_t* status, hipStream_t stream)
{
	hipLaunchKernelGGL(walk_superblocks, dim3(1), dim3(64), 0, stream, frame, size, first, nsb, off, status);
	return hipGetLastError();
}

template <uint32_t TT>
static hipError_t launch_decode_t(const DecodeArgs& a, hipStream_t stream)
{
	size_t lds = stenos_k_decode_lds_bytes(a.T);
	if (getenv("STENOS_EXP_DEC_LDS")) // occupancy experiments: a larger allocation leaves fewer waves per SIMD
		lds = (size_t)atoi(getenv("STENOS_EXP_DEC_LDS"));
	hipError_t e = hipFuncSetAttribute((const void*)decode_superblocks<TT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
	if (e != hipSuccess)
		return e;
	hipLaunchKernelGGL(decode_superblocks<TT>, dim3((uint32_t)a.nsb), dim3(64), lds, stream, a);
	return hipGetLastError();
}

hipError_t stenos_k_launch_decode(const DecodeArgs& a, hipStream_t stream)
{
	switch (a.T) {
		case 2: return launch_decode_t<2>(a, stream);
		case 4: return launch_decode_t<4>(a, stream);
		case 8: return launch_decode_t<8>(a, stream);
		default: return launch_decode_t<0>(a, stream);
	}
}
