// walk.h -- the superblock chain of a frame, found in parallel.
//
// A frame is [header] then per superblock [code][csize:3][payload of csize bytes] (stenos.cpp:862-907); the decoder finds
// superblock s by walking the chain of headers from the first one (stenos.cpp:1126-1134, 1166-1182).  One lane doing that
// on a device-resident frame pays a dependent HBM read per superblock (0.35 us each: 23 ms for the 65 536 superblocks of
// 8 GiB of int32, ten times the decode kernel).  Here the chain is cut into segments of the frame that are walked
// concurrently, and the result is EXACTLY the serial walk's:
//
//   A  speculate  segment k (one wavefront) looks at every byte position of the window [B_k, B_k + W), W = superblock
//                 size + 4 = the longest hop, for headers that look like one (code 1..6, csize <= superblock size) and hop
//                 out of the window: the "roots".  The last header of the true chain inside the window is one of them.
//                 All roots are followed in lock step, one per lane, while the headers they land on look like headers.
//                 Chains that start inside payload bytes die within a hop or two (a random position looks like a header
//                 with probability 2e-4); the survivor's root r_k, exit x_k (first position at or behind B_k+1) and number
//                 of hops are recorded (several survivors that have merged into one chain, up to four, are all kept).  Segment 0 starts at the first header and needs no window.
//   B  verify     segment k hops from x_k-1 -- plain hops, nothing is assumed about the bytes -- until it reaches r_k or
//                 passes it.  Reaching it proves by induction that x_k-1 -> ... -> r_k -> ... -> x_k is the serial chain.
//                 The hops are counted.  The last segment is walked to the end of the frame (no speculation).
//   C  write      segment k sums the counts in front of it and hops once more from x_k-1, writing the offsets.
//
// Anything else -- no survivor, survivors that disagree, a walk that passes its root, a chain that ends early -- raises a flag and the
// serial walk runs after all (frames that are malformed, or that hold whole Stenos frames inside stored superblocks
// aligned just so; never a wrong index).
//
// Written as per-lane functions for host and device: tests/emul runs them lane by lane against the serial walk.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define WALK_HD __host__ __device__ inline
#else
#define WALK_HD inline
#endif

namespace walk {

#ifndef STENOS_WALK_MAX_SEGMENTS
#define STENOS_WALK_MAX_SEGMENTS 2048
#endif
#ifndef STENOS_WALK_SEG_WINDOWS
#define STENOS_WALK_SEG_WINDOWS 4
#endif
constexpr uint32_t MAX_SEGMENTS = STENOS_WALK_MAX_SEGMENTS;
constexpr uint32_t LANES = 64;
constexpr uint32_t MIN_SEGMENTS = 4;       // below, the serial walk
constexpr uint32_t SEG_WINDOWS = STENOS_WALK_SEG_WINDOWS; // a segment is at most this many windows long, unless the frame needs more than MAX_SEGMENTS of them
constexpr uint32_t PROOF_HOPS = 2;         // plausible headers a speculated chain must meet behind its segment
enum : uint32_t { SEG_UNRESOLVED = 0, SEG_OK = 1 };
enum : uint32_t { WALK_FAILED = 1 };

struct Plan {
	uint64_t first, size, nsb; // offset of the first superblock header, frame bytes, superblocks the frame header announces
	uint64_t seg_len;          // bytes per segment
	uint32_t nseg;             // 0: serial walk
	uint32_t window;           // W
	uint32_t sb_bytes;
};
constexpr uint32_t MAX_ROOTS = 4; // speculated chains of a segment that survive (they have merged: one exit)
struct Segment {
	uint64_t exit;
	uint64_t count; // phase B: chain positions inside the segment
	uint32_t state, nroots;
	uint64_t root[MAX_ROOTS];
	uint32_t hops[MAX_ROOTS];
};

WALK_HD uint64_t seg_begin(const Plan& P, uint32_t k) { return P.first + (uint64_t)k * P.seg_len; }
WALK_HD uint64_t seg_end(const Plan& P, uint32_t k) { return k + 1 >= P.nseg ? ~0ull : P.first + (uint64_t)(k + 1) * P.seg_len; }

// seg_len_override: tests force short segments
inline Plan make_plan(uint64_t first, uint64_t size, uint64_t nsb, uint32_t sb_bytes, uint64_t seg_len_override = 0)
{
	Plan P;
	P.first = first;
	P.size = size;
	P.nsb = nsb;
	P.sb_bytes = sb_bytes;
	P.window = sb_bytes + 4;
	P.nseg = 0;
	P.seg_len = 0;
	if (size <= first || nsb < 2 * MIN_SEGMENTS)
		return P;
	const uint64_t span = size - first;
	// Segment length: a few windows.  (One window per segment would suit frames of small payloads -- sorted keys: 2 KB per
	// 128 KiB superblock, 230 serial hops per segment in phases B and C -- but then a false chain only has to survive a hop or
	// two, and structured payloads hold enough header look-alikes for several to survive with different exits: measured, the
	// 8 GiB sorted frame then falls back to the serial walk.)
	uint64_t len = seg_len_override ? seg_len_override : (uint64_t)P.window * SEG_WINDOWS;
	if (len < P.window)
		len = P.window;
	if ((span + len - 1) / len > MAX_SEGMENTS)
		len = (span + MAX_SEGMENTS - 1) / MAX_SEGMENTS;
	len = (len + 15) & ~15ull;
	const uint64_t n = (span + len - 1) / len;
	if (n < MIN_SEGMENTS)
		return P;
	P.seg_len = len;
	P.nseg = (uint32_t)n;
	return P;
}

WALK_HD bool readable(const Plan& P, uint64_t p) { return p <= P.size && P.size - p >= 4; }
typedef uint32_t __attribute__((aligned(1))) u32_unaligned;
WALK_HD uint32_t header_word(const uint8_t* frame, uint64_t p) { return *(const u32_unaligned*)(frame + p); } // little endian: code | csize << 8
WALK_HD bool plausible(const Plan& P, uint32_t w) { return (w & 0xFFu) - 1u < 6u && (w >> 8) <= P.sb_bytes; }
WALK_HD uint64_t hop(uint64_t p, uint32_t w) { return p + 4 + (uint64_t)(w >> 8); }

// ---- phase A ----------------------------------------------------------------------------------------------
// Positions [base, base + 16) of segment k's window: appends the roots among them to roots[0..64) (count in *nroots, which
// may run past 64: the caller then gives the segment up).  add(counter) must return the old value and add one atomically.
template <class Add>
WALK_HD void scan_window16(const Plan& P, const uint8_t* frame, uint32_t k, uint64_t base, uint64_t* roots, uint32_t* nroots, Add add)
{
	const uint64_t wend = seg_begin(P, k) + P.window;
	uint32_t w[5];
	if (base + 20 <= P.size) {
		for (int i = 0; i < 5; ++i)
			w[i] = header_word(frame, base + 4 * (uint64_t)i);
	}
	else {
		for (int i = 0; i < 5; ++i) {
			w[i] = 0;
			for (int b = 0; b < 4; ++b) {
				const uint64_t q = base + 4 * (uint64_t)i + (uint64_t)b;
				if (q < P.size)
					w[i] |= (uint32_t)frame[q] << (8 * b);
			}
		}
	}
	// First a filter on the sixteen code bytes at once: a byte b is a candidate when b - 1 < 8 (codes 1..6 and two more:
	// the price of doing it on four bytes per instruction); 3 % of the bytes of a payload pass and get the full test.
	uint32_t cand = 0;
	for (int i = 0; i < 4; ++i) {
		const uint32_t y = (w[i] | 0x80808080u) - 0x01010101u;                   // per byte (b - 1) mod 128, no borrow between bytes
		const uint32_t z = (y & 0x78787878u) | (w[i] & 0x80808080u);             // zero byte: b in 1..8
		const uint32_t nz = ((z & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | z;               // bit 7 of every byte: the byte is not zero
		const uint32_t f = ~nz & 0x80808080u;
		cand |= (((f >> 7) & 1u) | ((f >> 14) & 2u) | ((f >> 21) & 4u) | ((f >> 28) & 8u)) << (4 * i);
	}
	while (cand) {
		const uint32_t j = (uint32_t)__builtin_ctz(cand);
		cand &= cand - 1u;
		const uint64_t p = base + j;
		if (p >= wend || !readable(P, p))
			break;
		const uint32_t sh = (j & 3u) * 8u;
		const uint32_t h = sh ? (w[j >> 2] >> sh) | (w[(j >> 2) + 1] << (32u - sh)) : w[j >> 2];
		if (plausible(P, h) && hop(p, h) >= wend) {
			// (payload bytes look like a header two hundred times per window of 128 KiB: where they point must look like one too)
			const uint64_t q = hop(p, h);
			if (readable(P, q) ? !plausible(P, header_word(frame, q)) : q != P.size)
				continue;
			const uint32_t idx = add(nroots);
			if (idx < LANES)
				roots[idx] = p;
		}
	}
}
// one lane follows one root while the headers look like headers; false: the chain broke
WALK_HD bool follow_root(const Plan& P, const uint8_t* frame, uint32_t k, uint64_t root, uint64_t* exit, uint32_t* hops)
{
	const uint64_t end = seg_end(P, k);
	uint64_t p = root;
	uint32_t n = 0;
	while (p < end && readable(P, p)) {
		const uint32_t w = header_word(frame, p);
		if (!plausible(P, w))
			return false;
		p = hop(p, w);
		++n;
	}
	*exit = p;
	*hops = n;
	// a chain that leaves the segment after a hop or two has proven little: the next headers must look like headers too
	for (uint32_t extra = 0; extra < PROOF_HOPS && readable(P, p); ++extra) {
		const uint32_t w = header_word(frame, p);
		if (!plausible(P, w))
			return false;
		p = hop(p, w);
	}
	// ... and a chain that ends, ends with the frame (a hop out of a truncated frame proves nothing: the serial walk sorts that out)
	return readable(P, p) || p == P.size;
}
// segment 0: plain hops from the first header
WALK_HD void follow_first(const Plan& P, const uint8_t* frame, Segment* s)
{
	const uint64_t end = seg_end(P, 0);
	uint64_t p = P.first;
	uint32_t n = 0;
	while (p < end && readable(P, p) && n <= P.nsb) {
		p = hop(p, header_word(frame, p));
		++n;
	}
	s->nroots = 1;
	s->root[0] = P.first;
	s->exit = p;
	s->hops[0] = n;
	s->state = p >= end || !readable(P, p) ? SEG_OK : SEG_UNRESOLVED;
	s->count = n;
}

// ---- phase B ----------------------------------------------------------------------------------------------
// false: the speculation does not hold, the serial walk has to run
WALK_HD bool verify_segment(const Plan& P, const uint8_t* frame, uint32_t k, Segment* seg)
{
	if (k == 0)
		return seg[0].state == SEG_OK;
	if (seg[k - 1].state != SEG_OK)
		return false;
	uint64_t p = seg[k - 1].exit;
	uint64_t n = 0;
	if (k + 1 < P.nseg) {
		if (seg[k].state != SEG_OK || p < seg_begin(P, k))
			return false;
		// the roots that survived have merged into one chain: the true chain meets exactly one of them
		uint64_t last = 0;
		for (uint32_t i = 0; i < seg[k].nroots; ++i)
			last = seg[k].root[i] > last ? seg[k].root[i] : last;
		for (;;) {
			for (uint32_t i = 0; i < seg[k].nroots; ++i)
				if (p == seg[k].root[i]) {
					seg[k].count = n + seg[k].hops[i];
					return true;
				}
			if (p > last || !readable(P, p))
				return false;
			p = hop(p, header_word(frame, p));
			++n;
		}
	}
	// the last segment: to the end of the frame, or until the index is full
	if (p < seg_begin(P, k))
		return false;
	while (readable(P, p) && n <= P.nsb) {
		p = hop(p, header_word(frame, p));
		++n;
	}
	seg[k].nroots = 1;
	seg[k].root[0] = seg[k - 1].exit;
	seg[k].exit = p;
	seg[k].hops[0] = (uint32_t)n;
	seg[k].state = SEG_OK;
	seg[k].count = n;
	return true;
}

// ---- phase C ----------------------------------------------------------------------------------------------
// one lane: the positions of segment k go to off[base ...]; the last segment adds what the serial walk leaves behind the
// chain (stenos.cpp:1126-1127: a chain that ends early is a truncated frame).  Returns the status bits to raise.
WALK_HD uint32_t write_segment(const Plan& P, const uint8_t* frame, uint32_t k, const Segment* seg, uint64_t base, uint64_t* off, uint32_t truncated_bit)
{
	uint64_t p = k ? seg[k - 1].exit : P.first;
	const uint64_t n = seg[k].count;
	for (uint64_t i = 0; i < n && base + i <= P.nsb; ++i) {
		off[base + i] = p;
		p = hop(p, header_word(frame, p));
	}
	if (k + 1 < P.nseg)
		return 0;
	const uint64_t total = base + n; // readable positions of the whole chain (counted up to nsb + 1)
	if (total < P.nsb) {
		for (uint64_t s = total; s <= P.nsb; ++s)
			off[s] = P.size;
		return truncated_bit;
	}
	if (total == P.nsb) {
		off[P.nsb] = p;
		return p > P.size ? truncated_bit : 0;
	}
	return 0;
}

} // namespace walk
