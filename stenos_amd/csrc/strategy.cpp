// strategy.cpp -- see strategy.h
#include "strategy.h"

#include <math.h>
#include <string.h>

namespace strategy {

namespace {
inline uint32_t rd32(const uint8_t* p)
{
	uint32_t v;
	memcpy(&v, p, 4);
	return v; // little endian host
}
inline unsigned hash8(uint32_t seq) { return (seq * 2654435761U) >> 24; } // LZ4_hash4 with a hash log of 8 (lz4dry.cpp:117, 141, 607-613)
} // namespace

size_t lz4_dry_size(const uint8_t* src, size_t n_in, int accel)
{
	enum { MINMATCH = 4, MFLIMIT = 12, LASTLITERALS = 5, MAXD = 65535, ML_MASK = 15, RUN_MASK = 15, SKIP_TRIGGER = 6 };
	if (n_in > 0x7E000000u)
		return 0;
	const int n = (int)n_in;
	uint32_t table[256] = { 0 }; // the stream state is zeroed for every call (lz4dry.cpp:815-835)
	if (accel < 1)
		accel = 1;
	int ip = 0, anchor = 0, count = 0;
	const int mflimit = n - MFLIMIT, matchlimit = n - LASTLITERALS;
	if (n >= MFLIMIT + 1) {
		table[hash8(rd32(src))] = 0;
		ip = 1;
		unsigned fwd = hash8(rd32(src + ip));
		for (;;) {
			int match;
			{ // search with growing steps (lz4dry.cpp:705-722)
				int next = ip;
				unsigned step = 1, tries = (unsigned)accel << SKIP_TRIGGER;
				for (;;) {
					const unsigned h = fwd;
					ip = next;
					next += (int)step;
					step = tries++ >> SKIP_TRIGGER;
					if (next > mflimit)
						goto tail;
					match = (int)table[h];
					fwd = hash8(rd32(src + next));
					table[h] = (uint32_t)ip;
					if (match + MAXD >= ip && rd32(src + match) == rd32(src + ip))
						break;
				}
			}
			while (ip > anchor && match > 0 && src[ip - 1] == src[match - 1]) { // extend backwards (:725-728)
				--ip;
				--match;
			}
			{
				const int lit = ip - anchor; // token + literal length bytes + literals (:731-745)
				count += 1 + lit;
				if (lit >= RUN_MASK)
					count += 1 + (lit - RUN_MASK) / 256;
			}
			for (;;) {
				count += 2; // offset
				int m = 0;
				while (ip + MINMATCH + m < matchlimit && src[ip + MINMATCH + m] == src[match + MINMATCH + m])
					++m;
				ip += MINMATCH + m;
				if (m >= ML_MASK) { // match length bytes (:759-774)
					m -= ML_MASK;
					while (m >= 4 * 255) {
						count += 4;
						m -= 4 * 255;
					}
					count += 1 + m / 255;
				}
				anchor = ip;
				if (ip > mflimit)
					goto tail;
				table[hash8(rd32(src + ip - 2))] = (uint32_t)(ip - 2);
				const unsigned h = hash8(rd32(src + ip));
				match = (int)table[h];
				table[h] = (uint32_t)ip;
				if (match + MAXD >= ip && rd32(src + match) == rd32(src + ip)) { // immediate next match (:784-789)
					++count;
					continue;
				}
				break;
			}
			fwd = hash8(rd32(src + ++ip));
		}
	}
tail: {
	const int last = n - anchor; // last literals (:795-813)
	count += last >= RUN_MASK ? 2 + (last - RUN_MASK) / 256 : 1;
	count += last;
}
	return (size_t)count;
}

size_t middle_step(size_t T, size_t bytes, int level)
{
	const size_t elements = bytes / T;
	size_t step = elements / (size_t)(16 / (level - 1));
	return step < 64 ? elements : step;
}

double transposed_ratio(const uint8_t* planes, size_t T, size_t step, int level)
{
	size_t csize = 0, processed = 0;
	for (size_t i = 0; i < T; ++i) {
		csize += lz4_dry_size(planes + i * step, step, 10 - level);
		processed += step;
	}
	// (one fused multiply-add, as the reference's compilers make of 1 + level * 0.02 on x86-64-v3: the last bit of the factor
	// decides exact ties between the estimate on the transposed input and 1.1 x the one on transposed + delta; oracle and
	// tests/golden/levels_manifest.json, the seeded case)
	return ((double)processed / (double)csize) * fma((double)level, 0.02, 1.);
}

} // namespace strategy
