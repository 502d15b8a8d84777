// wavevec_host.h -- the host side of stenos_amd/csrc/wavevec.h: the one-wavefront vocabulary as 64 lanes executed in
// lockstep by g++ (-DWV_HOST_EMULATION), so that the codec's source can be diffed against the CPU oracle without a GPU.
// Test infrastructure: it lives with the emulation (tests/emul), the product's sources include it only when the emulation
// is being built, and the shipped library never sees it.
#pragma once
// ------------------------------------------------------------------------------------------------
// host emulation: 64 lanes in lockstep
// ------------------------------------------------------------------------------------------------
#include <string.h>
#define WV_FN static inline
#define WV_MFN inline
#define WV_HD static inline
#define WV_TABLE static const
#define WV_MARK(name)
#define WV_NESTED()
namespace wv {
constexpr int WAVE = 64;

struct Pred {
	bool l[WAVE];
};
struct U32 {
	uint32_t l[WAVE];
	U32() {}
	U32(uint32_t s)
	{
		for (int i = 0; i < WAVE; ++i) l[i] = s;
	}
};
typedef uint8_t* Lds;

#define WV_BINOP(op)                                                                                                   \
	WV_FN U32 operator op(const U32& a, const U32& b)                                                                  \
	{                                                                                                                  \
		U32 r;                                                                                                         \
		for (int i = 0; i < WAVE; ++i) r.l[i] = a.l[i] op b.l[i];                                                      \
		return r;                                                                                                      \
	}
WV_BINOP(+) WV_BINOP(-) WV_BINOP(*) WV_BINOP(&) WV_BINOP(|) WV_BINOP(^)
#undef WV_BINOP
WV_FN U32 operator<<(const U32& a, const U32& b)
{
	U32 r;
	for (int i = 0; i < WAVE; ++i) r.l[i] = a.l[i] << (b.l[i] & 31);
	return r;
}
WV_FN U32 operator>>(const U32& a, const U32& b)
{
	U32 r;
	for (int i = 0; i < WAVE; ++i) r.l[i] = a.l[i] >> (b.l[i] & 31);
	return r;
}
WV_FN U32 operator~(const U32& a)
{
	U32 r;
	for (int i = 0; i < WAVE; ++i) r.l[i] = ~a.l[i];
	return r;
}
#define WV_CMP(op)                                                                                                     \
	WV_FN Pred operator op(const U32& a, const U32& b)                                                                 \
	{                                                                                                                  \
		Pred r;                                                                                                        \
		for (int i = 0; i < WAVE; ++i) r.l[i] = a.l[i] op b.l[i];                                                      \
		return r;                                                                                                      \
	}
WV_CMP(==) WV_CMP(!=) WV_CMP(<) WV_CMP(<=) WV_CMP(>) WV_CMP(>=)
#undef WV_CMP
WV_FN Pred operator&(const Pred& a, const Pred& b)
{
	Pred r;
	for (int i = 0; i < WAVE; ++i) r.l[i] = a.l[i] && b.l[i];
	return r;
}
WV_FN Pred operator|(const Pred& a, const Pred& b)
{
	Pred r;
	for (int i = 0; i < WAVE; ++i) r.l[i] = a.l[i] || b.l[i];
	return r;
}
WV_FN Pred operator!(const Pred& a)
{
	Pred r;
	for (int i = 0; i < WAVE; ++i) r.l[i] = !a.l[i];
	return r;
}
WV_FN Pred pred_all(bool v)
{
	Pred r;
	for (int i = 0; i < WAVE; ++i) r.l[i] = v;
	return r;
}
WV_FN U32 sel(const Pred& p, const U32& a, const U32& b)
{
	U32 r;
	for (int i = 0; i < WAVE; ++i) r.l[i] = p.l[i] ? a.l[i] : b.l[i];
	return r;
}
WV_FN U32 lane_id()
{
	U32 r;
	for (int i = 0; i < WAVE; ++i) r.l[i] = (uint32_t)i;
	return r;
}
WV_FN U32 lane_id_plain() { return lane_id(); }
// number of set bits of mask below the lane's own (v_mbcnt): the lane's rank among the lanes of the mask
WV_FN U32 lane_rank(uint64_t mask)
{
	U32 r;
	for (int i = 0; i < WAVE; ++i) r.l[i] = (uint32_t)__builtin_popcountll(mask & ((1ull << i) - 1ull));
	return r;
}
WV_FN U32 umin(const U32& a, const U32& b)
{
	U32 r;
	for (int i = 0; i < WAVE; ++i) r.l[i] = a.l[i] < b.l[i] ? a.l[i] : b.l[i];
	return r;
}
WV_FN U32 umax(const U32& a, const U32& b)
{
	U32 r;
	for (int i = 0; i < WAVE; ++i) r.l[i] = a.l[i] > b.l[i] ? a.l[i] : b.l[i];
	return r;
}
WV_FN U32 popc(const U32& a)
{
	U32 r;
	for (int i = 0; i < WAVE; ++i) r.l[i] = (uint32_t)__builtin_popcount(a.l[i]);
	return r;
}
// min / max of the two 16-bit halves, each half on its own (v_pk_min_u16 / v_pk_max_u16)
WV_FN U32 pk_min_u16(const U32& a, const U32& b)
{
	U32 r;
	for (int i = 0; i < WAVE; ++i) {
		uint32_t lo = (a.l[i] & 0xFFFFu) < (b.l[i] & 0xFFFFu) ? (a.l[i] & 0xFFFFu) : (b.l[i] & 0xFFFFu);
		uint32_t hi = (a.l[i] >> 16) < (b.l[i] >> 16) ? (a.l[i] >> 16) : (b.l[i] >> 16);
		r.l[i] = lo | (hi << 16);
	}
	return r;
}
WV_FN U32 pk_max_u16(const U32& a, const U32& b)
{
	U32 r;
	for (int i = 0; i < WAVE; ++i) {
		uint32_t lo = (a.l[i] & 0xFFFFu) > (b.l[i] & 0xFFFFu) ? (a.l[i] & 0xFFFFu) : (b.l[i] & 0xFFFFu);
		uint32_t hi = (a.l[i] >> 16) > (b.l[i] >> 16) ? (a.l[i] >> 16) : (b.l[i] >> 16);
		r.l[i] = lo | (hi << 16);
	}
	return r;
}
// number of bits needed to represent a (0 for 0)
WV_FN U32 bitlen(const U32& a)
{
	U32 r;
	for (int i = 0; i < WAVE; ++i) r.l[i] = a.l[i] ? 32u - (uint32_t)__builtin_clz(a.l[i]) : 0u;
	return r;
}
// 32x32 -> high 32 bits of the 64-bit product
WV_FN U32 mulhi(const U32& a, const U32& b)
{
	U32 r;
	for (int i = 0; i < WAVE; ++i) r.l[i] = (uint32_t)(((uint64_t)a.l[i] * b.l[i]) >> 32);
	return r;
}
// v_bfe_u32: `width` (< 32) bits of x from bit `off` (< 32) on
WV_FN U32 bfe(const U32& x, const U32& off, const U32& width)
{
	U32 r;
	for (int i = 0; i < WAVE; ++i) r.l[i] = (x.l[i] >> (off.l[i] & 31u)) & ((1u << (width.l[i] & 31u)) - 1u);
	return r;
}
// v_alignbit_b32: the low 32 bits of {hi:lo} >> (sh & 31)
WV_FN U32 funnel_shr(const U32& hi, const U32& lo, const U32& sh)
{
	U32 r;
	for (int i = 0; i < WAVE; ++i) r.l[i] = (uint32_t)((((uint64_t)hi.l[i] << 32) | lo.l[i]) >> (sh.l[i] & 31u));
	return r;
}
// v_dot4_u32_u8: c + a.b0*b.b0 + a.b1*b.b1 + a.b2*b.b2 + a.b3*b.b3
WV_FN U32 dot4_u8(const U32& a, uint32_t b, const U32& c)
{
	U32 r;
	for (int i = 0; i < WAVE; ++i) {
		uint32_t s = c.l[i];
		for (int k = 0; k < 4; ++k)
			s += ((a.l[i] >> (8 * k)) & 0xFFu) * ((b >> (8 * k)) & 0xFFu);
		r.l[i] = s;
	}
	return r;
}
// v_msad_u8: c + the sum over the four bytes of |a.byte - ref.byte|, the bytes where ref.byte is 0 left out
WV_FN U32 msad_u8(const U32& a, const U32& ref, const U32& c)
{
	U32 r;
	for (int i = 0; i < WAVE; ++i) {
		uint32_t s = c.l[i];
		for (int k = 0; k < 4; ++k) {
			const uint32_t x = (a.l[i] >> (8 * k)) & 0xFFu, y = (ref.l[i] >> (8 * k)) & 0xFFu;
			if (y)
				s += x > y ? x - y : y - x;
		}
		r.l[i] = s;
	}
	return r;
}
WV_FN U32 mul24(const U32& a, const U32& b)
{
	U32 r;
	for (int i = 0; i < WAVE; ++i) r.l[i] = (a.l[i] & 0xFFFFFFu) * (b.l[i] & 0xFFFFFFu);
	return r;
}
// v_mad_u32_u24: a * b + c, the low 24 bits of a and b
WV_FN U32 mad24(const U32& a, const U32& b, const U32& c)
{
	U32 r;
	for (int i = 0; i < WAVE; ++i) r.l[i] = (a.l[i] & 0xFFFFFFu) * (b.l[i] & 0xFFFFFFu) + c.l[i];
	return r;
}
// v_perm_b32: result byte i = byte (sel byte i) of the 8 bytes {hi:lo}; selector 0x0c gives 0
WV_FN U32 perm_bytes(const U32& hi, const U32& lo, uint32_t selw)
{
	U32 r;
	for (int i = 0; i < WAVE; ++i) {
		uint64_t both = ((uint64_t)hi.l[i] << 32) | lo.l[i];
		uint32_t v = 0;
		for (int k = 0; k < 4; ++k) {
			uint32_t sb = (selw >> (8 * k)) & 0xFF;
			uint32_t byte = sb < 8 ? (uint32_t)((both >> (8 * sb)) & 0xFF) : 0u;
			v |= byte << (8 * k);
		}
		r.l[i] = v;
	}
	return r;
}
// the same with a selector per lane
WV_FN U32 perm_bytes_v(const U32& hi, const U32& lo, const U32& selw)
{
	U32 r;
	for (int i = 0; i < WAVE; ++i) {
		uint64_t both = ((uint64_t)hi.l[i] << 32) | lo.l[i];
		uint32_t v = 0;
		for (int k = 0; k < 4; ++k) {
			uint32_t sb = (selw.l[i] >> (8 * k)) & 0xFF;
			uint32_t byte = sb < 8 ? (uint32_t)((both >> (8 * sb)) & 0xFF) : 0u;
			v |= byte << (8 * k);
		}
		r.l[i] = v;
	}
	return r;
}
WV_FN uint64_t ballot(const Pred& p)
{
	uint64_t m = 0;
	for (int i = 0; i < WAVE; ++i) m |= (uint64_t)(p.l[i] ? 1 : 0) << i;
	return m;
}
WV_FN uint32_t readlane(const U32& a, uint32_t lane) { return a.l[lane & 63]; }
// {hi, lo} = (64-bit) x << sh, sh in 0..32
WV_FN void shl64(const U32& x, const U32& sh, U32& lo, U32& hi)
{
	for (int i = 0; i < WAVE; ++i) {
		const uint64_t v = (uint64_t)x.l[i] << sh.l[i];
		lo.l[i] = (uint32_t)v;
		hi.l[i] = (uint32_t)(v >> 32);
	}
}
// bit r of lane 16k + q: p in lane 16k + r (the ballot of the lane's own group of 16 lanes)
WV_FN U32 row_ballot16(const Pred& p)
{
	const uint64_t m = ballot(p);
	U32 r;
	for (int i = 0; i < WAVE; ++i) r.l[i] = (uint32_t)(m >> (i & 48)) & 0xFFFFu;
	return r;
}
// 1 when the lane mask is not empty, else 0
WV_FN uint32_t mask_nonzero(uint64_t m) { return m ? 1u : 0u; }
template <uint32_t BIT>
WV_FN uint32_t mask_bit(uint64_t m) { return m ? BIT : 0u; }
// lane i reads a[src[i] & 63]
WV_FN U32 shfl(const U32& a, const U32& src)
{
	U32 r;
	for (int i = 0; i < WAVE; ++i) r.l[i] = a.l[src.l[i] & 63];
	return r;
}
// lane i reads a[i - n]; lanes < n read `fill`
WV_FN U32 shfl_up(const U32& a, uint32_t n, uint32_t fill)
{
	U32 r;
	for (int i = 0; i < WAVE; ++i) r.l[i] = (uint32_t)i >= n ? a.l[i - (int)n] : fill;
	return r;
}
// lane i reads a[i ^ m]
WV_FN U32 shfl_xor(const U32& a, uint32_t m)
{
	U32 r;
	for (int i = 0; i < WAVE; ++i) r.l[i] = a.l[i ^ (int)m];
	return r;
}
// rotate right by n inside each aligned group of 16 lanes: lane i reads a[(i & ~15) | ((i + n) & 15)]
WV_FN U32 row_ror(const U32& a, uint32_t n)
{
	U32 r;
	for (int i = 0; i < WAVE; ++i) r.l[i] = a.l[(i & ~15) | ((i - (int)n) & 15)];
	return r;
}
// shift right by n inside each aligned group of 16 lanes; the first n lanes of each group read `fill`
WV_FN U32 row_shr(const U32& a, uint32_t n, uint32_t fill)
{
	U32 r;
	for (int i = 0; i < WAVE; ++i) r.l[i] = (uint32_t)(i & 15) >= n ? a.l[i - (int)n] : fill;
	return r;
}
// every lane reads the last lane of its group of 4
WV_FN U32 quad_last(const U32& a)
{
	U32 r;
	for (int i = 0; i < WAVE; ++i) r.l[i] = a.l[i | 3];
	return r;
}
WV_FN void wave_sync() {}
// lanes 16k .. 16k+15 receive a_k (four uniform values)
WV_FN U32 row_select4(uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3)
{
	U32 r;
	for (int i = 0; i < WAVE; ++i) r.l[i] = i < 16 ? a0 : (i < 32 ? a1 : (i < 48 ? a2 : a3));
	return r;
}
// lanes 16k .. 16k+15 keep their own value of a_k (four vectors)
WV_FN U32 row_select4v(const U32& a0, const U32& a1, const U32& a2, const U32& a3)
{
	U32 r;
	for (int i = 0; i < WAVE; ++i) r.l[i] = i < 16 ? a0.l[i] : (i < 32 ? a1.l[i] : (i < 48 ? a2.l[i] : a3.l[i]));
	return r;
}

// ---- LDS (byte addressed; 32-bit accesses must be 4-byte aligned unless named *_unaligned) ----
WV_FN U32 lds_ld8(Lds m, const U32& a)
{
	U32 r;
	for (int i = 0; i < WAVE; ++i) r.l[i] = m[a.l[i]];
	return r;
}
WV_FN void lds_rle_walk_row(Lds m, uint32_t base, uint32_t first, U32& e, U32& mask) // (wavevec.h: one run-length row among rows of other kinds)
{
	const uint32_t at = base + e.l[first];
	const uint32_t mk = (uint32_t)m[at] | ((uint32_t)m[at + 1] << 8);
	for (int i = (int)first; i < WAVE; ++i)
		mask.l[i] = mk;
	for (int i = (int)first + 4; i < WAVE; ++i)
		e.l[i] += 18 - (uint32_t)__builtin_popcount(mk);
}
WV_FN void lds_lz_walk32(Lds m, U32& at, U32& flags, uint32_t B) // (wavevec.h: the 32 groups of a mini-LZ block, every match taken for one byte)
{
	uint32_t a = at.l[0];
	for (int g = 0; g < 32; ++g) {
		const uint32_t fl = m[a];
		for (int i = g; i < (g == 31 ? 32 : g + 1); ++i) {
			at.l[i] = a;
			flags.l[i] = fl;
		}
		a += 1 + 8 * B - (B - 1) * (uint32_t)__builtin_popcount(fl);
	}
	for (int i = 32; i < WAVE; ++i) { // (one step on; not used)
		at.l[i] = a;
		flags.l[i] = m[a];
	}
}
WV_FN void lds_rle_walk16(Lds m, U32& at, U32& mask) // (wavevec.h: sixteen run-length rows in a chain, the lane's row's offset and mask)
{
	uint32_t a = at.l[0];
	for (int r = 0; r < 16; ++r) {
		const uint32_t mk = (uint32_t)m[a] | ((uint32_t)m[a + 1] << 8);
		for (int i = 4 * r; i < 4 * r + 4; ++i) {
			at.l[i] = a;
			mask.l[i] = mk;
		}
		a += 18 - (uint32_t)__builtin_popcount(mk);
	}
}
WV_FN U32 lds_ld32(Lds m, const U32& a)
{
	U32 r;
	for (int i = 0; i < WAVE; ++i) memcpy(&r.l[i], m + (a.l[i] & ~3u), 4);
	return r;
}
WV_FN void lds_ld64(Lds m, const U32& a, U32& lo, U32& hi)
{
	for (int i = 0; i < WAVE; ++i) {
		memcpy(&lo.l[i], m + (a.l[i] & ~3u), 4);
		memcpy(&hi.l[i], m + (a.l[i] & ~3u) + 4, 4);
	}
}
WV_FN void lds_st32(Lds m, const U32& a, const U32& v, const Pred& p)
{
	for (int i = 0; i < WAVE; ++i)
		if (p.l[i]) memcpy(m + (a.l[i] & ~3u), &v.l[i], 4);
}
WV_FN void lds_st8(Lds m, const U32& a, const U32& v, const Pred& p)
{
	for (int i = 0; i < WAVE; ++i)
		if (p.l[i]) m[a.l[i]] = (uint8_t)v.l[i];
}
// atomic add returning the previous value (lane order on the host; any order is a valid device order)
WV_FN U32 lds_add_rtn32(Lds m, const U32& a, const U32& v, const Pred& p)
{
	U32 r(0u);
	for (int i = 0; i < WAVE; ++i)
		if (p.l[i]) {
			uint32_t t;
			memcpy(&t, m + (a.l[i] & ~3u), 4);
			r.l[i] = t;
			t += v.l[i];
			memcpy(m + (a.l[i] & ~3u), &t, 4);
		}
	return r;
}
// compare-and-swap returning the previous value
WV_FN U32 lds_cas32(Lds m, const U32& a, const U32& expect, const U32& v, const Pred& p)
{
	U32 r(0u);
	for (int i = 0; i < WAVE; ++i)
		if (p.l[i]) {
			uint32_t t;
			memcpy(&t, m + (a.l[i] & ~3u), 4);
			r.l[i] = t;
			if (t == expect.l[i]) memcpy(m + (a.l[i] & ~3u), &v.l[i], 4);
		}
	return r;
}
// atomic OR returning the previous value (lane order on the host; any order is a valid device order)
WV_FN U32 lds_or_rtn32(Lds m, const U32& a, const U32& v)
{
	U32 r;
	for (int i = 0; i < WAVE; ++i) {
		uint32_t t;
		memcpy(&t, m + a.l[i], 4);
		r.l[i] = t;
		t |= v.l[i];
		memcpy(m + a.l[i], &t, 4);
	}
	return r;
}
// every lane ORs v into the dword at a (a multiple of 4)
WV_FN void lds_or32_all(Lds m, const U32& a, const U32& v)
{
	for (int i = 0; i < WAVE; ++i) {
		uint32_t t;
		memcpy(&t, m + a.l[i], 4);
		t |= v.l[i];
		memcpy(m + a.l[i], &t, 4);
	}
}
WV_FN void lds_or32(Lds m, const U32& a, const U32& v, const Pred& p)
{
	for (int i = 0; i < WAVE; ++i)
		if (p.l[i]) {
			uint32_t t;
			memcpy(&t, m + (a.l[i] & ~3u), 4);
			t |= v.l[i];
			memcpy(m + (a.l[i] & ~3u), &t, 4);
		}
}

// ---- global memory (plain pointers on the host) ----
struct U128 {
	U32 x, y, z, w;
};
WV_FN U32 gld8(const uint8_t* g, const U32& off, const Pred& p)
{
	U32 r(0u);
	for (int i = 0; i < WAVE; ++i)
		if (p.l[i]) r.l[i] = g[off.l[i]];
	return r;
}
WV_FN U32 gld32(const uint8_t* g, const U32& off, const Pred& p) // off multiple of 4, g 4-byte aligned
{
	U32 r(0u);
	for (int i = 0; i < WAVE; ++i)
		if (p.l[i]) memcpy(&r.l[i], g + off.l[i], 4);
	return r;
}
WV_FN void gld64(const uint8_t* g, const U32& off, U32& lo, U32& hi) // 8-byte aligned, all lanes
{
	for (int i = 0; i < WAVE; ++i) {
		memcpy(&lo.l[i], g + off.l[i], 4);
		memcpy(&hi.l[i], g + off.l[i] + 4, 4);
	}
}
WV_FN U128 gld128(const uint8_t* g, const U32& off, const Pred& p) // 16-byte aligned
{
	U128 r;
	r.x = r.y = r.z = r.w = U32(0u);
	for (int i = 0; i < WAVE; ++i)
		if (p.l[i]) {
			memcpy(&r.x.l[i], g + off.l[i], 4);
			memcpy(&r.y.l[i], g + off.l[i] + 4, 4);
			memcpy(&r.z.l[i], g + off.l[i] + 8, 4);
			memcpy(&r.w.l[i], g + off.l[i] + 12, 4);
		}
	return r;
}
// the same accesses at any byte address
WV_FN U128 gld128_unaligned(const uint8_t* g, const U32& off, const Pred& p) { return gld128(g, off, p); }
WV_FN void gld64_unaligned(const uint8_t* g, const U32& off, U32& lo, U32& hi) { gld64(g, off, lo, hi); }
WV_FN void gst8(uint8_t* g, const U32& off, const U32& v, const Pred& p)
{
	for (int i = 0; i < WAVE; ++i)
		if (p.l[i]) g[off.l[i]] = (uint8_t)v.l[i];
}
WV_FN void gst32(uint8_t* g, const U32& off, const U32& v, const Pred& p)
{
	for (int i = 0; i < WAVE; ++i)
		if (p.l[i]) memcpy(g + off.l[i], &v.l[i], 4);
}
WV_FN void gst64(uint8_t* g, const U32& off, const U32& lo, const U32& hi, const Pred& p)
{
	for (int i = 0; i < WAVE; ++i)
		if (p.l[i]) {
			memcpy(g + off.l[i], &lo.l[i], 4);
			memcpy(g + off.l[i] + 4, &hi.l[i], 4);
		}
}
WV_FN void gst128(uint8_t* g, const U32& off, const U128& v, const Pred& p)
{
	for (int i = 0; i < WAVE; ++i)
		if (p.l[i]) {
			memcpy(g + off.l[i], &v.x.l[i], 4);
			memcpy(g + off.l[i] + 4, &v.y.l[i], 4);
			memcpy(g + off.l[i] + 8, &v.z.l[i], 4);
			memcpy(g + off.l[i] + 12, &v.w.l[i], 4);
		}
}
WV_FN void gst128_unaligned(uint8_t* g, const U32& off, const U128& v, const Pred& p) { gst128(g, off, v, p); }
WV_FN void gst128_streamed(uint8_t* g, const U32& off, const U128& v, const Pred& p) { gst128(g, off, v, p); }
WV_FN void gst8_streamed(uint8_t* g, const U32& off, const U32& v, const Pred& p) { gst8(g, off, v, p); }
WV_FN void gst64_unaligned(uint8_t* g, const U32& off, const U32& lo, const U32& hi, const Pred& p) { gst64(g, off, lo, hi, p); }
WV_FN void gst128_through(uint8_t* g, const U32& off, const U128& v) { gst128(g, off, v, pred_all(true)); }
WV_FN void gst64_through(uint8_t* g, const U32& off, const U32& lo, const U32& hi) { gst64(g, off, lo, hi, pred_all(true)); }
WV_FN void gst_through_wait() {}
// wave-uniform scalar accesses to global memory
WV_FN uint32_t gload_uniform(const uint32_t* p) { return *p; }
WV_FN uint32_t gload_uniform8(const uint8_t* p) { return *p; }
WV_FN uint64_t gload_uniform64(const uint64_t* p) { return *p; }
WV_FN void gstore_uniform(uint32_t* p, uint32_t v) { *p = v; }
WV_FN void gstore_uniform8(uint8_t* p, uint32_t v) { *p = (uint8_t)v; }
WV_FN void gstore_uniform64(uint64_t* p, uint64_t v) { *p = v; }
WV_FN void gmin32(uint32_t* p, uint32_t v)
{
	if (v < *p) *p = v;
}
WV_FN U128 lds_ld128(Lds m, const U32& a) // 16-byte aligned
{
	U128 r;
	for (int i = 0; i < WAVE; ++i) {
		memcpy(&r.x.l[i], m + a.l[i], 4);
		memcpy(&r.y.l[i], m + a.l[i] + 4, 4);
		memcpy(&r.z.l[i], m + a.l[i] + 8, 4);
		memcpy(&r.w.l[i], m + a.l[i] + 12, 4);
	}
	return r;
}
WV_FN void lds_st128(Lds m, const U32& a, const U128& v, const Pred& p)
{
	for (int i = 0; i < WAVE; ++i)
		if (p.l[i]) {
			memcpy(m + a.l[i], &v.x.l[i], 4);
			memcpy(m + a.l[i] + 4, &v.y.l[i], 4);
			memcpy(m + a.l[i] + 8, &v.z.l[i], 4);
			memcpy(m + a.l[i] + 12, &v.w.l[i], 4);
		}
}

// ---- wave-wide scans and reductions (the device forms, DPP, stand in wavevec.h) ----
// maximum over the 64 lanes (uniform result)
WV_FN uint32_t wave_max(U32 x)
{
	x = umax(x, shfl_xor(x, 1));
	x = umax(x, shfl_xor(x, 2));
	x = umax(x, shfl_xor(x, 4));
	x = umax(x, shfl_xor(x, 8));
	x = umax(x, shfl_xor(x, 16));
	x = umax(x, shfl_xor(x, 32));
	return readlane(x, 0);
}
// OR over the 64 lanes (uniform result)
WV_FN uint32_t wave_or(const U32& x)
{
	uint32_t r = 0;
	for (int i = 0; i < WAVE; ++i) r |= x.l[i];
	return r;
}
// The six data movements of a wave-wide inclusive scan in DPP order (any associative operator): steps 0-3 shift by 1, 2,
// 4, 8 inside the rows of 16 lanes, step 4 hands the total of rows 0 / 2 (lanes 15 / 47) to rows 1 / 3, step 5 the total of
// rows 0-1 (lane 31) to rows 2 and 3.  Lanes without a source read `ident`.
WV_FN U32 scan_source(const U32& x, int step, uint32_t ident)
{
	U32 r(ident);
	for (int i = 0; i < WAVE; ++i) {
		if (step < 4) {
			int d = 1 << step;
			if ((i & 15) >= d) r.l[i] = x.l[i - d];
		}
		else if (step == 4) {
			if ((i >> 4) & 1) r.l[i] = x.l[(i & ~15) - 1];
		}
		else if (i >= 32)
			r.l[i] = x.l[31];
	}
	return r;
}
// inclusive prefix maximum over the 64 lanes
WV_FN U32 wave_incl_scan_max(U32 s)
{
	for (int i = 1; i < WAVE; ++i) s.l[i] = s.l[i] > s.l[i - 1] ? s.l[i] : s.l[i - 1];
	return s;
}
// inclusive prefix sum over the 64 lanes
WV_FN U32 wave_incl_scan(U32 s)
{
	s = s + shfl_up(s, 1, 0);
	s = s + shfl_up(s, 2, 0);
	s = s + shfl_up(s, 4, 0);
	s = s + shfl_up(s, 8, 0);
	s = s + shfl_up(s, 16, 0);
	return s + shfl_up(s, 32, 0);
}
} // namespace wv
