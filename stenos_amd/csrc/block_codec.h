// block_codec.h -- the Stenos 256-element block codec, one wavefront per block, written in the
// wavevec.h vocabulary (gfx950 device code; the same source runs as a lockstep host emulation in tests).
//
// Bit stream (reference: stenos/internal/block_compress.h, lz_compress.h):
//   full block  = [ceil(T/2) plane-type nibbles][plane 0]...[plane T-1]   or   [253][mini-LZ stream]
//   plane       = SAME [v] | RAW [256 bytes] | NORMAL [8 B row headers][mins][rows] | NORMAL_RLE [8 B][mask16][mins][rows]
//   row payload = 16 raw | rle [mask16][literals] | delta-rle | two halves of 8 values packed LSB first
//
// Lane mapping.  "Element lanes": lane l owns elements 4l..4l+3 of the block, so one row of 16
// elements is one quad of lanes and a plane's 4 bytes per lane travel as one packed dword.
// "Row lanes": lane 16*p + r owns row r of plane p of the current group of four planes; per-row
// decisions (bit widths, rle, headers, offsets) are taken once there for four planes at a time.
// The two views exchange data through the wave's private LDS scratch.
//
// Everything written to the output goes through lds_put_bits() (an OR into a zeroed LDS image), so
// pieces of any bit length land at any byte offset without read-modify-write ordering problems.
#pragma once
#include "wavevec.h"

namespace codec {
using namespace wv;

enum { PLANE_SAME = 0, PLANE_RAW = 1, PLANE_NORMAL = 2, PLANE_NORMAL_RLE = 3 };
enum { BLOCK_COPY = 252, BLOCK_LZ = 253, BLOCK_PARTIAL = 254 };
constexpr uint32_t LZ_NONE = 0xFFFFFFFFu;
// Largest bytesoftype of the LDS-resident codec.  Above it (kernels_wide.hip, -DSTENOS_WIDE) the same source runs with the
// wave's scratch in HBM: a block of 256*T bytes, its image and its tables no longer fit the LDS of a workgroup.
#ifdef STENOS_WIDE
constexpr uint32_t MAX_T = 65534; // stenos.h:65
#else
constexpr uint32_t MAX_T = 64;
#endif
constexpr uint32_t LZ_MAX_T = 512; // lz_compress_generic gives up above (15-bit distances, lz_compress.h:281-283)
// plinfo entry: type | size_or_offset << PL_SHIFT (offsets reach 280 * 65534)
constexpr uint32_t PL_SHIFT = 4, PL_TYPE = 15;

// Byte offsets of the regions of one wave's LDS scratch.
struct Layout {
	uint32_t in;      // raw block, element major: 256*T bytes (+8 slack)
	uint32_t out;     // encoded image: out_capacity(T) bytes, zeroed by the codec; the 16 bytes in front of it belong to RunStream
	uint32_t rowinfo; // T*16 entries of 8 bytes
	uint32_t plinfo;  // T entries of 4 bytes
	uint32_t rlelut;  // row lanes (bytesoftype 2, 4, 8): the 16 v_perm_b32 selectors that compact the literals of a run-length group, written once per run
	uint32_t grp;     // bytesoftype 2, 4: three tables of eight words for a group of blocks (slot_codec.h, "groups of any shape")
	uint32_t aux;     // 64 entries of 8 bytes: row statistics of the current plane group (dead once the planes are analysed)
	uint32_t lz;      // mini-LZ chain count*4 + cur count*4, on top of aux; its 256-entry table uses the not yet written image
	uint32_t skip;    // mini-LZ: one bit per group that was left raw (32 bytes up to bytesoftype 64)
	uint32_t tab;     // 1 KiB for the tables of the mini-LZ (hash table, counters and bitmaps of its rejection tests): the image, which is
	                  // only written once they are dead
	uint32_t total;
};

WV_HD uint32_t align16(uint32_t x) { return (x + 15u) & ~15u; }
WV_HD uint32_t lz_width(uint32_t T) { return (T % 8 == 0) ? 8u : 4u; } // lz_compress.h:285-290 for T%4==0
WV_HD uint32_t header_bytes(uint32_t T) { return (T + 1) >> 1; }
WV_HD uint32_t lz_skip_bytes(uint32_t T) // one bit per group of 8 values
{
	const uint32_t bytes = align16(256 * T / lz_width(T) / 64);
	return (T % 4 == 0 && T <= LZ_MAX_T && bytes > 32) ? bytes : 32;
}
// a partial block can take 1 + T/2 + T*(8 + 15*17) + (16*T - 1) bytes, more than a full one
// bytesoftype 2: two full blocks of a batch are written into the image together (slot_codec.h)
WV_HD uint32_t out_capacity(uint32_t T) { return T == 2 ? 1088u : align16(280 * T + header_bytes(T) + 40); }

WV_HD Layout make_layout(uint32_t T, bool with_lz)
{
	Layout L;
	uint32_t o = 0;
	o += 16; // bytes of the previous block that wait for a full 16-byte group (superblock_codec.h, RunStream)
	L.out = o;
	o += out_capacity(T);
	// The raw block stands behind the image: the row-lane encoder of bytesoftype 2 and 4 never fills it while it writes an image
	// (only the mini-LZ attempt and the plane-group encoder read a block from LDS), so the image of a group of four blocks
	// (superblock_codec.h, quad_image_bytes: up to 2 112 bytes) may run on into it -- and, for bytesoftype 2, into the row table.
	L.in = o;
	o += align16(256 * T + 16);
	L.rowinfo = o;
	o += (T < 4 ? 4 : T) * 16 * 8;

	L.plinfo = o;
	o += align16(T * 4);
	L.rlelut = o; // (64 bytes; slot_codec.h)
	o += (T == 2 || T == 4 || T == 8) ? 64 : 0;
	L.grp = o; // (96 bytes: first elements, plane sizes and types, plane positions)
	o += (T == 2 || T == 4) ? 96 : 0;
	L.skip = o;
	o += lz_skip_bytes(T);
	L.aux = o;
	L.lz = o;
	uint32_t scratch = 64 * 8;
	if (T == 2 || T == 4)
		scratch = 8 * 256; // the plane slots of slot_codec.h: four per pass, two passes for a group of four blocks
	if (with_lz && T % 4 == 0 && T <= LZ_MAX_T) {
		const uint32_t count = 256 * T / lz_width(T);
		scratch = count * 8 > scratch ? count * 8 : scratch;
	}
	o += scratch;
	L.total = align16(o);
	L.tab = L.out;
	return L;
}

// ------------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------------

// For bytesoftype 4, 8, 12, ... the plane words of four consecutive planes [g, g+4) of a lane's four elements come
// out of one dword per element and three v_perm_b32 per plane (bytesoftype 2: both planes out of two dwords);
// other sizes gather the bytes from the LDS copy.
struct PlaneRegs {
	U32 w[4];
	bool valid;
	uint32_t g; // first plane held
};
// bytes 0..3 of four consecutive elements (e0..e3, one dword each) -> the four plane words of the lane
WV_FN void plane_words_from_elements(PlaneRegs& r, const U32& e0, const U32& e1, const U32& e2, const U32& e3)
{
	for (uint32_t j = 0; j < 4; ++j) {
		const uint32_t s2 = 0x0c0c0000u | ((4u + j) << 8) | j; // [lo.bj, hi.bj, 0, 0]
		U32 p01 = perm_bytes(e1, e0, s2), p23 = perm_bytes(e3, e2, s2);
		r.w[j] = perm_bytes(p23, p01, 0x05040100u);
	}
}
// the lane's four 16-bit elements (two dwords) -> two plane words
WV_FN void plane_words_from_int16(PlaneRegs& r, const U32& lo, const U32& hi)
{
	r.w[0] = perm_bytes(hi, lo, 0x06040200u);
	r.w[1] = perm_bytes(hi, lo, 0x07050301u);
	r.w[2] = r.w[3] = U32(0u);
}
// A block of bytesoftype 2 or 4 straight from HBM into registers, from any byte address (gfx9 and later serve unaligned
// global accesses in hardware: a slice of an array compresses as fast as the array).
struct RawBlock { // the lane's share of a block of bytesoftype 2 (x, y) or 4 (x, y, z, w): its four elements
	U128 e;
};
WV_FN RawBlock load_raw_block(const uint8_t* g, uint32_t T)
{
	RawBlock r;
	if (T == 2) {
		gld64_unaligned(g, lane_id() * 8u, r.e.x, r.e.y);
		r.e.z = r.e.w = U32(0u);
	}
	else
		r.e = gld128_unaligned(g, lane_id() * 16u, pred_all(true));
	return r;
}
WV_FN PlaneRegs plane_regs_of(const RawBlock& b, uint32_t T)
{
	PlaneRegs r;
	r.valid = true;
	r.g = 0;
	if (T == 2)
		plane_words_from_int16(r, b.e.x, b.e.y);
	else
		plane_words_from_elements(r, b.e.x, b.e.y, b.e.z, b.e.w);
	return r;
}
// the block -> L.in, element major as load_block leaves it
WV_FN void store_raw_block(Lds lds, uint32_t in, const RawBlock& b, uint32_t T)
{
	if (T == 2) {
		lds_st32(lds, U32(in) + lane_id() * 8u, b.e.x, pred_all(true));
		lds_st32(lds, U32(in + 4u) + lane_id() * 8u, b.e.y, pred_all(true));
	}
	else
		lds_st128(lds, U32(in) + lane_id() * 16u, b.e, pred_all(true));
}
WV_FN PlaneRegs load_plane_regs(Lds lds, uint32_t in, uint32_t T, uint32_t g)
{
	PlaneRegs r;
	r.valid = T % 4 == 0 || T == 2;
	r.g = g;
	if (T == 2) { // the lane's four 16-bit elements: two dwords, two planes
		U32 lo, hi;
		lds_ld64(lds, U32(in) + lane_id() * 8u, lo, hi);
		plane_words_from_int16(r, lo, hi);
	}
	else if (r.valid) {
		U32 e0, e1, e2, e3; // bytes g..g+3 of the lane's four elements
		if (T == 4) {
			U128 e = lds_ld128(lds, U32(in) + lane_id() * 16u);
			e0 = e.x;
			e1 = e.y;
			e2 = e.z;
			e3 = e.w;
		}
		else if (T == 8) {
			// the lane's four doubles are 32 consecutive bytes: two 16-byte reads (lanes 8 apart share banks: 2-way) instead
			// of four dword reads at a stride of 32 bytes (lanes 4 apart share a bank: 8-way conflicts on every one of them)
			const U32 a = U32(in) + lane_id() * 32u;
			const U128 lo = lds_ld128(lds, a), hi = lds_ld128(lds, a + 16u);
			e0 = g ? lo.y : lo.x;
			e1 = g ? lo.w : lo.z;
			e2 = g ? hi.y : hi.x;
			e3 = g ? hi.w : hi.z;
		}
		else {
			U32 a = U32(in + g) + lane_id() * (4u * T);
			e0 = lds_ld32(lds, a);
			e1 = lds_ld32(lds, a + T);
			e2 = lds_ld32(lds, a + 2u * T);
			e3 = lds_ld32(lds, a + 3u * T);
		}
		plane_words_from_elements(r, e0, e1, e2, e3);
	}
	return r;
}
// the four bytes of plane j owned by this element lane: elements 4l..4l+3
WV_FN U32 fetch_plane_word(Lds lds, uint32_t in, uint32_t T, uint32_t j, const PlaneRegs& regs)
{
	if (regs.valid && j - regs.g < 4u && j < T) {
		const uint32_t k = j - regs.g;
		return k == 0 ? regs.w[0] : (k == 1 ? regs.w[1] : (k == 2 ? regs.w[2] : regs.w[3]));
	}
	U32 a = U32(in + j) + lane_id() * (4u * T);
	U32 b0 = lds_ld8(lds, a);
	U32 b1 = lds_ld8(lds, a + T);
	U32 b2 = lds_ld8(lds, a + 2u * T);
	U32 b3 = lds_ld8(lds, a + 3u * T);
	return b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
}

WV_FN U32 min4(const U32& x) { return umin(umin(byte_of(x, 0), byte_of(x, 1)), umin(byte_of(x, 2), byte_of(x, 3))); }
WV_FN U32 max4(const U32& x) { return umax(umax(byte_of(x, 0), byte_of(x, 1)), umax(byte_of(x, 2), byte_of(x, 3))); }

// W(): bits needed for a byte range with 7 promoted to 8 (block_compress.h:334-352)
WV_FN U32 width_of(const U32& range)
{
	U32 b = bitlen(range);
	return sel(b >= U32(7u), U32(8u), b);
}

// per-lane pieces every plane pass needs: packed bytes w, packed deltas dw (previous byte in plane
// order, 0 before the plane's first byte), zero masks of the deltas (rle) and of the delta differences
// (delta-rle, previous delta of a row's first column is 0)
struct PlaneWords {
	U32 w, dw, z1, z2;
};
WV_FN PlaneWords plane_words(const U32& w)
{
	PlaneWords p;
	p.w = w;
	U32 prevb = shfl_up(w, 1, 0) >> 24; // lane 0: 0  (block_compress.h:399-401)
	p.dw = bytes_sub(w, (w << 8) | prevb);
	p.z1 = bytes_zero_mask(p.dw); // byte == previous byte  (:268-275)
	U32 pd = sel((lane_id() & 3u) == U32(0u), U32(0u), shfl_up(p.dw, 1, 0) >> 24);
	p.z2 = bytes_zero_mask(p.dw ^ ((p.dw << 8) | pd)); // delta == previous delta  (:248-255, 449-458)
	return p;
}

// keep the bytes of x whose flag bit (bits 0..3 of f) is 0, packed towards the low end
WV_FN U32 compact_unflagged(const U32& x, const U32& f)
{
	U32 out(0u), n(0u);
	for (int k = 0; k < 4; ++k) {
		Pred keep = ((f >> U32((uint32_t)k)) & 1u) == U32(0u);
		out = out | sel(keep, byte_of(x, k) << (n << 3), U32(0u));
		n = n + sel(keep, U32(1u), U32(0u));
	}
	return out;
}

// ------------------------------------------------------------------------------------------------
// encoder
// ------------------------------------------------------------------------------------------------

// Stage 1 of the analysis for bytesoftype 4, all four planes at once.  The quad of lanes that holds a row of 16
// elements transposes its plane words (4 x 4 dwords, two DPP exchanges) so that lane 4r+q owns the 16 bytes of row r
// of plane q; minima, maxima, delta ranges and the two run counts then need no cross-lane step at all, and constant
// planes cost nothing extra (their rows come out as mx == mn with equal minima, which is what makes a plane SAME).
// Same statistics, same L.aux entries (plane*16 + row) as the plane-at-a-time form below.
WV_FN U32 bytes_min16(const U32& e0, const U32& o0, const U32& e1, const U32& o1, const U32& e2, const U32& o2, const U32& e3, const U32& o3)
{
	U32 m = pk_min_u16(pk_min_u16(pk_min_u16(e0, o0), pk_min_u16(e1, o1)), pk_min_u16(pk_min_u16(e2, o2), pk_min_u16(e3, o3)));
	return umin(m & 0xFFFFu, m >> 16);
}
WV_FN U32 bytes_max16(const U32& e0, const U32& o0, const U32& e1, const U32& o1, const U32& e2, const U32& o2, const U32& e3, const U32& o3)
{
	U32 m = pk_max_u16(pk_max_u16(pk_max_u16(e0, o0), pk_max_u16(e1, o1)), pk_max_u16(pk_max_u16(e2, o2), pk_max_u16(e3, o3)));
	return umax(m & 0xFFFFu, m >> 16);
}
// Lane 4r+q holds in b[0..3] the 16 bytes of row r of plane (or plane slot) q: statistics of that row -> L.aux.
WV_FN void row_stats_store(Lds lds, const Layout& L, const U32* b)
{
	const U32 lane = lane_id();
	const U32 q = lane & 3u;
	// deltas against the previous byte in plane order; before a row comes the last byte of the row above (lane - 4),
	// before the plane 0 (block_compress.h:399-401)
	U32 prev = shfl_up(b[3], 4, 0) >> 24;
	U32 d[4];
	d[0] = bytes_sub(b[0], (b[0] << 8) | prev);
	for (int k = 1; k < 4; ++k)
		d[k] = bytes_sub(b[k], (b[k] << 8) | (b[k - 1] >> 24));
	// runs: byte == previous byte (:268-275); delta == previous delta, 0 before the row's first column (:248-255, 449-458)
	U32 nrle(0u), ndrle(0u);
	for (int k = 0; k < 4; ++k) {
		nrle = nrle + popc(bytes_zero_mask(d[k]));
		U32 pd = k ? (d[k - 1] >> 24) : U32(0u);
		ndrle = ndrle + popc(bytes_zero_mask(d[k] ^ ((d[k] << 8) | pd)));
	}
	// ranges in signed order (:407-411): even and odd bytes side by side as 16-bit halves
	U32 e[4], o[4], de[4], dod[4];
	for (int k = 0; k < 4; ++k) {
		U32 s = b[k] ^ 0x80808080u, ds = d[k] ^ 0x80808080u;
		e[k] = s & 0x00FF00FFu;
		o[k] = (s >> 8) & 0x00FF00FFu;
		de[k] = ds & 0x00FF00FFu;
		dod[k] = (ds >> 8) & 0x00FF00FFu;
	}
	U32 mn = bytes_min16(e[0], o[0], e[1], o[1], e[2], o[2], e[3], o[3]);
	U32 mx = bytes_max16(e[0], o[0], e[1], o[1], e[2], o[2], e[3], o[3]);
	U32 dmn = bytes_min16(de[0], dod[0], de[1], dod[1], de[2], dod[2], de[3], dod[3]);
	U32 dmx = bytes_max16(de[0], dod[0], de[1], dod[1], de[2], dod[2], de[3], dod[3]);
	U32 addr = U32(L.aux) + (q * 16u + (lane >> 2)) * 8u;
	lds_st32(lds, addr, mn | (mx << 8) | (dmn << 16) | (dmx << 24), pred_all(true));
	lds_st32(lds, addr + 4u, nrle | (ndrle << 16), pred_all(true));
}
WV_FN void analyse_rows_int32(Lds lds, const Layout& L, const PlaneRegs& regs)
{
	const U32 q = lane_id() & 3u;
	// 4 x 4 transpose inside the quad: b[k] = plane-q word of the quad's lane k (elements 4k..4k+3 of the row)
	U32 b[4];
	{
		Pred odd = (q & 1u) == U32(1u), hi = (q & 2u) == U32(2u);
		U32 r0 = shfl_xor(sel(odd, regs.w[0], regs.w[1]), 1), r1 = shfl_xor(sel(odd, regs.w[2], regs.w[3]), 1);
		U32 c0 = sel(odd, r0, regs.w[0]), c1 = sel(odd, regs.w[1], r0);
		U32 c2 = sel(odd, r1, regs.w[2]), c3 = sel(odd, regs.w[3], r1);
		U32 s0 = shfl_xor(sel(hi, c0, c2), 2), s1 = shfl_xor(sel(hi, c1, c3), 2);
		b[0] = sel(hi, s0, c0);
		b[2] = sel(hi, c2, s0);
		b[1] = sel(hi, s1, c1);
		b[3] = sel(hi, c3, s1);
	}
	row_stats_store(lds, L, b);
}

// rowinfo entry: lo = hdr | min<<8 | poff<<16 ; hi = minpos | emitmin<<12 | eq<<13
// plinfo entry : type | size_or_offset<<PL_SHIFT

// Stage 2 of the analysis, on row lanes: lane 16p + r decides row r of plane g + p from the statistics in L.aux
// (entry p*16 + r) and leaves the row's header, minimum and offsets in L.rowinfo (entry (g + p)*16 + r).
WV_FN void analyse_stage2(Lds lds, const Layout& L, uint32_t g, uint32_t np, bool rle, uint32_t lines, uint32_t* praw)
{
	const U32 lane = lane_id();
	WV_MARK("analyse_stage2");

	const U32 r = lane & 15u;
	const U32 pl = lane >> 4;
	const Pred valid = pl < U32(np);
	const Pred act = r < U32(lines);
	U32 lo, hi;
	lds_ld64(lds, U32(L.aux) + lane * 8u, lo, hi);
	U32 mn = lo & 0xFFu, mx = (lo >> 8) & 0xFFu, dmn = (lo >> 16) & 0xFFu, dmx = lo >> 24;
	U32 nrle = hi & 0xFFFFu, ndrle = hi >> 16;
	U32 b0 = width_of(mx - mn), b1 = width_of(dmx - dmn);
	b0 = sel(b0 == U32(6u), U32(8u), b0); // header 6 is reserved for delta-rle (:422)
	U32 bits = umin(b0, b1);
	Pred type0 = b0 == bits; // ties go to frame-of-reference (:423-427)
	U32 minv = sel(type0, mn, dmn) ^ 0x80u;
	U32 cost = bits * 2u + sel(bits != U32(8u), U32(1u), U32(0u)); // (:433-435)
	U32 hdr = sel(type0, sel(b0 == U32(8u), U32(15u), b0), b1 + 8u); // (:497-503)
	if (rle) {
		U32 c1 = U32(18u) - nrle; // 2 + 16 - popcnt, strictly smaller wins (:464-467)
		Pred u1 = c1 < cost;
		cost = sel(u1, c1, cost);
		hdr = sel(u1, U32(7u), hdr);
		U32 c2 = U32(18u) - ndrle; // (:470-472)
		Pred u2 = c2 < cost;
		cost = sel(u2, c2, cost);
		hdr = sel(u2, U32(6u), hdr);
	}
	Pred nomin = (hdr == U32(6u)) | (hdr == U32(7u)) | (hdr == U32(15u));
	Pred eq = minv == row_shr(minv, 1, 0); // min equals previous row's min, 0 before row 0 (:483)
	Pred rowsame = (mx == mn) & ((r == U32(0u)) | eq);

	U32 packed = sel(act, cost, U32(0u)) | sel(act & nomin, U32(1u << 12), U32(0u)) | sel(eq, U32(1u << 17), U32(0u)) |
		     sel(rowsame, U32(1u << 22), U32(0u));
	U32 tot = row_add(packed);
	U32 sumcost = tot & 0xFFFu, count8 = (tot >> 12) & 31u, eqc = (tot >> 17) & 31u, samec = (tot >> 22) & 31u;
	const uint32_t nh = (lines + 1) >> 1; // row-header bytes: 8 for full blocks
	U32 size = sumcost + nh;
	Pred is_same = samec == U32(16u);
	// mins rle (:478-490): 2 + non-repeated mins < mins that would be written
	Pred minsrle = pred_all(rle) & ((U32(18u) - eqc) < (U32(16u) - count8));
	size = sel(minsrle, size - ((U32(16u) - count8) - (U32(18u) - eqc)), size);
	U32 type = sel(is_same, U32(PLANE_SAME), sel(minsrle, U32(PLANE_NORMAL_RLE), U32(PLANE_NORMAL)));
	size = sel(is_same, U32(1u), size);
	if (rle) { // target size 256 (:1190, 1200-1204)
		Pred raw = size > U32(256u);
		type = sel(raw, U32(PLANE_RAW), type);
		size = sel(raw, U32(256u), size);
	}
	U32 pay = sel(act, sel(nomin, cost, cost - 1u), U32(0u));
	Pred emit = act & ((minsrle & !eq) | (!minsrle & !nomin));
	U32 ex = row_excl_scan(pay | sel(emit, U32(1u << 16), U32(0u)));
	U32 minslen = sel(minsrle, U32(18u) - eqc, U32(lines) - count8);
	U32 poff = U32(nh) + minslen + (ex & 0xFFFFu);
	U32 minpos = U32(nh) + sel(minsrle, U32(2u), U32(0u)) + (ex >> 16);

	U32 ri = U32(L.rowinfo) + (U32(g) * 16u + lane) * 8u;
	lds_st32(lds, ri, hdr | (minv << 8) | (poff << 16), valid);
	lds_st32(lds, ri + 4u, minpos | sel(emit, U32(1u << 12), U32(0u)) | sel(eq, U32(1u << 13), U32(0u)), valid);
	if (praw) {
		const U32 pinfo = type | (size << PL_SHIFT);
		for (uint32_t k = 0; k < 4; ++k) { // constant indices keep praw in scalar registers
			const uint32_t v = readlane(pinfo, 16 * k);
			praw[k] = k < np ? v : praw[k];
		}
	}
	else
		lds_st32(lds, U32(L.plinfo) + (U32(g) + pl) * 4u, type | (size << PL_SHIFT), valid & (r == U32(0u)));
	wave_sync();
}

// Analyse the planes [g, g+np) (np <= 4).  rle: full-block mode (rle + raw override enabled);
// lines: number of rows that will be emitted (16 for full blocks).
// praw: when given (bytesoftype <= 4, a single group), the planes' type | size << PL_SHIFT come back as scalars instead of
// going through L.plinfo.
WV_FN void analyse_group(Lds lds, const Layout& L, uint32_t T, uint32_t g, uint32_t np, bool rle, uint32_t lines, const PlaneRegs& regs,
			 uint32_t* praw = nullptr)
{
	const U32 lane = lane_id();
	WV_MARK("analyse_stage1");
	if (regs.valid && np == 4) {
		if (regs.g == g)
			analyse_rows_int32(lds, L, regs);
		else
			analyse_rows_int32(lds, L, load_plane_regs(lds, L.in, T, g));
	}
	else
	// stage 1: element lanes, one plane at a time -> per-row statistics in L.aux
	for (uint32_t pj = 0; pj < np; ++pj) {
		U32 w = fetch_plane_word(lds, L.in, T, g + pj, regs);
		Pred leader = (lane & 3u) == U32(0u);
		U32 addr = U32(L.aux + pj * 128u) + (lane >> 2) * 8u;
		const uint32_t first = readlane(w, 0) & 0xFFu;
		if (!any(w != U32(first * 0x01010101u))) {
			// all 256 bytes equal (block_compress.h:396, 406, 415-418): statistics of a constant plane, nothing to measure
			lds_st32(lds, addr, U32((first ^ 0x80u) * 0x0101u), leader);
			lds_st32(lds, addr + 4u, U32(0u), leader);
			continue;
		}
		PlaneWords p = plane_words(w);
		U32 cnt = quad_add(popc(p.z1) | (popc(p.z2) << 16));
		U32 s = p.w ^ 0x80808080u, ds = p.dw ^ 0x80808080u; // signed order (:407-411)
		U32 mn = quad_min(min4(s)), mx = quad_max(max4(s));
		U32 dmn = quad_min(min4(ds)), dmx = quad_max(max4(ds));
		lds_st32(lds, addr, mn | (mx << 8) | (dmn << 16) | (dmx << 24), leader);
		lds_st32(lds, addr + 4u, cnt, leader);
	}
	wave_sync();
	analyse_stage2(lds, L, g, np, rle, lines, praw);
}
// ---- plane slots (bytesoftype 2 and 4) ----------------------------------------------------------------------
// Planes whose 256 bytes are all equal (SAME) need no analysis, and a block of 16-bit elements has only two planes:
// the row lanes of analyse_stage2 (four planes x 16 rows) would be half idle.  So the planes that do need it are
// written, in plane order, to up to four "slots" in LDS, which may come from two consecutive blocks; lane 4r + q
// then reads row r of slot q with one 16-byte load (no cross-lane transpose) and one pass decides all four slots.
struct SameScan {
	uint32_t act;   // bit k: plane k is not constant
	uint32_t nact;  // number of such planes
	uint32_t first; // first byte of planes 0..3, packed (the byte of a SAME plane)
};
// What the frame assembly needs to know about an encoded block to reproduce the reference's
// capacity rules (block_compress.h:1214, 1225, 1241; block_compress_partial :984, 994, 1013):
//   info = full | eligible << 30 | lz_ok << 31
//   full : sum of the plane sizes (the non-LZ encoding is header_bytes + full)
//   need : bytes of capacity, counted from the block's first byte, that the non-LZ encoding requires
//          (the largest of the reference's "dst + x > dst_end" tests)
//   eligible: the mini-LZ may be tried (T % 4 == 0 and 3 * full > 256 * T); lz_ok: it succeeded
struct BlockInfo {
	uint32_t size; // bytes of the emitted encoding
	uint32_t info;
	uint32_t need;
};
WV_HD uint32_t info_full(uint32_t info) { return info & 0x3FFFFFFFu; }
WV_HD bool info_eligible(uint32_t info) { return (info >> 30) & 1u; }
WV_HD bool info_lz_ok(uint32_t info) { return (info >> 31) & 1u; }

// Turn plane sizes into plane offsets (relative to the block start, after the type nibbles).
// Returns the sum of the plane sizes ("full_size", block_compress.h:1189-1207); *need receives the
// capacity requirement of the plane loop: max(hs + full, max over non-RAW planes of offset + size + slack)
// with slack 16 for full blocks (:1241) and, for partial blocks, the reference's
// 8 + (8 + sum of row costs) test (:989-995) or offset + 1 for SAME planes (:984).
WV_FN uint32_t plane_offsets(Lds lds, const Layout& L, uint32_t T, bool full_block, uint32_t lines, uint32_t* need)
{
	const U32 lane = lane_id();
	uint32_t full = 0, m = 0;
	for (uint32_t c = 0; c < T; c += 64) { // 64 planes at a time
		const U32 j = U32(c) + lane;
		const Pred valid = j < U32(T);
		U32 pi = sel(valid, lds_ld32(lds, U32(L.plinfo) + sel(valid, j, U32(0u)) * 4u), U32(0u));
		U32 type = pi & PL_TYPE;
		U32 size = pi >> PL_SHIFT;
		U32 incl = wave_incl_scan(size);
		U32 off = U32(header_bytes(T) + full) + incl - size;
		lds_st32(lds, U32(L.plinfo) + j * 4u, type | (off << PL_SHIFT), valid);
		U32 req;
		if (full_block)
			req = sel(valid & (type != U32(PLANE_RAW)), off + size + 16u, U32(0u));
		else {
			const uint32_t nh = (lines + 1) >> 1;
			req = sel(valid, sel(type == U32(PLANE_SAME), off + 1u, off + (size - nh + 8u) + 8u), U32(0u));
		}
		const uint32_t mc = wave_max(req);
		m = mc > m ? mc : m;
		full += readlane(incl, 63);
	}
	wave_sync();
	const uint32_t whole = header_bytes(T) + full;
	*need = m > whole ? m : whole;
	return full;
}

// The same on scalars for bytesoftype <= 4: tab[k] = type | size << PL_SHIFT on entry, type | offset << PL_SHIFT on return.
WV_HD uint32_t plane_offsets_small(uint32_t T, bool full_block, uint32_t lines, uint32_t* tab, uint32_t* need)
{
	const uint32_t hs = header_bytes(T), nh = (lines + 1) >> 1;
	uint32_t off = hs, m = 0;
	for (uint32_t k = 0; k < T; ++k) {
		const uint32_t type = tab[k] & PL_TYPE, size = tab[k] >> PL_SHIFT;
		uint32_t req;
		if (full_block)
			req = type != PLANE_RAW ? off + size + 16u : 0u;
		else
			req = type == PLANE_SAME ? off + 1u : off + (size - nh + 8u) + 8u;
		m = req > m ? req : m;
		tab[k] = type | (off << PL_SHIFT);
		off += size;
	}
	*need = m > off ? m : off;
	return off - hs;
}

// Write the planes of an analysed block into the (zeroed) output image starting at byte `base`.
// tab: the scalar plane table of plane_offsets_small (bytesoftype <= 4) or null (L.plinfo holds it).
WV_FN void emit_planes(Lds lds, const Layout& L, uint32_t T, uint32_t base, uint32_t lines, const PlaneRegs& regs, const uint32_t* tab = nullptr)
{
	const U32 lane = lane_id();
	Lds out = lds + L.out;
	WV_MARK("emit_nibbles");
	// plane type nibbles (block_compress.h:1246-1257)
	if (tab) {
		uint32_t nib = 0;
		for (uint32_t k = 0; k < T; ++k)
			nib |= (tab[k] & 0xFu) << (4 * k);
		lds_put_bits(out, U32(base * 8u), U32(nib), lane == U32(0u));
	}
	else
		for (uint32_t c = 0; c < T; c += 64) {
			const U32 j = U32(c) + lane;
			Pred valid = j < U32(T);
			U32 pi = sel(valid, lds_ld32(lds, U32(L.plinfo) + sel(valid, j, U32(0u)) * 4u), U32(0u));
			lds_put_small(out, U32(base * 8u) + j * 4u, pi & 0xFu, valid);
		}
	// row lanes: row-header nibbles, mins, SAME byte
	WV_MARK("emit_rowlanes");
	for (uint32_t g = 0; g < T; g += 4) {
		const uint32_t np = T - g < 4 ? T - g : 4;
		const U32 r = lane & 15u, pl = lane >> 4;
		const Pred valid = pl < U32(np);
		U32 lo, hi;
		U32 pi, type, pbase;
		lds_ld64(lds, U32(L.rowinfo) + (U32(g) * 16u + lane) * 8u, lo, hi);
		pi = tab ? row_select4(tab[0], tab[1], tab[2], tab[3]) : lds_ld32(lds, U32(L.plinfo) + sel(valid, U32(g) + pl, U32(0u)) * 4u);
		type = pi & PL_TYPE;
		pbase = U32(base) + (pi >> PL_SHIFT); // byte offset of this plane in the image
		Pred normal = valid & ((type == U32(PLANE_NORMAL)) | (type == U32(PLANE_NORMAL_RLE)));
		Pred act = r < U32(lines);
		U32 hdr = lo & 0xFFu, minv = (lo >> 8) & 0xFFu;
		lds_put_small(out, pbase * 8u + r * 4u, hdr, normal & act); // (:768-779, 758-762)
		Pred emit = ((hi >> 12) & 1u) == U32(1u);
		lds_put_small(out, (pbase + (hi & 0xFFFu)) * 8u, minv, normal & emit);
		// mins rle mask (:765): bit r = min equals previous min
		uint64_t eqb = ballot(((hi >> 13) & 1u) == U32(1u));
		U32 m16 = sel(pl == U32(0u), U32((uint32_t)(eqb & 0xFFFF)),
			      sel(pl == U32(1u), U32((uint32_t)((eqb >> 16) & 0xFFFF)),
				  sel(pl == U32(2u), U32((uint32_t)((eqb >> 32) & 0xFFFF)), U32((uint32_t)(eqb >> 48)))));
		lds_put_bits(out, (pbase + 8u) * 8u, m16, valid & (type == U32(PLANE_NORMAL_RLE)) & (r == U32(0u)));
		// SAME: the plane's byte; every row has mx == mn so minv is that byte (:747-750)
		lds_put_small(out, pbase * 8u, minv, valid & (type == U32(PLANE_SAME)) & (r == U32(0u)));
	}
	// element lanes: row payloads.  Every lane contributes one piece per plane (its 4 raw bytes, its 4
	// packed values or its rle literals); rle rows add their 4 mask bits.
	const U32 row = lane >> 2, q = lane & 3u;
	PlaneRegs cur = regs;
	for (uint32_t j = 0; j < T; ++j) {
		WV_MARK("emit_plane");
		if (regs.valid && j % 4 == 0 && j != cur.g)
			cur = load_plane_regs(lds, L.in, T, j);
		uint32_t pi = tab ? tab[j] : readlane(lds_ld32(lds, U32(L.plinfo + j * 4u)), 0);
		uint32_t type = pi & PL_TYPE;
		uint32_t pbase = base + (pi >> PL_SHIFT);
		if (type == PLANE_SAME)
			continue;
		U32 w = fetch_plane_word(lds, L.in, T, j, cur);
		if (type == PLANE_RAW) {
			lds_put_bits(out, (U32(pbase) + lane * 4u) * 8u, w, pred_all(true));
			continue;
		}
		U32 lo = lds_ld32(lds, U32(L.rowinfo + j * 128u) + row * 8u);
		U32 hdr = lo & 0xFFu, minv = (lo >> 8) & 0xFFu;
		U32 rbase = U32(pbase) + (lo >> 16);
		Pred act = row < U32(lines);
		// deltas against the previous byte in plane order, 0 before the plane (block_compress.h:399-401)
		U32 dw = bytes_sub(w, (w << 8) | (shfl_up(w, 1, 0) >> 24));
		Pred is15 = hdr == U32(15u), is7 = hdr == U32(7u), is6 = hdr == U32(6u);
		// bit-packed rows (:562-602, 649-664): two halves of 8 values, `bits` bytes each
		U32 bits = hdr & 7u;
		U32 v = bytes_sub(sel(hdr < U32(8u), w, dw), bytes_splat(minv));
		U32 x = byte_of(v, 0) | (byte_of(v, 1) << bits) | (byte_of(v, 2) << (bits * 2u)) | (byte_of(v, 3) << (bits * 3u));
		U32 piece = sel(is15, w, x); // raw row (:674-676): the 4 bytes as they are
		U32 bitpos = sel(is15, (rbase + q * 4u) * 8u, (rbase + (q >> 1) * bits) * 8u + (q & 1u) * bits * 4u);
		Pred emit = act & (is15 | (!is7 & !is6 & (bits != U32(0u))));
		// rle / delta-rle rows (:258-265, 285-293): [mask16][literals]
		Pred isr = act & (is7 | is6);
		if (any(isr)) {
			U32 z1 = bytes_zero_mask(dw); // byte == previous byte (:268-275)
			U32 pd = sel(q == U32(0u), U32(0u), shfl_up(dw, 1, 0) >> 24);
			U32 z2 = bytes_zero_mask(dw ^ ((dw << 8) | pd)); // delta == previous delta, 0 before the row (:248-255)
			U32 f = zero_mask_to_bits(sel(is7, z1, z2));
			U32 nlit = U32(4u) - popc(f);
			U32 before = quad_add(sel(isr, nlit << (q << 3), U32(0u))); // literal counts of the quad's lanes
			U32 prior = (before & ((U32(1u) << (q << 3)) - 1u));
			prior = (prior & 0xFFu) + ((prior >> 8) & 0xFFu) + ((prior >> 16) & 0xFFu);
			lds_put_small(out, rbase * 8u + q * 4u, f, isr);
			piece = sel(isr, compact_unflagged(sel(is7, w, dw), f), piece);
			bitpos = sel(isr, (rbase + 2u + prior) * 8u, bitpos);
			emit = emit | (isr & (nlit != U32(0u)));
		}
		lds_put_bits(out, bitpos, piece, emit);
	}
	WV_MARK("emit_end");
	wave_sync();
}

// zero `bytes` (multiple of 16, 16-byte aligned) of LDS at `off`
WV_FN void lds_zero(Lds lds, uint32_t off, uint32_t bytes)
{
	const U32 lane = lane_id();
	U128 z;
	z.x = z.y = z.z = z.w = U32(0u);
	for (uint32_t o = 0; o < bytes; o += 1024)
		lds_st128(lds, U32(off + o) + lane * 16u, z, (U32(o) + lane * 16u) < U32(bytes));
	wave_sync();
}

// Zeroed image for an encoding of `size` bytes that starts at byte `base` (< 16) of the image.  A block that is
// appended to a stream (superblock_codec.h, RunStream) starts behind the bytes of its predecessor that did not fill
// a 16-byte group: they wait, padded with zeros, in the 16 bytes in front of the image and become its first group.
WV_FN void image_reset(Lds lds, const Layout& L, uint32_t base, uint32_t size)
{
	lds_zero(lds, L.out, align16(base + size) + 16u);
	if (base) {
		const U32 lane = lane_id();
		Pred p = lane < U32(4u);
		U32 a = sel(p, lane, U32(0u)) * 4u;
		lds_st32(lds, U32(L.out) + a, lds_ld32(lds, U32(L.out - 16u) + a), p);
		wave_sync();
	}
}

// The same for the slot encoder (bytesoftype 2 and 4, slot_codec.h): the whole image is cleared by a fixed number of
// stores and the waiting bytes are always taken over -- no loop, no size arithmetic, no branch in the scalar unit, which is
// as busy as the vector unit in that encoder.  The 16 bytes in front of the image are zero when nothing waits
// (stream_begin, stream_append).
WV_FN void image_reset_fixed(Lds lds, const Layout& L, uint32_t T)
{
	const U32 lane = lane_id();
	U128 z;
	z.x = z.y = z.z = z.w = U32(0u);
	const uint32_t bytes = out_capacity(T); // 1088 or 1168: one full store and a partial one
	lds_st128(lds, U32(L.out) + lane * 16u, z, pred_all(true));
	lds_st128(lds, U32(L.out + 1024u) + lane * 16u, z, lane * 16u < U32(bytes - 1024u));
	wave_sync();
	const Pred p = lane < U32(4u);
	const U32 a = sel(p, lane, U32(0u)) * 4u;
	lds_st32(lds, U32(L.out) + a, lds_ld32(lds, U32(L.out - 16u) + a), p);
	wave_sync();
}

// ------------------------------------------------------------------------------------------------
// mini-LZ (lz_compress.h:161-232).  Positions are visited 64 at a time (lane = pos & 63).
// ------------------------------------------------------------------------------------------------

struct LzVal {
	U32 lo, hi;
};
WV_FN LzVal lz_value(Lds lds, uint32_t in, uint32_t B, const U32& pos)
{
	LzVal v;
	if (B == 8)
		lds_ld64(lds, U32(in) + pos * 8u, v.lo, v.hi);
	else {
		v.lo = lds_ld32(lds, U32(in) + pos * 4u);
		v.hi = U32(0u);
	}
	return v;
}
// hash_val / hash_val64 (lz_compress.h:47-56)
WV_FN U32 lz_hash(const LzVal& v, uint32_t B)
{
	if (B == 8) {
		// (v * 14313749767032793493) >> 56 ; K = 0xC6A4A7935BD1E995
		const uint32_t klo = 0x5BD1E995u, khi = 0xC6A4A793u;
		U32 top = v.lo * khi + v.hi * klo + mulhi(v.lo, U32(klo));
		return top >> 24;
	}
	return (v.lo * 2654435761u) & 255u;
}

// Try to encode the block held at L.in with the mini-LZ.  Returns the number of bytes produced
// (after the 253 marker) or 0 when the reference would give up.  On success the stream has been
// written to the zeroed output image at byte base+1 and the marker at base.
// First rejection test of the mini-LZ (most attempts end here).  Every value costs B bytes unless it matches (>= 1 byte)
// and only values with an earlier same-hash value can match; each group adds its flag byte (lz_compress.h:203-219).
// Counting the distinct hash keys among the first nq values gives a lower bound of the bytes produced when the
// reference runs its early-stop test; if even that bound fails the test (:221-229) the attempt is over without
// building the chains.  nq: values covered up to and including the first group whose start index exceeds count/4.
WV_HD uint32_t lz_precheck_values(uint32_t T)
{
	const uint32_t count = 256 * T / lz_width(T);
	return 8 * ((count / 4) / 8 + 2);
}
WV_HD bool lz_precheck_passes(uint32_t T, uint32_t distinct, uint32_t max_size)
{
	const uint32_t B = lz_width(T), nq = lz_precheck_values(T);
	const uint32_t lower = nq / 8 + nq * B - (nq - distinct) * (B - 1);
	// the reference compares doubles, lower > max_size * 0.4 (lz_compress.h:226); for these small integers that is
	// 5 * lower > 2 * max_size: the product is exact when max_size is a multiple of 5 and at least 0.2 away from an integer otherwise.
	// (Its other test, lower > max_size (:221-223), is implied: it would make 5 * lower > 5 * max_size.)
	return !(5 * lower > 2 * max_size);
}
// distinct hash keys among the first nq values of the block at L.in; uses the first KiB of the image (not L.lz)
WV_FN uint32_t lz_distinct_keys(Lds lds, const Layout& L, uint32_t T)
{
	const U32 lane = lane_id();
	const uint32_t B = lz_width(T);
	const uint32_t tab = L.tab;
	const uint32_t nq = lz_precheck_values(T);
	WV_MARK("lz_try");
	U128 z;
	z.x = z.y = z.z = z.w = U32(0u);
	lds_st128(lds, U32(tab) + lane * 16u, z, pred_all(true)); // 256 counters
	wave_sync();
	uint32_t distinct = 0;
	for (uint32_t c = 0; c <= (nq - 1) / 64; ++c) {
		const U32 pos = U32(c * 64u) + lane;
		Pred in = pos < U32(nq);
		U32 key = lz_hash(lz_value(lds, L.in, B, sel(in, pos, U32(0u))), B);
		U32 old = lds_add_rtn32(lds, U32(tab) + key * 4u, U32(1u), in);
		distinct += (uint32_t)__builtin_popcountll(ballot(in & (old == U32(0u))));
	}
	wave_sync();
	return distinct;
}
WV_FN bool lz_precheck(Lds lds, const Layout& L, uint32_t T, uint32_t max_size) { return lz_precheck_passes(T, lz_distinct_keys(lds, L, T), max_size); }
// *scratch_used (optional) is set once the attempt gets past its first rejection test and starts using L.lz.
WV_FN uint32_t lz_try(Lds lds, const Layout& L, uint32_t T, uint32_t max_size, uint32_t base, bool* scratch_used = nullptr)
{
	const U32 lane = lane_id();
	const uint32_t B = lz_width(T);
	const uint32_t count = 256 * T / B, nchunks = count / 64;
	const uint32_t tab = L.tab, chain = L.lz, cur = chain + count * 4; // the image is written last: it serves as the table until then
	const uint32_t quarter = count / 4; // the early-stop test fires at the first group start i > count/4
	const uint32_t sampled_keys = lz_distinct_keys(lds, L, T);
	if (!lz_precheck_passes(T, sampled_keys, max_size))
		return 0;
	// The two tests that follow only ever end an attempt early that would fail anyway; they are work for nothing where the
	// attempt succeeds.  Where three out of four sampled values repeat a hash key (dictionary-like data, the mini-LZ's own
	// ground) they are left out, and the exact count is only taken where the cheap bound comes close to the limit.
	const bool plenty = sampled_keys * 4 <= lz_precheck_values(T);
	bool close = false;
	WV_MARK("lz_test2");
	if (!plenty) {
		// Second rejection test, for data whose values hardly repeat (noise, floats): a value can only match when an
		// equal value precedes it, and equal values have equal hashes.  A 13-bit hash and one bit per hash value (1 KiB,
		// the image area again) give an upper bound `maybe` of the values with an equal predecessor, so the whole stream
		// takes at least ngroups + count*B - maybe*(B-1) bytes; above max_size the reference fails (lz_compress.h:221-223)
		// after doing all the work.  Leaves L.lz alone.
		U128 z;
		z.x = z.y = z.z = z.w = U32(0u);
		lds_st128(lds, U32(tab) + lane * 16u, z, pred_all(true));
		wave_sync();
		uint32_t maybe = 0;
		for (uint32_t c = 0; c < nchunks; ++c) {
			LzVal v = lz_value(lds, L.in, B, U32(c * 64u) + lane);
			U32 h = ((v.lo ^ (v.hi * 0x9E3779B1u)) * 0x85EBCA6Bu) >> 19;
			U32 bit = U32(1u) << (h & 31u);
			U32 old = lds_or_rtn32(lds, U32(tab) + (h >> 5) * 4u, bit);
			maybe += (uint32_t)__builtin_popcountll(ballot((old & bit) != U32(0u)));
		}
		wave_sync();
		const uint32_t least = count / 8 + count * B - maybe * (B - 1);
		if (least > max_size)
			return 0;
		close = least + 64 > max_size; // (13 hash bits over 256 or 512 values: a handful of false repeats at most)
	}
	if (scratch_used)
		*scratch_used = true;

	WV_MARK("lz_test3");
	if (close) {
		// Third rejection test: the same bound with the exact number `dups` of values that have an equal predecessor.
		// Distinct values are counted with an open-addressing table of positions (2*count slots in the chain/candidate
		// area, linear probing, LDS compare-and-swap).
		const uint32_t slots = 2 * count;
		for (uint32_t o = 0; o < slots * 4; o += 1024) {
			U128 none;
			none.x = none.y = none.z = none.w = U32(LZ_NONE);
			lds_st128(lds, U32(chain + o) + lane * 16u, none, (U32(o) + lane * 16u) < U32(slots * 4));
		}
		wave_sync();
		uint32_t dups = 0;
		for (uint32_t c = 0; c < nchunks; ++c) {
			const U32 pos = U32(c * 64u) + lane;
			LzVal v = lz_value(lds, L.in, B, pos);
			U32 h = mulhi((v.lo ^ (v.hi * 0x9E3779B1u)) * 0x85EBCA6Bu, U32(slots));
			Pred todo = pred_all(true);
			Pred dup = pred_all(false);
			while (any(todo)) {
				U32 found = lds_cas32(lds, U32(chain) + h * 4u, U32(LZ_NONE), pos, todo);
				Pred occupied = todo & (found != U32(LZ_NONE));
				LzVal o = lz_value(lds, L.in, B, sel(occupied, found, U32(0u)));
				Pred same = occupied & (o.lo == v.lo) & (o.hi == v.hi);
				dup = dup | same;
				todo = occupied & !same; // inserted or duplicate: done; different value: next slot
				h = sel(h + 1u == U32(slots), U32(0u), h + 1u);
			}
			dups += (uint32_t)__builtin_popcountll(ballot(dup));
		}
		wave_sync();
		if (count / 8 + count * B - dups * (B - 1) > max_size)
			return 0;
	}

	WV_MARK("lz_pass1");
	// empty table: every entry "no position"
	{
		U128 none;
		none.x = none.y = none.z = none.w = U32(LZ_NONE);
		lds_st128(lds, U32(tab) + lane * 16u, none, pred_all(true));
	}
	wave_sync();

	uint32_t failed = 0, max_failed = 3, produced = 0, nskipped = 0;
	bool once = false;
	// skip bits (one per group)
	const uint32_t skipbits = L.skip;
	for (uint32_t o = 0; o < lz_skip_bytes(T); o += 256)
		lds_st32(lds, U32(skipbits + o) + lane * 4u, U32(0u), U32(o) + lane * 4u < U32(lz_skip_bytes(T)));
	wave_sync();
	auto skipped = [&](const U32& h) -> Pred {
		Pred has = h != U32(LZ_NONE);
		U32 gidx = sel(has, h >> 3, U32(0u));
		U32 word = lds_ld32(lds, U32(skipbits) + (gidx >> 5) * 4u);
		return has & (((word >> (gidx & 31u)) & 1u) == U32(1u));
	};

	// ---- pass 1: chain[pos] = nearest earlier position with the same hash (what the table would hold if
	// no group were skipped)
	{
		U128 z;
		z.x = z.y = z.z = z.w = U32(0u);
		lds_st128(lds, U32(cur) + lane * 16u, z, pred_all(true)); // the 256 class words
		wave_sync();
	}
	for (uint32_t c = 0; c < nchunks; ++c) {
		const U32 pos = U32(c * 64u) + lane;
		LzVal v = lz_value(lds, L.in, B, pos);
		U32 key = lz_hash(v, B);
		// lanes of this chunk with the same key: every lane sets its bit in its key's word of a table of 256 (the area of the
		// second pass's candidates, not in use yet), lanes 0-31 first, then lanes 32-63, and reads the word back -- two LDS
		// round trips instead of eight ballots and their selects
		const U32 slot = U32(cur) + key * 4u, mybit = U32(1u) << (lane & 31u);
		lds_or32(lds, slot, mybit, lane < U32(32u));
		wave_sync();
		const U32 cls_lo = lds_ld32(lds, slot);
		wave_sync();
		lds_st32(lds, slot, U32(0u), pred_all(true));
		wave_sync();
		lds_or32(lds, slot, mybit, lane >= U32(32u));
		wave_sync();
		const U32 cls_hi = lds_ld32(lds, slot);
		wave_sync();
		lds_st32(lds, slot, U32(0u), pred_all(true));
		// nearest lower lane with the same key, else the table
		U32 lmask = (U32(1u) << (lane & 31u)) - 1u;
		U32 below_lo = sel(lane < U32(32u), cls_lo & lmask, cls_lo);
		U32 below_hi = sel(lane < U32(32u), U32(0u), cls_hi & lmask);
		Pred has_intra = (below_lo | below_hi) != U32(0u);
		U32 intra = sel(below_hi != U32(0u), bitlen(below_hi) + 31u, bitlen(below_lo) - 1u);
		U32 tabv = lds_ld32(lds, U32(tab) + key * 4u);
		U32 H = sel(has_intra, U32(c * 64u) + intra, tabv);
		lds_st32(lds, U32(chain) + pos * 4u, H, pred_all(true));
		// the last lane of each class records its position (positions of groups that turn out to be
		// skipped are jumped over through the chain)
		U32 hmask = ~((U32(2u) << (lane & 31u)) - 1u);
		hmask = sel((lane & 31u) == U32(31u), U32(0u), hmask);
		U32 above_lo = sel(lane < U32(32u), cls_lo & hmask, U32(0u));
		U32 above_hi = sel(lane < U32(32u), cls_hi, cls_hi & hmask);
		lds_st32(lds, U32(tab) + key * 4u, pos, (above_lo | above_hi) == U32(0u));
		wave_sync();
	}

	WV_MARK("lz_pass2");
	// ---- pass 2: the groups in order
	for (uint32_t c = 0; c < nchunks; ++c) {
		const U32 pos = U32(c * 64u) + lane;
		LzVal v = lz_value(lds, L.in, B, pos);
		U32 H = lds_ld32(lds, U32(chain) + pos * 4u);
		// current candidate: follow the chain over groups that were skipped (not hashed)
		while (nskipped) {
			Pred s = skipped(H);
			if (!any(s))
				break;
			H = sel(s, lds_ld32(lds, U32(chain) + sel(s, H, U32(0u)) * 4u), H);
		}

		// All eight groups of the chunk at once where none of them can be left raw (a group is, when the count of groups without
		// a match has reached its limit: then the candidates of the lanes behind it change, and the groups go one by one below):
		// the usual case for data the mini-LZ is made for, where every group finds matches.
		{
			const Pred cand = H != U32(LZ_NONE);
			const LzVal hv = lz_value(lds, L.in, B, sel(cand, H, U32(0u)));
			const Pred m = cand & (hv.lo == v.lo) & (hv.hi == v.hi);
			const Pred far = m & ((pos - H) >= U32(128u));
			// groups without a match: fold every byte of the mask of matches onto its lowest bit
			uint64_t any8 = ballot(m);
			any8 |= any8 >> 4;
			any8 |= any8 >> 2;
			any8 |= any8 >> 1;
			const uint32_t empty = 8 - (uint32_t)__builtin_popcountll(any8 & 0x0101010101010101ull);
			if (failed + empty < max_failed) {
				// The bytes the chunk produces, as a running sum over the lanes (an item's bytes, and the flags byte with a group's
				// first lane): the sum behind a group's last lane is what the stream holds after that group.  The sizes only ever
				// grow, so the limit is tested once, behind the chunk (:221-223), and the early test (:224-229) at the one group it
				// belongs to -- one prefix sum in vector registers instead of eight groups of arithmetic on the scalar unit (two
				// dozen scalar instructions a group, 800 a block: scalar instructions cost half a vector instruction each here).
				const U32 isz = sel(m, sel(far, U32(2u), U32(1u)), U32(B)) + sel((lane & 7u) == U32(0u), U32(1u), U32(0u));
				const U32 upto = wave_incl_scan(isz);
				bool over = false;
				const uint32_t early = quarter / 8 + 1 > c * 8 ? quarter / 8 + 1 : c * 8; // the first group whose first value has an index above count / 4
				if (!once && early <= c * 8 + 7) {
					over = 5 * (produced + readlane(upto, 8 * (early - c * 8) + 7)) > 2 * max_size;
					once = true;
				}
				produced += readlane(upto, 63);
				over |= produced > max_size;
				failed += empty;
				if (over)
					return 0;
				lds_st32(lds, U32(cur) + pos * 4u, H, pred_all(true));
				wave_sync();
				continue;
			}
		}
		// the 8 groups of this chunk, in order
		for (uint32_t k = 0; k < 8; ++k) {
			const uint32_t g = c * 8 + k;
			const uint32_t i = g * 8; // index of the group's first value
			uint32_t gsize;
			if (failed == max_failed) { // raw group, not hashed (lz_compress.h:206-211)
				failed = 0;
				if (--max_failed == 0)
					max_failed = 1;
				lds_or32(lds, U32(skipbits + (g >> 5) * 4u), U32(1u << (g & 31)), lane == U32(0u));
				++nskipped;
				wave_sync();
				gsize = 1 + 8 * B;
				// later lanes of this chunk that pointed into the skipped group move down the chain
				for (;;) {
					Pred s = skipped(H) & (lane >= U32((k + 1) * 8));
					if (!any(s))
						break;
					H = sel(s, lds_ld32(lds, U32(chain) + sel(s, H, U32(0u)) * 4u), H);
				}
			}
			else {
				Pred ingroup = (lane >> 3) == U32(k);
				Pred cand = ingroup & (H != U32(LZ_NONE));
				LzVal hv = lz_value(lds, L.in, B, sel(cand, H, U32(0u)));
				Pred m = cand & (hv.lo == v.lo) & (hv.hi == v.hi);
				Pred far = m & ((pos - H) >= U32(128u));
				uint32_t nm = (uint32_t)__builtin_popcountll(ballot(m));
				uint32_t nf = (uint32_t)__builtin_popcountll(ballot(far));
				gsize = 1 + 8 * B - nm * (B - 1) + nf;
				failed += (nm == 0);
			}
			produced += gsize;
			if (produced > max_size) // (:221-223)
				return 0;
			if (!once && i > quarter) { // (:224-229)
				if (5 * produced > 2 * max_size) // produced > max_size * 0.4 in doubles, see lz_precheck_passes
					return 0;
				once = true;
			}
		}
		lds_st32(lds, U32(cur) + pos * 4u, H, pred_all(true));
		wave_sync();
	}

	WV_MARK("lz_write");
	// success: write the stream.  Items of a group follow its flag byte.
	image_reset(lds, L, base, produced + 1u); // the table is no longer needed
	Lds out = lds + L.out;
	lds_st8(out, U32(base), U32(BLOCK_LZ), lane == U32(0u));
	uint32_t run = base + 1; // byte offset of the next chunk's first group flag
	for (uint32_t c = 0; c < nchunks; ++c) {
		const U32 pos = U32(c * 64u) + lane;
		LzVal v = lz_value(lds, L.in, B, pos);
		U32 H = lds_ld32(lds, U32(cur) + pos * 4u);
		Pred cand = H != U32(LZ_NONE);
		if (nskipped)
			cand = cand & !skipped(pos);
		LzVal hv = lz_value(lds, L.in, B, sel(cand, H, U32(0u)));
		Pred m = cand & (hv.lo == v.lo) & (hv.hi == v.hi);
		U32 dist = pos - H;
		const Pred far = m & (dist >= U32(128u));
		U32 isz = sel(m, sel(far, U32(2u), U32(1u)), U32(B));
		U32 incl = wave_incl_scan(isz);
		uint64_t mb = ballot(m);
		// byte offset of this item: chunk base + flag bytes of groups up to mine + items before
		U32 ioff = U32(run) + (lane >> 3) + 1u + (incl - isz);
		// flag byte (first lane of each group)
		U32 mlo((uint32_t)mb), mhi((uint32_t)(mb >> 32));
		U32 flags = (sel(lane < U32(32u), mlo, mhi) >> (lane & 24u)) & 0xFFu;
		// Every byte of the stream is written by exactly one item or flag, each at its own byte address: plain byte stores
		// (nothing is OR-ed into the zeroed image here).
		lds_st8(out, ioff - 1u, flags, (lane & 7u) == U32(0u));
		// match: distance on 1 or 2 bytes (write_diff, :140-151)
		lds_st8(out, ioff, dist, m & !far);
		lds_st8(out, ioff, (dist & 127u) | 128u, far);
		lds_st8(out, ioff + 1u, dist >> 7, far);
		// raw value (byte by byte: the LDS does store 32 bits at any byte address, but an access that straddles a dword takes a slow
		// path -- ten times the latency with the device busy, tools/ubench_lds_bytes.hip; as stores here the two cost the same)
		for (uint32_t b = 0; b < 4; ++b)
			lds_st8(out, ioff + b, v.lo >> (8 * b), !m);
		if (B == 8)
			for (uint32_t b = 0; b < 4; ++b)
				lds_st8(out, ioff + 4u + b, v.hi >> (8 * b), !m);
		run += 8 + readlane(incl, 63);
	}
	wave_sync();
	return produced;
}

// Encode the full block at L.in into the output image (zeroed here), starting at byte `base` (< 16) of the image.
// allow_lz mirrors the reference's capacity condition for the LZ attempt (block_compress.h:1214).
WV_FN BlockInfo encode_full_block(Lds lds, const Layout& L, uint32_t T, bool allow_lz, uint32_t base = 0)
{
	WV_MARK("block_begin");
	const PlaneRegs regs = load_plane_regs(lds, L.in, T, 0);
	const bool small = T <= 4; // one plane group: its table lives in scalars
	uint32_t tab[4] = { 0u, 0u, 0u, 0u };
	for (uint32_t g = 0; g < T; g += 4)
			analyse_group(lds, L, T, g, T - g < 4 ? T - g : 4, true, 16, regs, small ? tab : nullptr);
	uint32_t need;
	WV_MARK("plane_offsets");
	uint32_t full = small ? plane_offsets_small(T, true, 16, tab, &need) : plane_offsets(lds, L, T, true, 16, &need);
	const bool eligible = T % 4 == 0 && T <= LZ_MAX_T && full * 3 > 256 * T; // (:1210; lz_compress.h:281-283)
	BlockInfo r;
	r.info = full | (eligible ? 1u << 30 : 0u);
	r.need = need;
	if (allow_lz && eligible) {
		uint32_t n = lz_try(lds, L, T, full, base);
		if (n) {
			r.size = n + 1;
			r.info |= 1u << 31;
			return r;
		}
	}
	WV_MARK("image_reset");
	image_reset(lds, L, base, header_bytes(T) + full); // a failed LZ attempt leaves its table there
	emit_planes(lds, L, T, base, 16, regs, small ? tab : nullptr);
	r.size = header_bytes(T) + full;
	return r;
}

// Encode the `lines` complete rows of a partial block (block_compress_partial, block_compress.h:947-1009).
// L.in must hold the tail padded to 256*T bytes with its last byte.  The image receives
// [254][plane types][planes] starting at byte 0; returns the bytes written (the caller appends the
// remaining raw bytes) and, in *need, the capacity the reference's tests require counted from the 254 byte.
WV_FN uint32_t encode_partial_lines(Lds lds, const Layout& L, uint32_t T, uint32_t lines, uint32_t* need)
{
	lds_zero(lds, L.out, out_capacity(T));
	lds_put_bits(lds + L.out, U32(0u), U32(BLOCK_PARTIAL), lane_id() == U32(0u));
	*need = 2; // dst + 2 > dst_end (:1284)
	if (lines == 0) {
		wave_sync();
		return 1;
	}
	const PlaneRegs regs = load_plane_regs(lds, L.in, T, 0);
	const bool small = T <= 4;
	uint32_t tab[4] = { 0u, 0u, 0u, 0u };
	for (uint32_t g = 0; g < T; g += 4)
		analyse_group(lds, L, T, g, T - g < 4 ? T - g : 4, false, lines, regs, small ? tab : nullptr);
	uint32_t pneed;
	uint32_t full = small ? plane_offsets_small(T, false, lines, tab, &pneed) : plane_offsets(lds, L, T, false, lines, &pneed);
	emit_planes(lds, L, T, 1, lines, regs, small ? tab : nullptr);
	if (1 + pneed > *need)
		*need = 1 + pneed;
	return 1 + header_bytes(T) + full;
}

// ------------------------------------------------------------------------------------------------
// decoder (block_compress.h:1488-1879, 2088-2175; lz_compress.h:234-277)
// ------------------------------------------------------------------------------------------------

constexpr uint32_t DEC_ERROR = 0xFFFFFFFFu;

struct DecLayout {
	uint32_t win;  // window of compressed bytes (+8 slack after the valid bytes)
	uint32_t img;  // decoded block, element major, 256*T bytes
	uint32_t lut;  // 16 v_perm_b32 selectors that put the literals of a run-length group back in their places (dec_write_lut)
	uint32_t total;
};

// uniform byte / LE16 reads from the window
WV_FN uint32_t win_u8(Lds lds, uint32_t addr) { return readlane(lds_ld8(lds, U32(addr)), 0); }
WV_FN uint32_t win_u16(Lds lds, uint32_t addr) { return win_u8(lds, addr) | (win_u8(lds, addr + 1) << 8); }

// store the four bytes of plane j owned by this element lane into the element-major image
WV_FN void store_plane_word(Lds lds, uint32_t img, uint32_t T, uint32_t j, const U32& w, const Pred& p)
{
	U32 a = U32(img + j) + lane_id_plain() * (4u * T);
	lds_st8(lds, a, byte_of(w, 0), p);
	lds_st8(lds, a + T, byte_of(w, 1), p);
	lds_st8(lds, a + 2u * T, byte_of(w, 2), p);
	lds_st8(lds, a + 3u * T, byte_of(w, 3), p);
}

// Per byte position k of a lane: out_k = A_k ? out_{k-1} + B_k : B_k  (mod 256).  A is a 4-bit
// field, Bw four packed bytes.  Returns the four outputs given the carry-in byte `c`.
WV_FN U32 chain_apply(const U32& A, const U32& Bw, const U32& c)
{
	U32 o0 = (byte_of(Bw, 0) + sel((A & 1u) != U32(0u), c, U32(0u))) & 0xFFu;
	U32 o1 = (byte_of(Bw, 1) + sel((A & 2u) != U32(0u), o0, U32(0u))) & 0xFFu;
	U32 o2 = (byte_of(Bw, 2) + sel((A & 4u) != U32(0u), o1, U32(0u))) & 0xFFu;
	U32 o3 = (byte_of(Bw, 3) + sel((A & 8u) != U32(0u), o2, U32(0u))) & 0xFFu;
	return o0 | (o1 << 8) | (o2 << 16) | (o3 << 24);
}
// compose (apply first f1 = (a1,b1), then f2 = (a2,b2)): packed as a<<8 | b
WV_FN U32 chain_compose(const U32& f1, const U32& f2)
{
	U32 a1 = f1 >> 8, a2 = f2 >> 8;
	U32 b = sel(a2 != U32(0u), (f1 + f2) & 0xFFu, f2 & 0xFFu);
	return ((a1 & a2) << 8) | b;
}
// exclusive scan of the lane functions; `seg` = scan restarts at lanes where (lane & seg_mask) == 0
// (seg_mask 63: whole wave, 3: per quad).  Returns the carry-in byte of each lane (0 at a start).
WV_FN U32 chain_carry(const U32& A, const U32& Bw, uint32_t seg_mask)
{
	const U32 lane = lane_id_plain();
	// the lane's own function: value of its last byte for carry-in 0, and whether the carry goes through
	U32 last = chain_apply(A, Bw, U32(0u)) >> 24;
	const Pred through = (A & 0xFu) == U32(0xFu);
	if (seg_mask == 63) {
		// Whole wave.  A lane either hands its carry on plus a constant (all four flags set) or ends with a value of its own.
		// The value behind lane l = that of the nearest lane of the second kind up to l + the constants behind it: one prefix sum
		// of the constants, and a running maximum of keys (lane + 1) << 8 | (own value - prefix there) that hands every lane the
		// newest such lane -- twelve DPP steps of one instruction each instead of six rounds of function composition.
		const U32 P = wave_incl_scan(sel(through, last, U32(0u)));
		const U32 K = wave_incl_scan_max(sel(through, U32(0u), ((lane + 1u) << 8) | ((last - P) & 0xFFu)));
		return shfl_up(K + P, 1, 0) & 0xFFu;
	}
	U32 f = (sel(through, U32(1u), U32(0u)) << 8) | last;
	const U32 ident(1u << 8);
	// inside the segments (quads: seg_mask 3): the row shifts, cut at the segment starts
	int step = 0;
	for (uint32_t d = 1; d <= seg_mask; d <<= 1, ++step) {
		U32 prev = sel((lane & U32(seg_mask)) >= U32(d), scan_source(f, step, 1u << 8), ident);
		f = chain_compose(prev, f);
	}
	U32 ex = shfl_up(f, 1, 1u << 8);
	ex = sel((lane & U32(seg_mask)) == U32(0u), ident, ex);
	return ex & 0xFFu; // carry-in for a start value of 0
}

// Entry f of the selector table: byte k of the result = flag bit k of f ? 0 : the next literal.  Lane f writes entry f.
// Behind it a second table of 32 entries for rows of run-length coded differences (decode_plane_packed): entry f
// (+ 16: no literal in front of the lane), byte k = pool byte [number of clear flags among bits 0..k of f], where pool byte 0
// is the literal in force when the lane begins (entries 16..31: zero instead, selector 0x0c).
WV_FN void dec_write_lut(Lds lds, const DecLayout& L)
{
	const U32 lane = lane_id_plain();
	U32 pat(0u), n(0u), pat2(0u);
	for (uint32_t k = 0; k < 4; ++k) {
		const Pred lit = ((lane >> k) & 1u) == U32(0u);
		pat = pat | (sel(lit, n, U32(0x0cu)) << U32(8u * k));
		n = n + sel(lit, U32(1u), U32(0u));
		pat2 = pat2 | (sel((n == U32(0u)) & (lane >= U32(16u)), U32(0x0cu), n) << U32(8u * k));
	}
	lds_st32(lds, U32(L.lut) + lane * 4u, pat, lane < U32(16u));
	lds_st32(lds, U32(L.lut + 64) + lane * 4u, pat2, lane < U32(32u));
	wave_sync();
}
// A plane of a full block without run-length rows (the usual shape), given its row headers: a row is bit-packed -- a
// minimum and (hdr & 7) * 2 payload bytes, absolute (hdr < 8) or made of differences (hdr >= 8) -- or raw (hdr 15: sixteen
// bytes, here an absolute row of 8-bit values with minimum 0) (block_compress.h:1650-1700).  Same result as the general
// form below, in fewer instructions:
//  - one scan over the rows (payload offsets, and the position of the row's minimum where raw rows have none);
//  - the lane's four values stay in four registers: absolute lanes pack them, difference lanes add them up on the way;
//  - the carry into a difference lane = last value of the nearest absolute row before it + the differences in between:
//    one prefix sum over the lanes, and for the absolute rows a key (row + 1) << 8 | (last value - prefix there) whose
//    running maximum over the rows (four steps) hands every later row the newest one.
//    No gather through the LDS crossbar.  Where only row 0 is absolute (or none), its last value is one readlane.
// has_rle: rows of run-length coded differences (hdr 6: mask16 + the differences that are not repeats, block_compress.h:
// 1678-1690) among them.  Their sizes come from their masks, read row by row as in the general form; then such a row is a
// row of differences with 8-bit values and no minimum, and a lane finds its four by position: the difference in force when
// the lane begins is the literal in front of its own ones (none: 0), so the five bytes from there on and a table entry for
// the lane's four flags (dec_write_lut: byte k = pool byte [number of clear flags up to k]) are one v_perm_b32.  No chain.
// (Three copies -- plain, raw rows, raw and run-length rows: merged, the compiler turns the extras into selects that every plane pays.)
template <bool has_raw, int has_rle> // has_rle: 0 no run-length rows, 1 of differences only (header 6), 2 of values too (header 7)
WV_FN uint32_t decode_plane_packed(Lds lds, const DecLayout& L, uint32_t T, uint32_t j, uint32_t type, uint32_t cur, const U32& hdr, U32* keep)
{
	const U32 lane = lane_id_plain();
	Lds win = lds + L.win;
	const U32 row = lane >> 2, q = lane & 3u;
	U32 bits = hdr & 7u, m = hdr >> 3; // m = 1: a row of differences
	U32 bytes = bits + bits;
	U32 upto, minv, minat = row, nomin(~0u), rowoff, rmask(0u), lead(0u);
	uint32_t minslen = 16, rle_total = 0;
	Pred isr = pred_all(false);
	if (has_raw) {
		WV_NESTED();
		U32 raw = (hdr + 1u) >> 4, plain = raw ^ 1u;
		bits = bits + raw;
		m = m ^ raw;
		if (has_rle) {
			isr = has_rle == 2 ? (hdr & 0xEu) == U32(6u) : hdr == U32(6u);
			plain = sel(isr, U32(0u), plain);
			bits = sel(isr, U32(0u), bits); // (no payload of known size: see below)
			m = sel(hdr == U32(6u), U32(1u), m);
		}
		nomin = U32(0u) - plain; // (no minimum is added to a raw row or a run-length row)
		bytes = bits + bits;
		const U32 both = quads_incl_scan(bytes | (plain << 16)); // rows with a minimum, counted in the upper half
		upto = both & 0xFFFFu;
		minat = (both >> 16) - plain;
		minslen = readlane(both, 63) >> 16;
	}
	else
		upto = quads_incl_scan(bytes);
	if (type == PLANE_NORMAL)
		minv = lds_ld8(win, U32(cur + 8) + minat);
	else { // NORMAL_RLE: mask16, then the minimums that differ from the one before (every row has one here)
		const uint32_t mask = readlane(lds_ld32_unaligned(win, U32(cur + 8)), 0) & 0xFFFFu;
		minslen = 2 + 16 - (uint32_t)__builtin_popcount(mask);
		U32 idx = popc((~U32(mask)) & ((U32(2u) << row) - 1u) & 0xFFFFu);
		minv = sel(idx == U32(0u), U32(0u), lds_ld8(win, U32(cur + 9) + idx)); // (idx 0 reads the mask's second byte and drops it)
	}
	if (has_raw)
		minv = minv & nomin;
	rowoff = U32(cur + 8 + minslen) + (upto - bytes);
	if (has_rle) {
		WV_NESTED();
		// (the same walk on scalars -- the payload read into registers once, two v_readlane_b32 and a 64-bit shift per row instead
		// of two LDS reads -- was measured: +5 % on float32 sine, whose planes have one or two such rows, and no gain on frames
		// made of run-length rows, whose time is the mini-LZ's)
		uint64_t todo = ballot(isr) & 0x1111111111111111ull; // one bit per row: that of its first lane
		U32 extra(0u), total(0u);
		if (todo == 0x1111111111111111ull) {
			// Every row of the plane is a run-length row (long runs, steps, piecewise linear data: whole frames are made of such
			// planes).  Sixteen fixed steps then: the position of the row's mask, the mask, its size -- no lane reads, no search
			// for the next row, no loop control, no selects: a row's lanes leave the walk with what the step of their row found.
			// (rowoff is the same in all lanes here: no row before has a payload of known size.)
			todo = 0;
			U32 at = rowoff;
			lds_rle_walk16(win, at, rmask);
			extra = at - rowoff;
			total = extra + (U32(18u) - popc(rmask)); // (in the last row's lanes: the size of all sixteen)
		}
		if (todo) {
			// run-length rows among rows of other kinds: in the order of the rows, each one step (lds_rle_walk_row: the lanes from
			// the row on read its mask, the lanes behind it add its size); `extra` is per lane what the run-length rows in front
			// of the lane's row take
			while (todo) {
				const uint32_t rl = (uint32_t)__builtin_ctzll(todo);
				todo &= todo - 1;
				lds_rle_walk_row(win, readlane(rowoff, rl), rl, extra, rmask);
			}
			total = extra + sel(isr, U32(18u) - popc(rmask), U32(0u)); // (in the last row's lanes: the bytes of all of them)
		}
		rle_total = readlane(total, 63); // (the same in all lanes behind the loop; the last row's behind the fixed walk)
		rowoff = rowoff + extra;
	}
	const uint32_t psize = 8 + minslen + (readlane(upto, 63) & 0xFFFFu) + rle_total;
	// the lane's four values: 4 * bits bits from bit q * 4 * bits of the row's payload on
	U32 px = lds_ld32_bits(win, (rowoff << 3) + (mul24(q, bits) << 2));
	if (has_rle) {
		WV_NESTED();
		const U32 shift = q << 2;
		const U32 f = (rmask >> shift) & 0xFu;
		const U32 before = popc(~rmask & ((U32(1u) << shift) - 1u)); // literals of the row in front of this lane's
		const U32 at = rowoff + 1u + before;                         // the last of them (none: a byte of the mask, not used)
		U32 lo, hi;
		lds_ld64(win, at, lo, hi);
		const U32 pool_lo = funnel_shr(hi, lo, at << 3), pool_hi = hi >> ((at & 3u) << 3);
		const U32 selw = lds_ld32(lds, U32(L.lut + 64) + ((f | sel(before == U32(0u), U32(16u), U32(0u))) << 2));
		px = sel(isr, perm_bytes_v(pool_hi, pool_lo, selw), px);
		bits = sel(isr, U32(8u), bits);
		bytes = bits + bits;
		if (has_rle == 2) {
			// A row of run-length coded VALUES (header 7, :285-293) repeats the value in front of an element whose flag is set: the
			// literal in force, which the pool's first byte is -- or, in front of the row's first literal, the last value of the row
			// above.  A lane all of whose elements do that hands its carry on like a lane of zero differences; in a lane that holds
			// the row's first literal the elements in front of it take the carry (`lead`: their flags), the others are absolute.
			const Pred nofront = (hdr == U32(7u)) & (before == U32(0u));
			m = sel(nofront & (f == U32(0xFu)), U32(1u), m);
			lead = sel(nofront, f, U32(0u));
		}
	}
	const U32 v0 = bfe(px, U32(0u), bits) + minv, v1 = bfe(px, bits, bits) + minv, v2 = bfe(px, bytes, bits) + minv, v3 = bfe(px, bytes + bits, bits) + minv;
	U32 o0 = v0, o1 = v1, o2 = v2, o3 = v3;
	const uint64_t diff_rows = ballot(m != U32(0u));
	if (has_rle == 2) {
		// as below with the keys per lane (an absolute lane in front of the lane's own row may be the nearest one now), and the value
		// behind the lane in front for every lane: lanes of differences add it to all their elements, the `lead` elements take it
		o1 = mad24(m, o0, v1);
		o2 = mad24(m, o1, v2);
		o3 = mad24(m, o2, v3);
		const U32 P = wave_incl_scan(mul24(m, o3));
		const U32 K = wave_incl_scan_max(sel(m != U32(0u), U32(0u), ((lane + 1u) << 8) | ((o3 - P) & 0xFFu)));
		const U32 cin = shfl_up(K + P, 1, 0);
		const U32 l0 = lead & 1u, l1 = l0 & (lead >> 1), l2 = l1 & (lead >> 2), l3 = l2 & (lead >> 3);
		o0 = mad24(m | l0, cin, o0), o1 = mad24(m | l1, cin, o1), o2 = mad24(m | l2, cin, o2), o3 = mad24(m | l3, cin, o3);
	}
	else if (diff_rows) {
		// running sums inside the lane: difference lanes add up, absolute lanes keep their values (m = 0)
		o1 = mad24(m, o0, v1);
		o2 = mad24(m, o1, v2);
		o3 = mad24(m, o2, v3);
		const U32 S = mul24(m, o3);       // the lane's sum of differences, 0 in absolute rows
		const U32 P = wave_incl_scan(S);  // ... of all lanes up to and with this one
		U32 K;
		if ((~diff_rows & ~0xFull) == 0) // no absolute row behind row 0
			K = U32((diff_rows & 1u) ? 0u : readlane(o3, 3));
		else {
			const U32 last = quad_last(o3);
			const U32 key = sel(m != U32(0u), U32(0u), ((row + 1u) << 8) | ((last - P) & 0xFFu));
			K = quads_incl_scan_max(key);
		}
		const U32 carry = mul24(m, (K + P) - S); // (its low byte counts; 0 in absolute rows)
		o0 = o0 + carry, o1 = o1 + carry, o2 = o2 + carry, o3 = o3 + carry;
	}
	const U32 outw = perm_bytes(perm_bytes(o3, o2, 0x0c0c0400u), perm_bytes(o1, o0, 0x0c0c0400u), 0x05040100u);
	if (keep)
		*keep = outw;
	else
		store_plane_word(lds, L.img, T, j, outw, pred_all(true));
	// (no code: different last statements keep the compiler from folding the tails of the copies into one again)
	if (has_rle == 2)
		WV_MARK("dec_packed_rle7_end");
	else if (has_rle)
		WV_MARK("dec_packed_rle_end");
	else if (has_raw)
		WV_MARK("dec_packed_raw_end");
	else
		WV_MARK("dec_packed_end");
	return psize;
}

// A plane of a full block all of whose sixteen rows are run-length coded VALUES (header 7: long runs, steps -- whole frames are
// made of such planes).  No row has a minimum or a payload of known size, so the rows follow the eight header bytes directly
// and only the walk finds them; an element is the literal in force at its place -- the last literal of its row in front of it
// or, in front of the row's first literal, what the row above ended with: its last literal, or what IT began with if it has
// none.  That carry is the last literal of the nearest row above that has one: one byte read per row and one running maximum
// of keys over the rows.  With the carry put where the lane's pool has the literal in force, the selector table gives the
// lane's four elements in one v_perm_b32: no differences, no prefix sums (the general form above needs both for the rows
// that are not of this kind).  Same bytes consumed, same result as decode_plane_packed<true, 2>.
#ifdef WV_HOST_EMULATION
inline uint64_t& emul_plane_runs_count() // (tests: planes decoded by decode_plane_runs)
{
	static uint64_t n = 0;
	return n;
}
#endif
WV_FN uint32_t decode_plane_runs(Lds lds, const DecLayout& L, uint32_t T, uint32_t j, uint32_t cur, U32* keep)
{
#ifdef WV_HOST_EMULATION
	++emul_plane_runs_count();
#endif
	const U32 lane = lane_id_plain();
	Lds win = lds + L.win;
	const U32 row = lane >> 2, q = lane & 3u;
	U32 at(cur + 8), rmask(0u);
	lds_rle_walk16(win, at, rmask); // the row's mask and where it is
	const U32 nlit = U32(16u) - popc(rmask);
	const uint32_t psize = readlane(at + 2u + nlit, 63) - cur;
	const U32 shift = q << 2;
	const U32 f = (rmask >> shift) & 0xFu;
	const U32 before = popc(~rmask & ((U32(1u) << shift) - 1u)); // literals of the row in front of this lane's
	// the five bytes from the literal in force on (the last one in front of the lane's own; none: a byte of the mask, replaced below)
	const U32 from = at + 1u + before;
	U32 lo, hi;
	lds_ld64(win, from, lo, hi);
	U32 pool_lo = funnel_shr(hi, lo, from << 3);
	const U32 pool_hi = hi >> ((from & 3u) << 3);
	// what each row ends with, handed down to the rows below: key = row + 1 | last literal for the rows that have one
	const U32 last = lds_ld8(win, at + 1u + nlit);
	const U32 key = sel(nlit != U32(0u), ((row + 1u) << 8) | last, U32(0u));
	const U32 above = shfl_up(quads_incl_scan_max(key), 4, 0); // (of the row above: the newest row with a literal up to it; none: 0)
	pool_lo = sel(before == U32(0u), (pool_lo & ~U32(0xFFu)) | (above & 0xFFu), pool_lo);
	const U32 selw = lds_ld32(lds, U32(L.lut + 64) + (f << 2));
	const U32 outw = perm_bytes_v(pool_hi, pool_lo, selw);
	if (keep)
		*keep = outw;
	else
		store_plane_word(lds, L.img, T, j, outw, pred_all(true));
	WV_MARK("dec_runs_end");
	return psize;
}

// The same for a plane all of whose rows are run-length coded DIFFERENCES (header 6: piecewise linear data).  A row starts with
// no difference in force (:248-255), so the rows only meet in the running sum: the lane's four differences out of the selector
// table (entries 16..31: nothing in front of the lane's own literals), summed inside the lane, one prefix sum over the lanes.
// Same bytes consumed, same result as decode_plane_packed<true, 1>.
WV_FN uint32_t decode_plane_slopes(Lds lds, const DecLayout& L, uint32_t T, uint32_t j, uint32_t cur, U32* keep)
{
#ifdef WV_HOST_EMULATION
	++emul_plane_runs_count(); // (the two short forms count together)
#endif
	const U32 lane = lane_id_plain();
	Lds win = lds + L.win;
	const U32 q = lane & 3u;
	U32 at(cur + 8), rmask(0u);
	lds_rle_walk16(win, at, rmask);
	const uint32_t psize = readlane(at + (U32(18u) - popc(rmask)), 63) - cur;
	const U32 shift = q << 2;
	const U32 f = (rmask >> shift) & 0xFu;
	const U32 before = popc(~rmask & ((U32(1u) << shift) - 1u));
	const U32 from = at + 1u + before;
	U32 lo, hi;
	lds_ld64(win, from, lo, hi);
	const U32 pool_lo = funnel_shr(hi, lo, from << 3), pool_hi = hi >> ((from & 3u) << 3);
	const U32 selw = lds_ld32(lds, U32(L.lut + 64) + ((f | sel(before == U32(0u), U32(16u), U32(0u))) << 2));
	const U32 d = perm_bytes_v(pool_hi, pool_lo, selw);
	const U32 o0 = d & 0xFFu, o1 = o0 + ((d >> 8) & 0xFFu), o2 = o1 + ((d >> 16) & 0xFFu), o3 = o2 + (d >> 24);
	const U32 carry = wave_incl_scan(o3) - o3; // what the lanes in front add up to (its low byte counts)
	const U32 outw = perm_bytes(perm_bytes(o3 + carry, o2 + carry, 0x0c0c0400u), perm_bytes(o1 + carry, o0 + carry, 0x0c0c0400u), 0x05040100u);
	if (keep)
		*keep = outw;
	else
		store_plane_word(lds, L.img, T, j, outw, pred_all(true));
	WV_MARK("dec_slopes_end");
	return psize;
}

// Decode one NORMAL / NORMAL_RLE plane whose bytes start at window offset `cur` (at most `avail`
// valid bytes).  Writes rows [0, lines) of plane j into the image or, with `keep`, hands the lane's plane word back instead.
// Returns bytes consumed or DEC_ERROR.
WV_FN uint32_t decode_plane(Lds lds, const DecLayout& L, uint32_t T, uint32_t j, uint32_t type, uint32_t cur, uint32_t avail, uint32_t lines,
			    U32* keep = nullptr)
{
	const U32 lane = lane_id_plain();
	Lds win = lds + L.win;
	const uint32_t nh = (lines + 1) >> 1;
	WV_MARK("dec_plane_rows");
	// ---- row view: lane l looks at row l >> 2, the row of its four elements (the four lanes of a quad compute the same) ----
	const U32 row = lane >> 2, q = lane & 3u;
	const Pred act = row < U32(lines);
	// No size checks here: whatever the stream says, a plane reads at most 8 + 18 + 16*18 + 8 bytes from its start (the
	// window's buffer is followed by that much readable LDS, superblock_codec.h make_dec_layout) and the caller compares
	// the bytes consumed by the whole block with the bytes there are (block_compress.h:1702, 1724-1745, 2056, 2071-2084).
	(void)avail;
	uint32_t minslen;
	U32 hdr, minv;
	hdr = (lds_ld8(win, U32(cur) + (row >> 1)) >> ((row & 1u) << 2)) & 0xFu;
#ifndef STENOS_DECODE_NO_PACKED_PATH
	if (lines == 16) {
		// which copy: two bits per header out of a constant (0 plain, 1 raw, 2 and 3 the run-length kinds) -- two cheap
		// instructions and one compare for the usual plane instead of three compares
		const U32 kind = (U32(0x4000E000u) >> (hdr + hdr)) & 3u;
		if (ballot(kind != U32(0u)) != 0) {
			WV_NESTED();
			if (const uint64_t sevens = ballot(kind == U32(3u))) {
				if (sevens == ~0ull && type == PLANE_NORMAL)
					return decode_plane_runs(lds, L, T, j, cur, keep);
				return decode_plane_packed<true, 2>(lds, L, T, j, type, cur, hdr, keep);
			}
			if (const uint64_t sixes = ballot(kind == U32(2u))) {
				if (sixes == ~0ull && type == PLANE_NORMAL)
					return decode_plane_slopes(lds, L, T, j, cur, keep);
				return decode_plane_packed<true, 1>(lds, L, T, j, type, cur, hdr, keep);
			}
			return decode_plane_packed<true, 0>(lds, L, T, j, type, cur, hdr, keep);
		}
		return decode_plane_packed<false, 0>(lds, L, T, j, type, cur, hdr, keep);
	}
#endif
	if (type == PLANE_NORMAL) {
		Pred emit = act & (hdr != U32(6u)) & (hdr != U32(7u)) & (hdr != U32(15u));
		U32 e = sel(emit, U32(1u), U32(0u));
		U32 ex = quads_excl_scan(e);
		minslen = readlane(ex + e, 63); // the last row's inclusive value is the total
		minv = lds_ld8(win, U32(cur + nh) + sel(emit, ex, U32(0u)));
	}
	else { // NORMAL_RLE: 8 header bytes, mask16, non-repeated mins
		const uint32_t mask = readlane(lds_ld32_unaligned(win, U32(cur + 8)), 0) & 0xFFFFu; // (one window read; the two bytes behind it are literals or slack)
		uint32_t nlit = 16 - (uint32_t)__builtin_popcount(mask);
		minslen = 2 + nlit;
		// min[r] = literal of the last row r' <= r whose mask bit is 0, or 0 when there is none
		U32 upto = (~U32(mask)) & ((U32(2u) << row) - 1u) & 0xFFFFu;
		U32 idx = popc(upto);
		minv = sel(idx == U32(0u), U32(0u), lds_ld8(win, U32(cur + 10) + sel(idx == U32(0u), U32(0u), idx - 1u)));
	}
	// row payload sizes; rle rows need their mask, found by walking them in order
	Pred isrle = act & ((hdr == U32(6u)) | (hdr == U32(7u)));
	U32 known = sel(act & !isrle, sel(hdr == U32(15u), U32(16u), (hdr & 7u) * 2u), U32(0u));
	U32 pre = quads_excl_scan(known);
	const uint32_t totals = readlane(pre + known, 63);
	U32 rmask(0u), extra(0u);
	const uint32_t base = cur + nh + minslen;
	uint64_t todo = ballot(isrle) & 0x1111111111111111ull; // one bit per row: that of its first lane
	const bool has_rle = todo != 0, has_raw = ballot(act & (hdr == U32(15u))) != 0; // row kinds present: the element view skips what no row needs
	uint32_t rle_total = 0;
	while (todo) {
		uint32_t rl = (uint32_t)__builtin_ctzll(todo); // first lane of the row
		todo &= todo - 1;
		uint32_t addr = base + readlane(pre + extra, rl);
		uint32_t m = win_u16(win, addr);
		uint32_t sz = 2 + 16 - (uint32_t)__builtin_popcount(m);
		rmask = sel(row == U32(rl >> 2), U32(m), rmask);
		extra = extra + sel(row > U32(rl >> 2), U32(sz), U32(0u));
		rle_total += sz;
	}
	uint32_t psize = nh + minslen + (totals & 0xFFFFu) + rle_total;

	WV_MARK("dec_plane_elems");
	// ---- element view: lane l owns elements 4l..4l+3 of its row ----
	const U32 eh = hdr, emin = minv, eoff = U32(base) + pre + extra, emask = rmask;
	Pred eact = row < U32(lines);
	Pred e15 = eh == U32(15u), e7 = eh == U32(7u), e6 = eh == U32(6u);
	Pred erle = e7 | e6;
	// bit-packed value bytes
	U32 bits = eh & 7u;
	U32 bitoff = (q & 1u) * bits * 4u;
	U32 px = lds_ld32_unaligned(win, sel(eact, eoff + (q >> 1) * bits + (bitoff >> 3), U32(0u))) >> (bitoff & 7u);
	// value k + the row's minimum, mod 256: four sums, their low bytes gathered
	U32 v0 = bfe(px, U32(0u), bits) + emin, v1 = bfe(px, bits, bits) + emin, v2 = bfe(px, bits * 2u, bits) + emin, v3 = bfe(px, bits * 3u, bits) + emin;
	U32 packed = perm_bytes(perm_bytes(v3, v2, 0x0c0c0400u), perm_bytes(v1, v0, 0x0c0c0400u), 0x05040100u);
	// raw row bytes
	U32 rawv(0u);
	if (has_raw)
		rawv = lds_ld32_unaligned(win, sel(eact & e15, eoff + q * 4u, U32(0u)));
	// rle rows: flags of this lane and its literals
	U32 f(0u), rlev(0u), dv(0u);
	if (has_rle) {
		f = (emask >> (q << 2)) & 0xFu;
		U32 litidx = popc((~emask) & ((U32(1u) << (q << 2)) - 1u) & 0xFFFFu);
		U32 lits = lds_ld32_unaligned(win, sel(eact & erle, eoff + 2u + litidx, U32(0u)));
		rlev = perm_bytes_v(U32(0u), lits, lds_ld32(lds, U32(L.lut) + f * 4u)); // byte k = flag bit k ? 0 : next literal, through the table
		// delta-rle rows first rebuild their deltas: d_k = flag ? d_{k-1} : literal, d_{-1} = 0 per row
		dv = rlev;
		if (any(eact & e6)) {
			U32 A6 = sel(e6, f, U32(0u));
			U32 cin = chain_carry(A6, rlev, 3);
			dv = sel(e6, chain_apply(A6, rlev, cin), rlev);
		}
	}
	// final chain: absolute rows (A=0), delta rows (A=1), rle rows (A = flags)
	U32 A, Bw;
	if (!has_rle && !has_raw) { // bit-packed rows only, absolute or delta -- the common shape: nothing to choose between
		A = sel(eh >= U32(8u), U32(0xFu), U32(0u));
		Bw = packed;
	}
	else {
		Pred isdelta = !e15 & !erle & (eh >= U32(8u));
		A = sel(e6 | isdelta, U32(0xFu), sel(e7, f, U32(0u)));
		Bw = sel(e15, rawv, sel(e6, dv, sel(e7, rlev, packed)));
	}
	U32 outw;
	if (!any(A != U32(0u)))
		outw = Bw;
	else if (!has_rle) {
		// Only delta rows and absolute rows (no run-length rows: the common shape of slowly varying data).  A delta lane's
		// bytes are prefix sums of its four deltas plus the last byte before it; that byte is the last byte of the
		// nearest absolute lane before it plus the deltas in between: two wave scans (sum, position of the last absolute
		// lane) and one gather instead of six rounds of function composition.
		const Pred dl = A != U32(0u);
		const U32 sum4 = dot4_u8(Bw, 0x01010101u, U32(0u)); // d0 + d1 + d2 + d3, not reduced: 64 of them still fit 16 bits
		const U32 total = sel(dl, sum4, U32(0u));
		const U32 P = wave_incl_scan(total);
		const U32 start = wave_incl_scan_max(sel(dl, U32(0u), lane + 1u)); // 1 + the last absolute lane up to here, 0: none
		const U32 sprev = shfl_up(start, 1, 0), Pprev = shfl_up(P, 1, 0);
		const U32 g = shfl((P & 0xFFFFu) | ((Bw >> 24) << 16), sel(sprev == U32(0u), U32(0u), sprev - 1u));
		const U32 gP = sel(sprev == U32(0u), U32(0u), g & 0xFFFFu), gB = sel(sprev == U32(0u), U32(0u), g >> 16);
		const U32 carry = (gB + Pprev - gP) & 0xFFu;
		// byte k: carry + d0 + .. + dk (mod 256)
		const U32 p0 = dot4_u8(Bw, 0x00000001u, carry), p1 = dot4_u8(Bw, 0x00000101u, carry), p2 = dot4_u8(Bw, 0x00010101u, carry), p3 = sum4 + carry;
		outw = sel(dl, perm_bytes(perm_bytes(p3, p2, 0x0c0c0400u), perm_bytes(p1, p0, 0x0c0c0400u), 0x05040100u), Bw);
	}
	else {
		U32 cin = chain_carry(A, Bw, 63);
		outw = chain_apply(A, Bw, cin);
	}
	WV_MARK("dec_plane_store");
	if (keep)
		*keep = outw;
	else
		store_plane_word(lds, L.img, T, j, outw, eact);
	WV_MARK("dec_plane_end");
	return psize;
}

#ifdef WV_HOST_EMULATION
inline uint64_t& emul_lz_serial_count() // (tests: mini-LZ blocks the serial decoder did, lz_decode_256 having given up or not applying)
{
	static uint64_t n = 0;
	return n;
}
#endif
// mini-LZ stream -> image.  Returns bytes consumed (after the 253 marker) or DEC_ERROR.
WV_FN uint32_t lz_decode(Lds lds, const DecLayout& L, uint32_t T, uint32_t cur, uint32_t avail)
{
	const U32 lane = lane_id_plain();
	Lds win = lds + L.win;
	const uint32_t B = lz_width(T);
	const uint32_t count = 256 * T / B;
	const Pred item = lane < U32(8u);
	uint32_t p = cur;
	const uint32_t end = cur + avail;
	for (uint32_t i = 0; i < count; i += 8) {
		if (p + 2 > end) // lz_compress.h:242
			return DEC_ERROR;
		uint32_t flags = win_u8(win, p);
		// item sizes: raw = B, match = 1 or 2 (second byte when the first has bit 7 set)
		Pred m = item & (((U32(flags) >> lane) & 1u) == U32(1u));
		U32 isz = sel(m, U32(1u), U32(B));
		U32 off;
		uint32_t total;
		for (;;) { // every round fixes at least the first still-mispredicted 2-byte distance
			U32 incl = sel(item, isz, U32(0u));
			incl = incl + row_shr(incl, 1, 0);
			incl = incl + row_shr(incl, 2, 0);
			incl = incl + row_shr(incl, 4, 0);
			off = U32(p + 1) + incl - sel(item, isz, U32(0u));
			total = readlane(incl, 7);
			if (p + 1 + total > end)
				return DEC_ERROR;
			U32 first = lds_ld8(win, sel(m, off, U32(p)));
			Pred wrong = m & (first > U32(127u)) & (isz == U32(1u));
			uint64_t wb = ballot(wrong);
			if (!wb)
				break;
			// only the first mispredicted item sits at a trustworthy offset
			isz = sel(lane == U32((uint32_t)__builtin_ctzll(wb)), U32(2u), isz);
		}
		// values: raw items read the window, matches copy an earlier value of the image
		U32 d0 = lds_ld8(win, sel(m, off, U32(p)));
		U32 d1 = lds_ld8(win, sel(m, off + 1u, U32(p)));
		U32 dist = sel(isz == U32(2u), (d0 & 127u) | (d1 << 7), d0);
		U32 pos = U32(i) + lane;
		Pred bad = m & ((dist == U32(0u)) | (dist > pos));
		if (any(bad))
			return DEC_ERROR;
		U32 vlo = lds_ld32_unaligned(win, sel(item & !m, off, U32(p)));
		U32 vhi = B == 8 ? lds_ld32_unaligned(win, sel(item & !m, off + 4u, U32(p))) : U32(0u);
		// resolve sources inside this group (at most 7 hops), the rest comes from the image
		U32 src = sel(m, pos - dist, pos); // raw items are their own source
		for (int hop = 0; hop < 3; ++hop) {
			Pred ingroup = src >= U32(i);
			U32 s2 = shfl(src, sel(ingroup, src - U32(i), U32(0u)));
			src = sel(ingroup, s2, src);
		}
		Pred fromimg = item & (src < U32(i));
		U32 ilo, ihi(0u);
		if (B == 8)
			lds_ld64(lds, U32(L.img) + sel(fromimg, src, U32(0u)) * 8u, ilo, ihi);
		else
			ilo = lds_ld32(lds, U32(L.img) + sel(fromimg, src, U32(0u)) * 4u);
		U32 glo = shfl(vlo, sel(fromimg, U32(0u), src - U32(i)));
		U32 ghi = shfl(vhi, sel(fromimg, U32(0u), src - U32(i)));
		U32 olo = sel(fromimg, ilo, glo), ohi = sel(fromimg, ihi, ghi);
		if (B == 8) {
			lds_st32(lds, U32(L.img) + pos * 8u, olo, item);
			lds_st32(lds, U32(L.img) + pos * 8u + 4u, ohi, item);
		}
		else
			lds_st32(lds, U32(L.img) + pos * 4u, olo, item);
		wave_sync();
		p += 1 + total;
	}
	return p - cur;
}

// The same for 256 items (bytesoftype 4 and 8) in two phases instead of a chain of 32 groups with eight LDS round trips each.
// Phase 1, the chain: where a group starts depends on the sizes of the items in front of it, and a match is one byte or two.
// Taking every match for one byte makes a group's size a function of its flags byte alone: 32 dependent byte reads, three
// vector instructions each (lds_lz_walk32); if a match of two bytes shows up (a distance of 128 or more), the chain is walked
// again with the groups sized exactly, on scalars.
// Phase 2, all items at once, four per lane: offsets from the flags, distances and literals read; the literals go to their
// places in the image, and a table of 256 bytes says for every item where its value comes from -- itself (a literal) or the
// item the match points at.  Pointer jumping on that table (a match's source may be a match: at most eight rounds, one byte
// read and one byte written per item and round), then the values follow their pointers.  A stream that ends early, a
// distance that points in front of the block: 0 is returned with nothing decided, and lz_decode above does the block -- it
// is what reports the errors.
// The table lies behind the image, in the bytes the layout keeps readable there (make_dec_layout: 256 fit for both sizes).
WV_FN uint32_t lz_decode_256(Lds lds, const DecLayout& L, uint32_t T, uint32_t cur, uint32_t avail)
{
	const U32 lane = lane_id_plain();
	Lds win = lds + L.win;
	const uint32_t B = lz_width(T); // 4 or 8; 256 items, 32 groups
	const uint32_t from = L.img + 256 * T; // from[i]: the item whose value item i takes
	const uint32_t end = cur + avail;
	U32 ptr[4];
	Pred match[4];
	uint32_t p = cur;
	// First with every match taken for one byte; if one turns out to have two, once more with the groups sized exactly.
	U32 rec(0u); // lane g: where group g starts | its flags << 16 | which of its matches have two bytes << 24
	uint32_t first_long = 0; // the first group that holds a match of two bytes: the groups before it are where the first pass saw them
	for (uint32_t exact = 0;; ++exact) {
		if (!exact) {
			// every lane g < 32 walks to group g (lds_lz_walk32: the chain of 32 dependent byte reads, in vector registers throughout)
			U32 at(cur), fl(0u);
			lds_lz_walk32(win, at, fl, B);
			if (any((lane < U32(32u)) & (at + 2u > U32(end))))
				return 0;
			rec = at | (fl << 16);
			p = readlane(at, 31) + 1 + 8 * B - (B - 1) * (uint32_t)__builtin_popcount(readlane(fl, 31));
		}
		else {
			p = readlane(rec, first_long) & 0xFFFFu;
			for (uint32_t g = first_long; g < 32; ++g) {
				if (p + 2 > end)
					return 0;
				// the group's bytes behind the flags, one per lane (an item starts at most 56 bytes in): a match has two bytes when
				// its first one has bit 7 set -- eight steps on scalars, one v_readlane_b32 per match, a literal is one addition
				// (the same walk over a mask of the bytes with bit 7 set, without lane reads or branches, took 1.5 x as long: every
				// item then costs its eight scalar instructions, and the scalar unit is shared by the four SIMDs)
				const uint32_t flags = win_u8(win, p);
				const U32 bytes = lds_ld8(win, U32(p + 1) + lane);
				uint32_t two = 0, at = 0;
				for (uint32_t j = 0; j < 8; ++j) {
					if ((flags >> j) & 1u) {
						const uint32_t t = readlane(bytes, at) >> 7;
						two |= t << j;
						at += 1 + t;
					}
					else
						at += B;
				}
				rec = sel(lane == U32(g), U32(p | (flags << 16) | (two << 24)), rec);
				p += 1 + at;
			}
		}
		if (p > end)
			return 0;
		const U32 mine = shfl(rec, lane >> 1); // the group of the lane's four items (items 4 * lane ...)
		const U32 gp = mine & 0xFFFFu, gf = (mine >> 16) & 0xFFu, gt = mine >> 24;
		Pred bad = pred_all(false), longer = pred_all(false);
		for (uint32_t k = 0; k < 4; ++k) {
			const U32 j = ((lane & 1u) << 2) + k; // item of the group
			const U32 idx = (lane << 2) + k;     // item of the block
			const U32 below = (U32(1u) << j) - 1u;
			const U32 mb = popc(gf & below); // matches in front of it in the group
			const U32 off = gp + 1u + mb + popc(gt & below) + (j - mb) * B;
			match[k] = ((gf >> j) & 1u) != U32(0u);
			const Pred two = ((gt >> j) & 1u) != U32(0u);
			const U32 lo = lds_ld32_unaligned(win, off); // a literal, or a match's distance in its first byte or two
			const U32 d = sel(two, (lo & 127u) | ((lo >> 1) & 0x7F80u), lo & 0xFFu);
			longer = longer | (match[k] & !two & ((lo & 0x80u) != U32(0u)));
			bad = bad | (match[k] & ((d == U32(0u)) | (d > idx)));
			ptr[k] = sel(match[k], idx - d, idx);
			lds_st32(lds, U32(L.img) + idx * B, lo, !match[k]);
			if (B == 8)
				lds_st32(lds, U32(L.img + 4) + idx * 8u, lds_ld32_unaligned(win, off + 4u), !match[k]);
		}
		if (const uint64_t lb = ballot(longer)) {
			if (exact)
				return 0;
			first_long = (uint32_t)__builtin_ctzll(lb) >> 1; // (two lanes a group)
			continue;
		}
		if (any(bad))
			return 0;
		break;
	}
	{
		const U32 four = ptr[0] | (ptr[1] << 8) | (ptr[2] << 16) | (ptr[3] << 24); // (distances of matches never reach 256 items back here)
		lds_st32(lds, U32(from) + (lane << 2), four, pred_all(true));
	}
	wave_sync();
	// pointer jumping: an item that is its own source is a literal; a match takes over its source's source
	for (uint32_t round = 0; round < 8; ++round) {
		U32 next[4];
		for (uint32_t k = 0; k < 4; ++k)
			next[k] = lds_ld8(lds, U32(from) + ptr[k]);
		wave_sync();
		Pred moved = pred_all(false);
		for (uint32_t k = 0; k < 4; ++k) {
			moved = moved | (next[k] != ptr[k]);
			ptr[k] = next[k];
		}
		const U32 four = ptr[0] | (ptr[1] << 8) | (ptr[2] << 16) | (ptr[3] << 24);
		lds_st32(lds, U32(from) + (lane << 2), four, pred_all(true));
		wave_sync();
		if (!any(moved))
			break;
	}
	// (after eight rounds every chain of at most 255 links has ended at a literal)
	U32 vlo[4], vhi[4];
	for (uint32_t k = 0; k < 4; ++k) {
		const U32 a = U32(L.img) + ptr[k] * B;
		vlo[k] = lds_ld32(lds, a);
		vhi[k] = B == 8 ? lds_ld32(lds, a + 4u) : U32(0u);
	}
	wave_sync();
	for (uint32_t k = 0; k < 4; ++k) {
		const U32 a = U32(L.img) + ((lane << 2) + k) * B;
		lds_st32(lds, a, vlo[k], match[k]);
		if (B == 8)
			lds_st32(lds, a + 4u, vhi[k], match[k]);
	}
	wave_sync();
	return p - cur;
}

// A full block of bytesoftype 2, 4 or 8 made of planes (not COPY, not LZ), straight to HBM: the lane's word of every plane
// stays in a register, three (bytesoftype 2: one) v_perm_b32 per element turn them into the lane's four elements and one
// or two 16-byte stores write them.  No byte stores into an element-major image (stride 4*T between the lanes: LDS bank
// conflicts), no image to read back.  g: where the block goes (any byte address).  head: the type nibbles.
template <uint32_t T>
WV_FN uint32_t decode_planes_to(Lds lds, const DecLayout& L, uint32_t cur, uint32_t avail, uint32_t head, uint8_t* g)
{
	const U32 lane = lane_id_plain();
	Lds win = lds + L.win;
	uint32_t p = cur + header_bytes(T), bad = 0;
	const uint32_t end = cur + avail;
	U32 w[T];
#ifndef WV_HOST_EMULATION
#pragma unroll
#endif
	for (uint32_t j = 0; j < T; ++j) {
		const uint32_t type = (head >> (4 * j)) & 15;
		w[j] = U32(0u);
		if (type >= PLANE_NORMAL) {
			if (type <= PLANE_NORMAL_RLE)
				p += decode_plane(lds, L, T, j, type, p, end - p, 16, &w[j]);
			else
				bad = 1; // (:1854-1855)
		}
		else if (type == PLANE_SAME) { // (:1567-1583)
			w[j] = bytes_splat(lds_ld8(win, U32(p)));
			p += 1;
		}
		else { // RAW (:1553-1565)
			w[j] = lds_ld32_run(win, p, lane * 4u);
			p += 256;
		}
	}
	if (bad) // (two tests, each a scalar compare and branch: as one condition they are put together from lane masks)
		return DEC_ERROR;
	WV_NESTED();
	if (p > end)
		return DEC_ERROR;
	if (T == 2)
		gst64_unaligned(g, lane * 8u, perm_bytes(w[1], w[0], 0x05010400u), perm_bytes(w[1], w[0], 0x07030602u), pred_all(true));
	else {
		U32 e[T / 4 ? T / 4 : 1][4]; // e[h][i]: bytes 4h .. 4h+3 of the lane's element i
		for (uint32_t h = 0; h < T / 4; ++h) {
			const U32 a01 = perm_bytes(w[4 * h + 1], w[4 * h], 0x05010400u), a23 = perm_bytes(w[4 * h + 3], w[4 * h + 2], 0x05010400u); // elements 0 and 1
			const U32 b01 = perm_bytes(w[4 * h + 1], w[4 * h], 0x07030602u), b23 = perm_bytes(w[4 * h + 3], w[4 * h + 2], 0x07030602u); // elements 2 and 3
			e[h][0] = perm_bytes(a23, a01, 0x05040100u);
			e[h][1] = perm_bytes(a23, a01, 0x07060302u);
			e[h][2] = perm_bytes(b23, b01, 0x05040100u);
			e[h][3] = perm_bytes(b23, b01, 0x07060302u);
		}
		U128 v;
		if (T == 4) {
			v.x = e[0][0], v.y = e[0][1], v.z = e[0][2], v.w = e[0][3];
			gst128_unaligned(g, lane * 16u, v, pred_all(true));
		}
		else {
			v.x = e[0][0], v.y = e[T / 4 - 1][0], v.z = e[0][1], v.w = e[T / 4 - 1][1];
			gst128_unaligned(g, lane * 32u, v, pred_all(true));
			v.x = e[0][2], v.y = e[T / 4 - 1][2], v.z = e[0][3], v.w = e[T / 4 - 1][3];
			gst128_unaligned(g, lane * 32u + 16u, v, pred_all(true));
		}
	}
	return p - cur;
}

// Decode `lines` rows (16 = a full block) of the block whose encoding starts at window offset cur.
// full: a full block (may be COPY / LZ, planes may be RAW / NORMAL_RLE).
// g, direct (optional, together; full blocks of bytesoftype 2, 4, 8 known at compile time): the block's place in HBM; a
// block made of planes is written there from registers, a copied or mini-LZ block from the image.  Without them the
// block lands in the image.
// Returns bytes consumed or DEC_ERROR.
WV_FN void store_block(uint8_t* g, Lds lds, uint32_t ldsoff, uint32_t n); // (superblock_codec.h)
WV_FN uint32_t decode_block(Lds lds, const DecLayout& L, uint32_t T, uint32_t cur, uint32_t avail, uint32_t lines, bool full, uint8_t* g = nullptr,
			    bool* direct = nullptr)
{
	const U32 lane = lane_id_plain();
	Lds win = lds + L.win;
	const uint32_t hs = header_bytes(T);
	if (avail <= hs) // src += header_len; src >= end  (block_compress.h:1819-1821)
		return DEC_ERROR;
	// the first four bytes of the encoding in one uniform read: the marker byte / all plane type nibbles up to bytesoftype 8
	const uint32_t head = readlane(lds_ld32_unaligned(win, U32(cur)), 0);
	uint32_t first = head & 0xFFu;
	if (full && first >= BLOCK_COPY) { // (one test in front of the two markers: a block made of planes passes a single compare)
		WV_NESTED();
		if (first == BLOCK_COPY) { // (:1823-1828)
			if (avail < 1 + 256 * T)
				return DEC_ERROR;
			for (uint32_t o = 0; o < 256 * T; o += 256) {
				U32 v = lds_ld32_run(win, cur + 1 + o, lane * 4u);
				lds_st32(lds, U32(L.img + o) + lane * 4u, v, pred_all(true));
			}
			wave_sync();
			if (direct) { // (with a place in HBM every block leaves this function stored: the caller has nothing to ask)
				store_block(g, lds, L.img, 256 * T);
				wave_sync();
			}
			return 1 + 256 * T;
		}
		if (first == BLOCK_LZ) { // (:1829-1835)
			if (T % 4 != 0)
				return DEC_ERROR;
			uint32_t n = 0;
#ifndef STENOS_LZ_DECODE_SERIAL
			if (T == 4 || T == 8)
				n = lz_decode_256(lds, L, T, cur + 1, avail - 1);
#endif
			if (n == 0) {
#ifdef WV_HOST_EMULATION
				++emul_lz_serial_count();
#endif
				n = lz_decode(lds, L, T, cur + 1, avail - 1);
			}
			if (n == DEC_ERROR)
				return DEC_ERROR;
			if (direct) {
				store_block(g, lds, L.img, 256 * T);
				wave_sync();
			}
			return n + 1;
		}
	}
	if (direct && full) { // (direct is only passed with a place in HBM: its address is known at compile time, g's value is not)
		*direct = true;
		if (T == 2)
			return decode_planes_to<2>(lds, L, cur, avail, head, g);
		if (T == 4)
			return decode_planes_to<4>(lds, L, cur, avail, head, g);
		return decode_planes_to<8>(lds, L, cur, avail, head, g);
	}
	uint32_t p = cur + hs;
	const uint32_t end = cur + avail;
	// No bounds check per plane (the bytes a block can read past its start are bounded and readable, see decode_plane):
	// an invalid plane type is remembered, and the bytes consumed are compared with the bytes there are once, at the end
	// -- a truncated stream cannot satisfy that (block_compress.h:1560, 1575, 1591-1598, 1642).  Returns from inside the
	// loop would cost a dozen scalar instructions of exit bookkeeping per plane on every block's critical path.
	uint32_t bad = 0;
#ifndef WV_HOST_EMULATION
#pragma unroll 4
#endif
	for (uint32_t j = 0; j < T; ++j) {
		WV_MARK("dec_plane_type");
		uint32_t type = j < 8 ? (head >> (4 * j)) & 15 : (win_u8(win, cur + (j >> 1)) >> (4 * (j & 1))) & 15;
		if (type >= PLANE_NORMAL) { // two tests per plane whatever its type
			if (type == PLANE_NORMAL || (type == PLANE_NORMAL_RLE && full))
				p += decode_plane(lds, L, T, j, type, p, end - p, lines);
			else
				bad = 1; // (:1854-1855, 1779-1780)
		}
		else if (type == PLANE_SAME) { // (:1567-1583)
			U32 v = bytes_splat(lds_ld8(win, U32(p)));
			store_plane_word(lds, L.img, T, j, v, (lane >> 2) < U32(lines));
			p += 1;
		}
		else if (full) { // RAW (:1553-1565)
			store_plane_word(lds, L.img, T, j, lds_ld32_unaligned(win, U32(p) + lane * 4u), pred_all(true));
			p += 256;
		}
		else
			bad = 1;
	}
	wave_sync();
	return (bad || p > end) ? DEC_ERROR : p - cur;
}

} // namespace codec
