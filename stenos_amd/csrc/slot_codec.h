// slot_codec.h -- the block encoder for bytesoftype 2 and 4, one pass on row lanes.
//
// Planes whose 256 bytes are all equal (SAME) need no analysis and a block of 16-bit elements has only two planes, so
// the planes that do need it are written, in plane order, to up to four "slots" of 256 bytes in LDS; the slots may come
// from two consecutive blocks (a batch).  Lane 16*s + r then owns row r of slot s from the first load to the last
// store: it reads its 16 bytes once, measures them (block_compress.h:385-535), takes the row's decisions, learns
// where the plane goes in the image and writes header nibble, minimum and payload (block_compress.h:739-806) without
// handing anything to another lane through memory.  Cross-row quantities (previous byte, previous minimum, plane totals,
// offsets) move with DPP inside the 16-lane rows, which are exactly the slots.
//
// Everything is computed on biased bytes (x ^ 0x80): the reference's signed comparisons (_mm_min_epi8, :407-411) become
// unsigned ones, and "value - minimum" of a row needs no borrow handling between the packed bytes.
#pragma once
#include "block_codec.h"

namespace codec {

#include "shape_tables.inc" // SHAPE_LANES_T4 / _T2, SHAPE_INFO_T4 / _T2 (tools/gen_shape_tables.py)

// What lane 16*s + r has to know about a pass of shape act0 | act1 << 4 (fields: tools/gen_shape_tables.py).
WV_FN U32 shape_lane_entry(uint32_t T, uint32_t shape)
{
	const uint8_t* t = (const uint8_t*)(T == 2 ? SHAPE_LANES_T2 : SHAPE_LANES_T4);
	return gld32(t + shape * 256u, lane_id() * 4u, pred_all(true));
}

constexpr uint32_t SLOT2_BYTES = 256; // lane 16*s + r reads LDS bytes [16*lane, 16*lane + 16) of the slot area: no bank conflicts
WV_HD uint32_t slot2_area(const Layout& L) { return L.aux; }

// ---- element lanes: a block in registers -> its non-constant planes in slots slot, slot + 1, ... ---------------------
// Byte k of the OR over all elements of (element ^ first element) is non-zero exactly when plane k is not constant
// (block_compress.h:396, 406, 415-418).
WV_FN SameScan scan_same_fast(const RawBlock& b, uint32_t T)
{
	SameScan s;
	U32 x;
	if (T == 2) {
		const uint32_t e0 = readlane(b.e.x, 0) & 0xFFFFu;
		const U32 e(e0 * 0x00010001u);
		x = (b.e.x ^ e) | (b.e.y ^ e);
		x = x | (x >> 16);
		s.first = e0;
	}
	else {
		const uint32_t e0 = readlane(b.e.x, 0);
		const U32 e(e0);
		x = ((b.e.x ^ e) | (b.e.y ^ e)) | ((b.e.z ^ e) | (b.e.w ^ e));
		s.first = e0;
	}
	s.act = mask_bit<1>(ballot((x & 0xFFu) != U32(0u))) | mask_bit<2>(ballot((x & 0xFF00u) != U32(0u)));
	if (T == 4)
		s.act |= mask_bit<4>(ballot((x & 0xFF0000u) != U32(0u))) | mask_bit<8>(ballot((x & 0xFF000000u) != U32(0u)));
	s.nact = (uint32_t)__builtin_popcount(s.act);
	return s;
}

// step: distance between the slots of consecutive planes (1: side by side; 4: a group of four blocks keeps the slots of one
// plane of its blocks side by side, superblock_codec.h)
WV_FN void write_slots_fast(Lds lds, const Layout& L, const RawBlock& b, uint32_t T, uint32_t act, uint32_t slot, uint32_t step = 1)
{
	const U32 a = U32(slot2_area(L)) + lane_id() * 4u;
	if (T == 2) {
		if (act & 1u) {
			lds_st32(lds, a + slot * SLOT2_BYTES, perm_bytes(b.e.y, b.e.x, 0x06040200u), pred_all(true));
			slot += step;
		}
		if (act & 2u)
			lds_st32(lds, a + slot * SLOT2_BYTES, perm_bytes(b.e.y, b.e.x, 0x07050301u), pred_all(true));
		return;
	}
	// 4 x 4 byte transpose in two steps; a step is skipped when none of its planes is wanted
	if (act & 3u) {
		const U32 t0 = perm_bytes(b.e.y, b.e.x, 0x05010400u), t2 = perm_bytes(b.e.w, b.e.z, 0x05010400u); // bytes 0 and 1 of two elements each
		if (act & 1u) {
			lds_st32(lds, a + slot * SLOT2_BYTES, perm_bytes(t2, t0, 0x05040100u), pred_all(true));
			slot += step;
		}
		if (act & 2u) {
			lds_st32(lds, a + slot * SLOT2_BYTES, perm_bytes(t2, t0, 0x07060302u), pred_all(true));
			slot += step;
		}
	}
	if (act & 12u) {
		const U32 t1 = perm_bytes(b.e.y, b.e.x, 0x07030602u), t3 = perm_bytes(b.e.w, b.e.z, 0x07030602u); // bytes 2 and 3
		if (act & 4u) {
			lds_st32(lds, a + slot * SLOT2_BYTES, perm_bytes(t3, t1, 0x05040100u), pred_all(true));
			slot += step;
		}
		if (act & 8u)
			lds_st32(lds, a + slot * SLOT2_BYTES, perm_bytes(t3, t1, 0x07060302u), pred_all(true));
	}
}

// Distinct hash keys among the first 20 * ROUNDS values of a block of bytesoftype 4 that is still in registers (lanes
// 0..19 hold the first 80, value k of a lane in round k): the first rejection test of the mini-LZ (block_codec.h,
// lz_precheck_passes) wants the count over all 80; the count over the first 40 is a lower bound of it that is enough to
// turn most blocks away.  hash_val (lz_compress.h:47-56) keeps the low byte of value * 2654435761, which only depends on
// the value's low byte.  Every value writes a tag of its own into table[key]; after all writes each key holds exactly one
// tag, so the values that find their own tag back are as many as there are distinct keys.  No zeroing, no atomics.
template <int ROUNDS>
WV_FN uint32_t lz_distinct_keys_fast(Lds lds, const Layout& L, const U128& e)
{
	uint32_t distinct = 0;
	lanes_below(lz_precheck_values(4) / 4, [&](const Pred& in) {
		const U32 lane = lane_id();
		const U32 v[4] = { e.x, e.y, e.z, e.w };
		U32 addr[4];
		for (int k = 0; k < ROUNDS; ++k) {
			addr[k] = U32(L.tab) + (mul24(v[k], U32(0xB1u * 4u)) & 0x3FCu);
			lds_st32(lds, addr[k], lane + U32(64u * (uint32_t)k), in);
		}
		wave_sync();
		for (int k = 0; k < ROUNDS; ++k)
			distinct += (uint32_t)__builtin_popcountll(ballot(in & (lds_ld32(lds, addr[k]) == lane + U32(64u * (uint32_t)k))));
	});
	wave_sync();
	return first_lane_value(distinct); // the other lanes have not counted
}

// Second rejection test of the mini-LZ for a block of bytesoftype 4 in registers (block_codec.h, lz_try): a value can only
// match when an equal value precedes it, and equal values have equal hashes.  One bit per 13-bit hash value (1 KiB at
// L.tab) counts the values whose hash has been seen before; if even with all of them matching the stream exceeds
// max_size, the reference fails after doing all the work (lz_compress.h:221-223).  True: the attempt is pointless.
WV_FN bool lz_repeats_reject(Lds lds, const Layout& L, const U128& e, uint32_t max_size)
{
	const U32 lane = lane_id();
	U128 z;
	z.x = z.y = z.z = z.w = U32(0u);
	lds_st128(lds, U32(L.tab) + lane * 16u, z, pred_all(true));
	wave_sync();
	const U32 v[4] = { e.x, e.y, e.z, e.w };
	uint32_t maybe = 0;
	for (int k = 0; k < 4; ++k) {
		const U32 h = (v[k] * 0x85EBCA6Bu) >> 19;
		const U32 bit = U32(1u) << (h & 31u);
		const U32 old = lds_or_rtn32(lds, U32(L.tab) + (h >> 5) * 4u, bit);
		maybe += (uint32_t)__builtin_popcountll(ballot((old & bit) != U32(0u)));
	}
	wave_sync();
	return 256 / 8 + 256 * 4 - maybe * 3 > max_size;
}

// ---- row lanes ---------------------------------------------------------------------------------------------------------

// per-byte a - p (mod 256) of biased operands, biased result: (a ^ H) - (p ^ H) = a - p, and bias the difference again
WV_FN U32 biased_sub(const U32& a, const U32& p)
{
	const U32 H(0x80808080u);
	return ((a | H) - (p & ~H)) ^ ((a ^ p) & H);
}
// (hi << 8) | (lo >> 24): the dword that starts one byte before hi
WV_FN U32 prev_bytes(const U32& hi, const U32& lo) { return (hi << 8) | (lo >> 24); }

// number of non-zero bytes among the 16 bytes x[0..3]: v_msad_u8 adds |a - ref| over the bytes whose ref is not zero, and
// a = ref ^ 1 differs from ref by exactly one in every byte -- two instructions per dword (the flags-and-dot-product form
// took four)
WV_FN U32 count_nonzero16(const U32* x)
{
	U32 acc(0u);
	for (int k = 0; k < 4; ++k)
		acc = msad_u8(x[k] ^ 0x01010101u, x[k], acc);
	return acc;
}
// Smallest and largest of the 16 bytes s[0..3] (unsigned), in two halves.  A 16-bit minimum has the smallest high byte, so
// the odd bytes are compared where they are (range_odd: both results still packed, [upper half | lower half] of 16-bit
// values whose high bytes count) and the even bytes after a shift by 8 (whatever follows them in the low byte does not matter).
WV_FN void range_odd(const U32* s, U32& lo, U32& hi)
{
	lo = pk_min_u16(pk_min_u16(s[0], s[1]), pk_min_u16(s[2], s[3]));
	hi = pk_max_u16(pk_max_u16(s[0], s[1]), pk_max_u16(s[2], s[3]));
}
WV_FN void range16(const U32* s, const U32& lo_odd, const U32& hi_odd, U32& mn, U32& mx)
{
	U32 e[4];
	for (int k = 0; k < 4; ++k)
		e[k] = s[k] << 8;
	const U32 lo = pk_min_u16(lo_odd, pk_min_u16(pk_min_u16(e[0], e[1]), pk_min_u16(e[2], e[3])));
	const U32 hi = pk_max_u16(hi_odd, pk_max_u16(pk_max_u16(e[0], e[1]), pk_max_u16(e[2], e[3])));
	mn = umin(lo >> 16, lo & 0xFFFFu) >> 8;
	mx = umax(hi >> 16, hi & 0xFFFFu) >> 8;
}
// the packed results of range_odd span at least 0x4100 as 16-bit values: the largest and the smallest of the bytes they were
// taken from -- and so of the whole row -- are at least 64 apart
WV_FN Pred odd_range_is_wide(const U32& lo, const U32& hi) { return umax(hi >> 16, hi & 0xFFFFu) - umin(lo >> 16, lo & 0xFFFFu) >= U32(0x4100u); }
// bits needed for v in 0..255 (0 for 0)
WV_FN U32 bitlen8(const U32& v) { return bitlen((v << 1) | 1u) - 1u; }

// What lane 16*s + r knows about row r of slot s once the slots are analysed.
struct SlotRows {
	U32 sb[4];  // the row's 16 bytes, biased
	U32 sd[4];  // their deltas against the previous byte in plane order (0 before the plane, block_compress.h:399-401), biased
	U32 hm;     // row header nibble (:497-503) | the row's minimum (of the bytes or of the deltas, whichever the row codes), biased, << 8
	U32 pm;     // offset of the row's payload in the plane (low half) and of its minimum, when it has one (high half): one register
	            // between the analysis and the emission, where they are scarce
	Pred emitmin, eq; // the row writes a minimum; its minimum equals that of the row above (bit of the mins-rle mask)
	U32 ts;           // size | type << 16 of the slot's plane (the same in its 16 lanes)
	// (pairs in one register each: between the analysis and the emission the row's 16 bytes, their deltas and the placement are
	// all alive, and what does not fit goes to scratch memory)
	WV_MFN U32 hdr() const { return hm & 0xFFu; }
	WV_MFN U32 minb() const { return hm >> 8; }
	WV_MFN U32 type() const { return ts >> 16; }
	WV_MFN U32 size() const { return ts & 0xFFFFu; }
};

// Analyse the four slots as full-block planes (rle enabled, raw above 256 bytes: block_compress.h:1110-1111, 1190, 1200-1204).
// slot_off: byte offset of the pass's four slots inside the slot area (a multiple of 1024; passes of a group of blocks keep
// their slots side by side).  nvalid: slots in use (the lanes of the others hold stale bytes).
//
// Planes of noise are proven RAW before the row decisions are taken.  A row whose bytes span at least 64 and whose
// differences span at least 64 needs 8 bits either way (:336-339, 422): 16 bytes and header 15, unless a run-length form is
// strictly shorter (:464-472), which takes three repeated values or three repeated differences in the row.  When no row of a
// plane gets below 16 bytes, no row writes a minimum (:480-490) and the plane measures 8 + 256 > 256: RAW (:1200-1204).  The
// spans are taken over the odd bytes only -- half of the range computation, which the full analysis continues from when
// some plane of the pass is not noise -- and that is enough: eight random bytes span less than 64 once in two thousand rows.
// Returns true when every slot in use was proven RAW: R then holds the rows' bytes, size and type, and nothing else.
WV_FN bool slot_rows_analyse(Lds lds, const Layout& L, SlotRows& R, uint32_t slot_off = 0, uint32_t nvalid = 4, bool try_raw = true)
{
	const U32 lane = lane_id();
	const U32 H(0x80808080u);
	WV_MARK("analyse_stage1");
	{
		const U128 v = lds_ld128(lds, U32(slot2_area(L) + slot_off) + lane * 16u);
		R.sb[0] = v.x ^ H;
		R.sb[1] = v.y ^ H;
		R.sb[2] = v.z ^ H;
		R.sb[3] = v.w ^ H;
	}
	// deltas: the byte before a row is the last byte of the row above, before the plane 0
	const U32 above = row_shr(R.sb[3], 1, 0x80808080u);
	U32 x[4]; // byte == previous byte  <=>  byte of x is zero (:268-275)
	{
		const U32 p0 = prev_bytes(R.sb[0], above);
		x[0] = R.sb[0] ^ p0;
		R.sd[0] = biased_sub(R.sb[0], p0);
		for (int k = 1; k < 4; ++k) {
			const U32 p = prev_bytes(R.sb[k], R.sb[k - 1]);
			x[k] = R.sb[k] ^ p;
			R.sd[k] = biased_sub(R.sb[k], p);
		}
	}
	const U32 c1 = count_nonzero16(x) + 2u; // rle cost: 2 + 16 - popcnt(mask)  (:464-467)
	// delta == previous delta, the delta before a row's first column being 0 (:248-255, 449-458); the bias cancels
	x[0] = R.sd[0] ^ ((R.sd[0] << 8) | 0x80u);
	for (int k = 1; k < 4; ++k)
		x[k] = R.sd[k] ^ prev_bytes(R.sd[k], R.sd[k - 1]);
	const U32 c2 = count_nonzero16(x) + 2u; // (:470-472)
	U32 lo, hi, dlo, dhi;
	range_odd(R.sb, lo, hi);
	range_odd(R.sd, dlo, dhi);
	if (try_raw) {
		const Pred noise = odd_range_is_wide(lo, hi) & odd_range_is_wide(dlo, dhi) & (umin(c1, c2) >= U32(16u));
		if (ballot(noise | (lane >= U32(16u * nvalid))) == ~0ull) {
			R.ts = U32(256u | (PLANE_RAW << 16));
			R.hm = R.pm = U32(0u);
			R.emitmin = R.eq = pred_all(false);
			WV_MARK("analyse_raw");
			return true;
		}
	}
	U32 mn, mx, dmn, dmx;
	range16(R.sb, lo, hi, mn, mx);
	range16(R.sd, dlo, dhi, dmn, dmx);

	WV_MARK("analyse_stage2");
	U32 b0 = bitlen8(mx - mn), b1 = bitlen8(dmx - dmn);
	b0 = sel(b0 >= U32(6u), U32(8u), b0); // 7 -> 8 (:336-339), header 6 is reserved for delta-rle (:422)
	b1 = sel(b1 >= U32(7u), U32(8u), b1);
	const U32 bits = umin(b0, b1);
	const Pred type0 = b0 == bits; // ties go to frame-of-reference (:423-427)
	const U32 minb = sel(type0, mn, dmn);
	U32 cost = bits * 2u + 1u - (bits >> 3); // (:433-435)
	U32 hdr = sel(type0, b0 + (b0 >> 3) * 7u, b1 + 8u); // (:497-503)
	const Pred u1 = c1 < cost; // strictly smaller wins
	cost = sel(u1, c1, cost);
	hdr = sel(u1, U32(7u), hdr);
	const Pred u2 = c2 < cost;
	cost = sel(u2, c2, cost);
	hdr = sel(u2, U32(6u), hdr);
	R.hm = hdr | (minb << 8);
	const Pred nomin = (hdr == U32(15u)) | ((hdr & 14u) == U32(6u));
	R.eq = minb == row_shr(minb, 1, 0x80u); // the minimum before row 0 counts as 0 (:483)
	const U32 tot = row_add(cost | sel(nomin, U32(1u << 12), U32(0u)) | sel(R.eq, U32(1u << 17), U32(0u)));
	const U32 sumcost = tot & 0xFFFu, count8 = (tot >> 12) & 31u, eqc = (tot >> 17) & 31u;
	// mins rle (:478-490): 2 + non-repeated mins < mins that would be written
	const U32 plain = U32(16u) - count8, packed = U32(18u) - eqc;
	const Pred minsrle = packed < plain;
	const U32 minslen = umin(packed, plain);
	U32 size = sumcost + 8u + minslen - plain; // (:476, 488)
	const Pred raw = size > U32(256u);
	R.ts = sel(raw, U32(256u | (PLANE_RAW << 16)), size | sel(minsrle, U32(PLANE_NORMAL_RLE << 16), U32(PLANE_NORMAL << 16)));
	const U32 pay = cost - sel(nomin, U32(0u), U32(1u));
	R.emitmin = (minsrle & !R.eq) | (!minsrle & !nomin);
	const U32 ex = row_excl_scan(pay | sel(R.emitmin, U32(1u << 16), U32(0u)));
	R.pm = ex + (minslen + 8u + sel(minsrle, U32(10u << 16), U32(8u << 16))); // (both halves stay far below 2^16)
	return false;
}

// A batch: one block or two consecutive ones whose non-constant planes fill slots 0 .. nslots-1 (block 0 first).
struct SlotBatch {
	uint32_t act[2];   // bit k: plane k of the block is not constant
	uint32_t first[2]; // first element of each block (the bytes of its SAME planes)
	uint32_t nact0;    // slots of block 0
	uint32_t nslots;   // slots of both
	uint32_t nblk;
	uint32_t full[2];  // sum of the plane sizes of each block (block_compress.h:1189-1207)
};

// Where the planes go: P.pbase = offset of the slot's plane in its block's encoding; P.e: the lane's entry of the shape table
// (plane number, block, SAME planes around the plane); P.first: first element of the slot's block.
struct SlotPlace {
	U32 pbase, e, first;
	Pred valid, second;
};
// The sizes of the batch's blocks (B.full) and, for the emission, where every slot's plane goes.  incl: inclusive sums of the
// plane sizes over the slots (the value of a slot is the same in its 16 lanes).
WV_FN U32 slot_rows_sizes(const SlotRows& R, SlotBatch& B, uint32_t T, uint32_t* p0_out)
{
	U32 incl = R.size() + scan_source(R.size(), 4, 0u);
	incl = incl + scan_source(incl, 5, 0u);
	const uint32_t p0 = B.nact0 ? readlane(incl, 16u * B.nact0 - 1u) : 0u;
	const uint32_t pt = B.nslots ? readlane(incl, 16u * B.nslots - 1u) : 0u;
	B.full[0] = p0 + (T - B.nact0);
	B.full[1] = B.nblk > 1 ? pt - p0 + (T - (B.nslots - B.nact0)) : 0u;
	*p0_out = p0;
	return incl;
}
// e: shape_lane_entry(T, B.act[0] | B.act[1] << 4); incl, p0: from slot_rows_sizes
WV_FN SlotPlace slot_rows_place(const SlotRows& R, const SlotBatch& B, uint32_t T, const U32& e, const U32& incl, uint32_t p0)
{
	SlotPlace P;
	P.e = e;
	P.valid = (e & 8u) != U32(0u);
	P.second = (e & 4u) != U32(0u);
	P.first = sel(P.second, U32(B.first[1]), U32(B.first[0]));
	// planes before mine in my block: SAME ones take a byte each (k - j of them), the others are the slots before mine
	P.pbase = U32(header_bytes(T)) + ((e >> 4) & 3u) + (incl - R.size() - sel(P.second, U32(p0), U32(0u)));
	return P;
}

// OR a value that cannot straddle a dword (a nibble at a nibble-aligned bit position, a byte at a byte-aligned one) into the
// zeroed image.  Lanes that have nothing to write OR what they computed into a dword of their own OUTSIDE the image (`dump`,
// a byte offset from the image: the slot area, whose rows are in registers by then) -- one shared address would serialise
// the wave in the LDS atomic unit, and a place inside the image would have the value masked to zero first.
WV_FN void put_small(Lds out, const U32& bitpos, const U32& value, const Pred& p, const U32& dump)
{
	lds_or32_all(out, sel(p, (bitpos >> 3) & ~3u, dump), value << (bitpos & 31u));
}
// the same for up to 32 bits at any bit position
WV_FN void put_bits(Lds out, const U32& bitpos, const U32& value, const Pred& p, const U32& dump)
{
	const U32 a = sel(p, (bitpos >> 3) & ~3u, dump), sh = bitpos & 31u;
	lds_or32_all(out, a, value << sh);
	lds_or32_all(out, a + 4u, sel(sh == U32(0u), U32(0u), value >> (U32(32u) - sh)));
}

// four values of at most `bits` bits (0..8), one per byte of x -> 4*bits bits; the even and the odd bytes are pulled apart
// with one v_perm_b32 each (0x0c selects a zero byte)
WV_FN U32 pack4v(const U32& x, const U32& bits)
{
	const U32 t = perm_bytes(x, x, 0x0c020c00u) | (perm_bytes(x, x, 0x0c030c01u) << bits);
	return (t & 0xFFFFu) | ((t >> 16) << (bits + bits));
}

// The payload of the lane's row (block_compress.h:562-602, 649-664): bit-packed rows as two halves of 8 values, `bits` bytes
// each, value - minimum on biased bytes; raw rows -- the rows of RAW planes and rows with header 15 -- are the same thing
// with 8 bits, no minimum and the bytes as they came.  Lanes that write nothing (run-length rows, rows of 0 bits, unused
// slots) OR whatever they computed into a dump behind the image (the slot area: its rows are in registers by now), so
// nothing has to be masked to zero for them.
WV_FN void emit_row_payload(Lds out, const Layout& L, const SlotRows& R, const Pred& rawrow, const Pred& packed, const U32& bits, const U32& rbase, uint32_t scr = 0)
{
	const U32 lane = lane_id();
	const U32 H(0x80808080u);
	const Pred usedelta = R.hdr() >= U32(8u);
	const U32 mins = splat_byte0(R.minb());
	const U32 ebits = sel(rawrow, U32(8u), bits);
	U32 pk[4];
	for (int k = 0; k < 4; ++k)
		pk[k] = pack4v(sel(rawrow, R.sb[k] ^ H, sel(usedelta, R.sd[k], R.sb[k]) - mins), ebits);
	const U32 sh4 = ebits << 2; // (32 for raw rows: the second dword of a half is pk[1] as it is)
	U32 s0lo, s0hi, s1lo, s1hi;
	shl64(pk[1], sh4, s0lo, s0hi);
	shl64(pk[3], sh4, s1lo, s1hi);
	const U32 dump = U32(slot2_area(L) + scr - L.out) + lane * 16u;
	const Pred anyw = rawrow | packed;
	lds_put_bytes8(out, sel(anyw, rbase, dump), s0lo | pk[0], s0hi);
	lds_put_bytes8(out, sel(anyw, rbase + ebits, dump), s1lo | pk[2], s1hi);
}

// The 16 v_perm_b32 selectors that move the bytes of a dword whose flag is 0 to its low end (entry f: flags f; lane f writes
// it).  Once per run of blocks, into a place of its own.
WV_FN void slot_write_rle_lut(Lds lds, const Layout& L)
{
	const U32 lane = lane_id();
	U32 pat(0x0c0c0c0cu), at(0u);
	for (uint32_t k = 0; k < 4; ++k) {
		const Pred keep = ((lane >> k) & 1u) == U32(0u);
		pat = sel(keep, (pat & ~(U32(0xFFu) << at)) | (U32(k) << at), pat);
		at = at + sel(keep, U32(8u), U32(0u));
	}
	lds_st32(lds, U32(L.rlelut) + lane * 4u, pat, lane < U32(16u));
	wave_sync();
}

// The rows of the slots into the zeroed image, once every slot knows where its plane starts (pbase, from the image's start):
// header nibbles, minima, the mask of repeated minima, payloads (block_compress.h:739-806).  Shared by the three batch forms.
// scr: which KiB of the slot area serves as scratch -- the dump of the lanes with nothing to write, the hand-over of the
// run-length rows -- (0: the pass's own slots, whose rows are in registers; a group of four blocks names the slots of its
// second pass, the first one's being read again later)
// rowwise: the form for passes made of run-length rows is compiled in (known where the call is compiled: the groups of four
// of 32-bit elements, the headline's path, go without -- their copies of this code would cost the int32 kernel registers)
WV_FN void slot_rows_emit_rows(Lds lds, const Layout& L, const SlotRows& R, const Pred& valid, const U32& pbase, uint32_t scr = 0, bool rowwise = true)
{
	const U32 lane = lane_id();
	const U32 r = lane & 15u;
	const U32 H(0x80808080u);
	Lds out = lds + L.out;
	const U32 own = U32(slot2_area(L) + scr - L.out) + lane * 16u; // where lanes with nothing to write OR what they have (outside the image)
	WV_MARK("emit_rowlanes");
	const U32 hdr = R.hdr();
	const Pred israw = valid & (R.type() == U32(PLANE_RAW));
	const Pred normal = valid & !israw;
	if (!any(normal)) {
		// Planes of noise only (the low bytes of doubles: four such planes fill the first pass of most blocks): their rows as they
		// came (:1553-1565), sixteen bytes per lane -- no headers, no minima, nothing to pack.
		const U32 to = pbase + r * 16u;
		lds_put_bytes8(out, sel(valid, to, own), R.sb[0] ^ H, R.sb[1] ^ H);
		lds_put_bytes8(out, sel(valid, to + 8u, own), R.sb[2] ^ H, R.sb[3] ^ H);
		WV_MARK("emit_end_raw");
		wave_sync();
		return;
	}
	put_small(out, pbase * 8u + r * 4u, hdr, normal, own); // (:768-779, 758-762)
	put_small(out, (pbase + (R.pm >> 16)) * 8u, R.minb() ^ 0x80u, normal & R.emitmin, own);
	{
		// mins rle mask (:765): bit r = min equals previous min
		const Pred isnrle = normal & (R.type() == U32(PLANE_NORMAL_RLE));
		if (any(isnrle)) {
			const U32 m16 = row_ballot16(R.eq);
			put_bits(out, (pbase + 8u) * 8u, m16, isnrle & (r == U32(0u)), own);
		}
	}
	WV_MARK("emit_plane");
	const Pred is15 = hdr == U32(15u), isr = (hdr & 14u) == U32(6u);
	const U32 bits = hdr & 7u;
	const Pred rawrow = israw | (normal & is15);
	const Pred packed = normal & !is15 & !isr & (bits != U32(0u));
	const Pred rle = normal & isr;
	const U32 rbase = pbase + sel(israw, r * 16u, R.pm & 0xFFFFu);
	emit_row_payload(out, L, R, rawrow, packed, bits, rbase, scr);
	// rle / delta-rle rows (:258-265, 285-293): [mask16][literals]
	if (any(rle)) {
		// Few rows of a pass are run-length rows (one or two per block of smooth floats), but a loop over the row's four
		// dwords by all 64 lanes costs the same for one such row as for 64.  So the rows are handed to quads of lanes: row
		// number i of the pass (in lane order) goes to lanes 4i .. 4i+3, lane k takes the row's dword k -- flags, literals
		// and where they go -- and one 8-byte OR per lane writes mask and literals.  The hand-over goes through the slot
		// area (free by now: the rows are in registers), sixteen rows at a time.
		const Pred is7 = hdr == U32(7u);
		const uint32_t lut = L.rlelut; // (slot_write_rle_lut, once per run)
		const uint64_t rows = ballot(rle);
		const uint32_t n = (uint32_t)__builtin_popcountll(rows);
		const U32 rank = lane_rank(rows);
		const uint32_t area = slot2_area(L) + scr; // entry i, 32 bytes: [payload offset][-][-][byte in front][the row's 16 bytes]
		U128 d;
		d.x = sel(is7, R.sb[0], R.sd[0]), d.y = sel(is7, R.sb[1], R.sd[1]), d.z = sel(is7, R.sb[2], R.sd[2]), d.w = sel(is7, R.sb[3], R.sd[3]);
		// in front of a row of values: the last byte of the row above (:268-275); of a row of differences: no difference (:248-255)
		const U32 front = sel(is7, row_shr(R.sb[3], 1, 0x80808080u), U32(0x80808080u));
		if (rowwise && n > 32) {
			// Most rows of the pass are run-length rows (long runs, steps: whole frames are made of such planes): then every lane
			// does its own row, four dwords one after the other -- 90 instructions whatever the number of rows, where the hand-over
			// below takes 40 per sixteen rows.
			U32 f16(0u), lp = rbase + 2u;
			for (int j = 0; j < 4; ++j) {
				// byte == previous byte (:268-275) / difference == previous difference (:248-255)
				const U32 bp = j ? prev_bytes(R.sb[j], R.sb[j - 1]) : prev_bytes(R.sb[0], row_shr(R.sb[3], 1, 0x80808080u));
				const U32 dp = j ? prev_bytes(R.sd[j], R.sd[j - 1]) : ((R.sd[0] << 8) | 0x80u);
				const U32 f = zero_mask_to_bits(bytes_zero_mask(sel(is7, R.sb[j] ^ bp, R.sd[j] ^ dp)));
				f16 = f16 | (f << U32(4u * (uint32_t)j));
				const U32 nlit = U32(4u) - popc(f);
				const U32 lits = perm_bytes_v(U32(0u), sel(is7, R.sb[j], R.sd[j]) ^ H, lds_ld32(lds, U32(lut) + f * 4u));
				put_bits(out, lp * 8u, lits, rle & (nlit != U32(0u)), own);
				lp = lp + nlit;
			}
			put_bits(out, rbase * 8u, f16, rle, own);
			WV_MARK("emit_end");
			wave_sync();
			return;
		}
		const U32 k = lane & 3u, k4 = k << 2, quad = lane >> 2;
		const U32 idle = U32(area - L.out + 512u) + (lane & 31u) * 16u; // where quads without a row OR what they have
		for (uint32_t c = 0; c < n; c += 16) {
			const U32 slotno = rank - U32(c);
			const Pred mine = rle & (slotno < U32(16u));
			const U32 ent = U32(area) + (slotno & 15u) * 32u;
			lds_st32(lds, ent, rbase, mine);
			lds_st32(lds, ent + 12u, front, mine);
			lds_st128(lds, ent + 16u, d, mine);
			wave_sync();
			const U32 qe = U32(area) + quad * 32u;
			U32 lo, hi;
			lds_ld64(lds, qe + 12u + k4, lo, hi);
			const U32 rb = lds_ld32(lds, qe);
			// byte == byte in front of it
			const U32 f = zero_mask_to_bits(bytes_zero_mask(hi ^ prev_bytes(hi, lo)));
			const U32 lits = perm_bytes_v(U32(0u), hi ^ H, lds_ld32(lds, U32(lut) + f * 4u));
			U32 f16 = f << k4;
			f16 = f16 | shfl_xor(f16, 1);
			f16 = f16 | shfl_xor(f16, 2);
			const U32 before = popc(~f16 & ((U32(1u) << k4) - 1u)); // literals of the row in front of this lane's
			const Pred first = k == U32(0u);
			const U32 pos = rb + sel(first, U32(0u), before + 2u);
			const Pred have = (U32(c) + quad) < U32(n);
			lds_put_bytes8(out, sel(have, pos, idle), sel(first, f16 | (lits << 16), lits), sel(first, lits >> 16, U32(0u)));
			wave_sync();
		}
	}
	WV_MARK("emit_end");
	wave_sync();
}

// Write the blocks of the batch into the zeroed image, block i at byte base[i]: type nibbles, SAME bytes, row headers,
// minima and payloads (block_compress.h:739-806, 1246-1257).
WV_FN void slot_rows_emit(Lds lds, const Layout& L, uint32_t T, const SlotRows& R, const SlotPlace& P, const SlotBatch& B, uint32_t base0, uint32_t base1)
{
	const U32 lane = lane_id();
	Lds out = lds + L.out;
	const uint32_t hs = header_bytes(T);
	const U32 own = U32(slot2_area(L) - L.out) + lane * 16u; // where lanes with nothing to write OR what they have (outside the image)
	WV_MARK("emit_nibbles");
	const U32 bbase = sel(P.second, U32(base1), U32(base0));
	const U32 pbase = bbase + P.pbase;
	{
		// One small write per lane of a slot, as the shape table says: lane 0 the plane's type nibble (:1246-1257); lanes 1..3
		// the bytes of the SAME planes that follow the plane directly; lanes 4..7 of a block's first slot the SAME planes in
		// front of it (:747-750).  Bit position: the table's constant from the block's start or from the end of the plane.
		const U32 role = (P.e >> 6) & 3u;
		const U32 byte = (P.first >> ((P.e >> 8) & 31u)) & 0xFFu;
		const U32 from = sel(role == U32(2u), pbase + R.size(), bbase);
		const Pred nib = role == U32(1u);
		put_small(out, from * 8u + ((P.e >> 13) & 63u), sel(nib, R.type(), byte), role != U32(0u), own);
	}
	for (uint32_t i = 0; i < B.nblk; ++i) // a block without a slot: its SAME bytes (the type nibbles are all 0)
		if (B.act[i] == 0) {
			const U32 byte = (U32(B.first[i]) >> (lane << 3)) & 0xFFu;
			put_small(out, (U32((i ? base1 : base0) + hs) + lane) * 8u, byte, lane < U32(T), own);
		}
	if (B.nslots == 0) {
		wave_sync();
		return;
	}
	slot_rows_emit_rows(lds, L, R, P.valid, pbase);
}

// A wide batch: three or four consecutive blocks with at most one non-constant plane each (small integers in wide
// elements, one byte that varies): the slots two such blocks leave free take the blocks behind them.  No mini-LZ in such
// a batch (a block with one plane is below its size threshold, block_compress.h:1210), and a plane is the only one of its
// block.  (Letting blocks with two planes into such batches as well was measured: the general placement costs the
// common two-block pass 3 % in registers and gains nothing.)
struct SlotBatch4 {
	uint32_t act[4], first[4]; // per block; constant indices only (a loop over the blocks would send them through memory)
	uint32_t nblk, nslots;
	uint32_t full[4];          // sum of the plane sizes of each block
};
WV_HD uint32_t pick4(uint32_t i, uint32_t a, uint32_t b, uint32_t c, uint32_t d) { return i == 0 ? a : (i == 1 ? b : (i == 2 ? c : d)); }
// base[q] receives the image offset of block q (the first one at img_base), *bbase that of its slot's block per lane
struct SlotPlace4 {
	U32 pbase, k, blk, act, first;
	Pred valid;
};
WV_FN SlotPlace4 slot_rows_place4(const SlotRows& R, SlotBatch4& B, uint32_t T, const uint32_t hs, uint32_t img_base, uint32_t* base, U32* bbase)
{
	SlotPlace4 P;
	const U32 lane = lane_id();
	const U32 s = lane >> 4;
	// the slots are the set bits of act[0] | act[1] << 4 | act[2] << 8 | act[3] << 12 in order
	uint32_t m = B.act[0] | (B.act[1] << 4) | (B.act[2] << 8) | (B.act[3] << 12) | 0xF0000u, pos[4];
	for (int i = 0; i < 4; ++i) {
		pos[i] = (uint32_t)__builtin_ctz(m);
		m &= m - 1u;
	}
	const U32 posv = row_select4(pos[0], pos[1], pos[2], pos[3]);
	P.k = posv & 3u;
	P.blk = posv >> 2;
	P.valid = s < U32(B.nslots);
	// every block has at most one slot: its plane's size is the block's plane sum
	const uint32_t z0 = readlane(R.size(), 0), z1 = readlane(R.size(), 16), z2 = readlane(R.size(), 32), z3 = readlane(R.size(), 48);
	const uint32_t n0 = B.act[0] ? 1u : 0u, n1 = B.act[1] ? 1u : 0u, n2 = B.act[2] ? 1u : 0u, n3 = B.act[3] ? 1u : 0u;
	const uint32_t s1 = n0, s2 = n0 + n1, s3 = n0 + n1 + n2; // the slot blocks 1, 2, 3 would take
	B.full[0] = (n0 ? z0 : 0u) + (T - n0);
	B.full[1] = (n1 ? pick4(s1, z0, z1, z2, z3) : 0u) + (T - n1);
	B.full[2] = B.nblk > 2 ? (n2 ? pick4(s2, z0, z1, z2, z3) : 0u) + (T - n2) : 0u;
	B.full[3] = B.nblk > 3 ? (n3 ? pick4(s3, z0, z1, z2, z3) : 0u) + (T - n3) : 0u;
	base[0] = img_base;
	base[1] = base[0] + hs + B.full[0];
	base[2] = base[1] + hs + B.full[1];
	base[3] = base[2] + hs + B.full[2];
	const uint32_t b0 = pos[0] >> 2, b1 = pos[1] >> 2, b2 = pos[2] >> 2, b3 = pos[3] >> 2; // block of each slot
#define STENOS_PER_SLOT(a, b, c, d) row_select4(pick4(b0, a, b, c, d), pick4(b1, a, b, c, d), pick4(b2, a, b, c, d), pick4(b3, a, b, c, d))
	P.act = STENOS_PER_SLOT(B.act[0], B.act[1], B.act[2], B.act[3]);
	P.first = STENOS_PER_SLOT(B.first[0], B.first[1], B.first[2], B.first[3]);
	*bbase = STENOS_PER_SLOT(base[0], base[1], base[2], base[3]);
#undef STENOS_PER_SLOT
	// the planes in front of mine in my block are SAME ones: a byte each
	P.pbase = U32(hs) + P.k;
	return P;
}

// The same for a wide batch: block q of the batch starts at byte base[q] of the image, bbase is that of the lane's slot.
WV_FN void slot_rows_emit4(Lds lds, const Layout& L, uint32_t T, const SlotRows& R, const SlotPlace4& P, const SlotBatch4& B, const U32& bbase, const uint32_t* base)
{
	const U32 lane = lane_id();
	const U32 r = lane & 15u;
	const U32 H(0x80808080u);
	Lds out = lds + L.out;
	const uint32_t hs = header_bytes(T);
	const U32 own = U32(slot2_area(L) - L.out) + lane * 16u; // where lanes with nothing to write OR what they have (outside the image)
	WV_MARK("emit_nibbles");
	const U32 pbase = bbase + P.pbase;
	{
		// One small write per lane of a slot: lane 0 the plane's type nibble (:1246-1257); lanes 1..3 the bytes of the SAME
		// planes that follow the plane directly; lanes 4..7 of a block's first slot the SAME planes in front of it (:747-750).
		const U32 after = r - 1u, before = r - 4u;
		const Pred follows = (r >= U32(1u)) & (r < U32(4u)) & (P.k + r < U32(T)) & (((P.act >> (P.k + 1u)) & ((U32(1u) << r) - 1u)) == U32(0u));
		const Pred leads = (r >= U32(4u)) & (r < U32(8u)) & (before < P.k) & ((P.act & ((U32(1u) << P.k) - 1u)) == U32(0u));
		const U32 plane = sel(follows, P.k + r, before);
		const U32 byte = (P.first >> (plane << 3)) & 0xFFu;
		const U32 where = sel(follows, pbase + R.size() + after, bbase + U32(hs) + before);
		const Pred nib = r == U32(0u);
		put_small(out, sel(nib, bbase * 8u + P.k * 4u, where * 8u), sel(nib, R.type(), byte), P.valid & (nib | follows | leads), own);
	}
	// a block without a slot: its SAME bytes (the type nibbles are all 0); constant indices keep the batch in scalar registers
#define STENOS_SAME_ONLY(q)                                                                                             \
	if (q < B.nblk && B.act[q] == 0) {                                                                               \
		const U32 byte = (U32(B.first[q]) >> (lane << 3)) & 0xFFu;                                                     \
		put_small(out, (U32(base[q] + hs) + lane) * 8u, byte, lane < U32(T), own);                                     \
	}
	STENOS_SAME_ONLY(0)
	STENOS_SAME_ONLY(1)
	STENOS_SAME_ONLY(2)
	STENOS_SAME_ONLY(3)
#undef STENOS_SAME_ONLY
	if (B.nslots == 0) {
		wave_sync();
		return;
	}
	slot_rows_emit_rows(lds, L, R, P.valid, pbase, 0, false);
}

// ---- groups of four blocks (bytesoftype 2 and 4) --------------------------------------------------------------------------
// Four consecutive blocks in which the same two planes are not constant -- 12-bit integers in 32-bit elements, every block
// of 16-bit samples -- are analysed plane by plane: the first pass takes the lower of the two planes of all four blocks, the
// second pass the upper one.  Planes of the same kind then share a pass: the noise in the low byte of such data is proven RAW
// (slot_rows_analyse) and written as it came, sixteen bytes per lane, and every lane of the other pass packs a row.  Taken
// block by block -- a noise plane and a coded plane of two blocks per pass -- every pass paid the whole analysis and the whole
// packing with half of its lanes.  Both passes are analysed before anything is written (a block's size is the sum of both),
// then written one behind the other; the rows of the first pass are read again from their slots for that.
constexpr uint32_t GROUP4_IMAGE_BYTES = 2112; // 15 waiting bytes + 4 x (2 + 2 + 2 x 256) and the slack of an 8-byte OR, in 16-byte groups

// the image of a group, zeroed, with the waiting bytes of the stream in front (block_codec.h, image_reset_fixed)
WV_FN void image_reset_group4(Lds lds, const Layout& L)
{
	const U32 lane = lane_id();
	U128 z;
	z.x = z.y = z.z = z.w = U32(0u);
	lds_st128(lds, U32(L.out) + lane * 16u, z, pred_all(true));
	lds_st128(lds, U32(L.out + 1024u) + lane * 16u, z, pred_all(true));
	lds_st128(lds, U32(L.out + 2048u) + lane * 16u, z, lane * 16u < U32(GROUP4_IMAGE_BYTES - 2048u));
	wave_sync();
	const Pred p = lane < U32(4u);
	const U32 a = sel(p, lane, U32(0u)) * 4u;
	lds_st32(lds, U32(L.out) + a, lds_ld32(lds, U32(L.out - 16u) + a), p);
	wave_sync();
}

// the rows of a pass back from their slots: the bytes, and their differences when a row codes differences
WV_FN void slot_rows_reload(Lds lds, const Layout& L, SlotRows& R, uint32_t slot_off, bool deltas)
{
	const U32 lane = lane_id();
	const U32 H(0x80808080u);
	const U128 v = lds_ld128(lds, U32(slot2_area(L) + slot_off) + lane * 16u);
	R.sb[0] = v.x ^ H;
	R.sb[1] = v.y ^ H;
	R.sb[2] = v.z ^ H;
	R.sb[3] = v.w ^ H;
	if (deltas) {
		const U32 above = row_shr(R.sb[3], 1, 0x80808080u);
		R.sd[0] = biased_sub(R.sb[0], prev_bytes(R.sb[0], above));
		for (int k = 1; k < 4; ++k)
			R.sd[k] = biased_sub(R.sb[k], prev_bytes(R.sb[k], R.sb[k - 1]));
	}
	else
		R.sd[0] = R.sd[1] = R.sd[2] = R.sd[3] = U32(0u);
}

// Type nibbles and the bytes of the constant planes of the four blocks of a group (block_compress.h:747-750, 1246-1257): in the
// sixteen lanes of block q, lane 0 writes the type of plane k0, lane 1 that of plane k1, lanes 2 and 3 (bytesoftype 4) the bytes
// of the two constant planes.  bbase: the block's start in the image; ts0, ts1: size | type << 16 of the block's two planes;
// firstv: an element of the block (any: the planes in question are constant).
WV_FN void group4_emit_heads(Lds lds, const Layout& L, uint32_t T, uint32_t k0, uint32_t k1, const U32& bbase, const U32& ts0, const U32& ts1, const U32& firstv, uint32_t scr)
{
	const U32 lane = lane_id();
	const U32 r = lane & 15u;
	Lds out = lds + L.out;
	const U32 own = U32(slot2_area(L) + scr - L.out) + lane * 16u;
	const uint32_t hs = header_bytes(T);
	// the planes the lanes 0..3 of a block stand for: k0, k1 and (bytesoftype 4) the two others, ascending
	uint32_t rest = 15u & ~((1u << k0) | (1u << k1));
	const uint32_t j0 = (uint32_t)__builtin_ctz(rest | 16u);
	rest &= rest - 1u;
	const uint32_t j1 = (uint32_t)__builtin_ctz(rest | 16u);
	const U32 plane = (U32(k0 | (k1 << 8) | (j0 << 16) | (j1 << 24)) >> (r << 3)) & 0xFFu;
	const Pred nib = r < U32(2u);
	const U32 type = sel(r == U32(0u), ts0, ts1) >> 16;
	// a constant plane stands behind the constant planes and the coded planes in front of it
	const Pred after0 = plane > U32(k0), after1 = plane > U32(k1);
	const U32 where = bbase + U32(hs) + plane - sel(after0, U32(1u), U32(0u)) - sel(after1, U32(1u), U32(0u)) + sel(after0, ts0 & 0xFFFFu, U32(0u)) +
			  sel(after1, ts1 & 0xFFFFu, U32(0u));
	const U32 byte = (firstv >> (plane << 3)) & 0xFFu;
	put_small(out, sel(nib, bbase * 8u + plane * 4u, where * 8u), sel(nib, type, byte), r < U32(T), own);
}

// ---- groups of any shape (bytesoftype 2 and 4) -----------------------------------------------------------------------------
// Consecutive blocks -- up to eight of 16-bit elements, four of 32-bit ones: sixteen registers of raw elements either way --
// whose planes that are not constant number eight at most are taken together whatever their shapes: the planes go into the eight
// slots in PLANE-MAJOR order (plane 0 of every block that has one, then plane 1, ...), so planes of one kind still share a pass
// and every pass is as full as the data allows -- a random walk of 16-bit samples has one or two planes to code per block in no
// fixed pattern, and pairs of its blocks filled three slots of four.  Which (block, plane) a slot holds is data here, not a
// shape known in advance: sixteen "element lanes" -- lane l stands for plane l mod T of block l / T -- work out the slot of
// their plane, fetch its size and type, add up the blocks and find where every plane and every constant byte goes; three tables
// of eight words in LDS (Layout::grp) carry first elements, sizes and positions between them and the row lanes.
constexpr uint32_t GRP_FIRST = 0, GRP_TS = 32, GRP_POS = 64; // offsets of the three tables inside Layout::grp

// the planes of `act` of a block into the slots slot_of[0], slot_of[1], ... (in plane order)
WV_FN void write_slots_at(Lds lds, const Layout& L, const RawBlock& b, uint32_t T, uint32_t act, const uint32_t* slot_of)
{
	const U32 a = U32(slot2_area(L)) + lane_id() * 4u;
	uint32_t k = 0;
	if (T == 2) {
		if (act & 1u)
			lds_st32(lds, a + slot_of[k++] * SLOT2_BYTES, perm_bytes(b.e.y, b.e.x, 0x06040200u), pred_all(true));
		if (act & 2u)
			lds_st32(lds, a + slot_of[k] * SLOT2_BYTES, perm_bytes(b.e.y, b.e.x, 0x07050301u), pred_all(true));
		return;
	}
	if (act & 3u) {
		const U32 t0 = perm_bytes(b.e.y, b.e.x, 0x05010400u), t2 = perm_bytes(b.e.w, b.e.z, 0x05010400u);
		if (act & 1u)
			lds_st32(lds, a + slot_of[k++] * SLOT2_BYTES, perm_bytes(t2, t0, 0x05040100u), pred_all(true));
		if (act & 2u)
			lds_st32(lds, a + slot_of[k++] * SLOT2_BYTES, perm_bytes(t2, t0, 0x07060302u), pred_all(true));
	}
	if (act & 12u) {
		const U32 t1 = perm_bytes(b.e.y, b.e.x, 0x07030602u), t3 = perm_bytes(b.e.w, b.e.z, 0x07030602u);
		if (act & 4u)
			lds_st32(lds, a + slot_of[k++] * SLOT2_BYTES, perm_bytes(t3, t1, 0x05040100u), pred_all(true));
		if (act & 8u)
			lds_st32(lds, a + slot_of[k] * SLOT2_BYTES, perm_bytes(t3, t1, 0x07060302u), pred_all(true));
	}
}

// What the element lanes of a group know once its passes are analysed (the sizes and types of the slots stand in the table
// GRP_TS): lane l < nb * T stands for plane l % T of block l / T.
struct GroupPlace {
	U32 pos;      // where the lane's plane (or constant byte) goes, from the image's start
	U32 ts;       // size | type << 16 of the lane's plane (1 | 0 for a constant one)
	U32 bstart;   // start of the lane's block in the image
	U32 bsize;    // size of the lane's block
	Pred valid, active;
	uint32_t total; // bytes of all blocks
};
// acts: bit l = plane l % T of block l / T is not constant; base: where the first block starts in the image
WV_FN GroupPlace group_place(Lds lds, const Layout& L, uint32_t T, uint32_t nb, uint32_t acts, uint32_t base)
{
	GroupPlace G;
	const U32 lane = lane_id();
	const uint32_t lg = T == 2 ? 1u : 2u, hs = header_bytes(T);
	const U32 j = lane & U32(T - 1u);
	G.valid = lane < U32(nb * T);
	G.active = G.valid & (((U32(acts) >> (lane & 31u)) & 1u) != U32(0u));
	// the slot of the lane's plane in plane-major order: the planes of lower number of all blocks, then this plane of the blocks in front
	const uint32_t every = T == 2 ? 0x5555u : 0x1111u; // plane 0 of every block
	const U32 lower = U32(every) * ((U32(1u) << j) - 1u);  // planes of lower number
	const U32 mine = U32(every) << j;
	const U32 slot = popc(U32(acts) & lower) + popc(U32(acts) & mine & ((U32(1u) << (lane & 31u)) - 1u));
	G.ts = sel(G.active, lds_ld32(lds, U32(L.grp + GRP_TS) + sel(G.active, slot, U32(0u)) * 4u), sel(G.valid, U32(1u), U32(0u)));
	const U32 zz = G.ts & 0xFFFFu;
	// inclusive sums inside the block's T lanes
	U32 incl = zz + sel(j >= U32(1u), shfl_up(zz, 1, 0u), U32(0u));
	if (T == 4)
		incl = incl + sel(j >= U32(2u), shfl_up(incl, 2, 0u), U32(0u));
	// the block's size in its last lane -> all of its lanes (a read from lane | (T - 1))
	const U32 bsum = shfl(incl, lane | U32(T - 1u));
	G.bsize = sel(G.valid, bsum + U32(hs), U32(0u));
	const U32 upto = wave_incl_scan(sel(j == U32(0u), G.bsize, U32(0u))); // blocks up to and with the lane's
	G.bstart = U32(base) + upto - G.bsize;
	G.pos = G.bstart + U32(hs) + (incl - zz);
	G.total = readlane(upto, 15);
	(void)lg;
	// the row lanes read where their slot's plane goes
	lds_st32(lds, U32(L.grp + GRP_POS) + sel(G.active, slot, U32(0u)) * 4u, G.pos, G.active);
	wave_sync();
	return G;
}
// type nibbles and the bytes of the constant planes of all blocks of the group, by the element lanes
WV_FN void group_emit_heads(Lds lds, const Layout& L, uint32_t T, const GroupPlace& G, uint32_t scr)
{
	const U32 lane = lane_id();
	Lds out = lds + L.out;
	const U32 own = U32(slot2_area(L) + scr - L.out) + lane * 16u;
	const U32 j = lane & U32(T - 1u);
	const U32 b = lane >> (T == 2 ? 1u : 2u);
	put_small(out, G.bstart * 8u + j * 4u, G.ts >> 16, G.active, own); // (a constant plane's type is 0: nothing to write)
	const U32 first = lds_ld32(lds, U32(L.grp + GRP_FIRST) + (b & 7u) * 4u);
	put_small(out, G.pos * 8u, (first >> (j << 3)) & 0xFFu, G.valid & !G.active, own);
}

// ---- bytesoftype 8 -----------------------------------------------------------------------------------------------------
// One block per batch: its non-constant planes take the four slots in plane order, the first four in one pass, the rest
// (doubles and 64-bit integers seldom have fewer than five) in a second one.  Both passes are analysed before anything is
// written -- the mini-LZ decision needs the block's whole size -- and then emitted one behind the other.  A lane holds four
// consecutive elements as (low, high) dword pairs; the low halves carry planes 0..3, the high halves planes 4..7, so the
// 4 x 4 byte transposes of bytesoftype 4 serve each half.
struct RawBlock8 {
	U128 a, b; // elements 4 lane + 0, 1 in a (low, high, low, high), 4 lane + 2, 3 in b
};
WV_FN RawBlock8 load_raw_block8(const uint8_t* g)
{
	RawBlock8 r;
	r.a = gld128_unaligned(g, lane_id() * 32u, pred_all(true));
	r.b = gld128_unaligned(g, lane_id() * 32u + 16u, pred_all(true));
	return r;
}
WV_FN RawBlock half_of(const RawBlock8& b, bool high)
{
	RawBlock h;
	h.e.x = high ? b.a.y : b.a.x;
	h.e.y = high ? b.a.w : b.a.z;
	h.e.z = high ? b.b.y : b.b.x;
	h.e.w = high ? b.b.w : b.b.z;
	return h;
}
struct SameScan8 {
	uint32_t act, nact;          // bit k: plane k is not constant
	uint32_t first_lo, first_hi; // the first element (the bytes of the SAME planes)
};
WV_FN SameScan8 scan_same8(const RawBlock8& b)
{
	SameScan8 s;
	s.first_lo = readlane(b.a.x, 0);
	s.first_hi = readlane(b.a.y, 0);
	const U32 f0(s.first_lo), f1(s.first_hi);
	const U32 xl = ((b.a.x ^ f0) | (b.a.z ^ f0)) | ((b.b.x ^ f0) | (b.b.z ^ f0));
	const U32 xh = ((b.a.y ^ f1) | (b.a.w ^ f1)) | ((b.b.y ^ f1) | (b.b.w ^ f1));
	s.act = mask_bit<1>(ballot((xl & 0xFFu) != U32(0u))) | mask_bit<2>(ballot((xl & 0xFF00u) != U32(0u))) | mask_bit<4>(ballot((xl & 0xFF0000u) != U32(0u))) |
		mask_bit<8>(ballot((xl & 0xFF000000u) != U32(0u))) | mask_bit<16>(ballot((xh & 0xFFu) != U32(0u))) | mask_bit<32>(ballot((xh & 0xFF00u) != U32(0u))) |
		mask_bit<64>(ballot((xh & 0xFF0000u) != U32(0u))) | mask_bit<128>(ballot((xh & 0xFF000000u) != U32(0u)));
	s.nact = (uint32_t)__builtin_popcount(s.act);
	return s;
}
// the planes of mask m (at most four) into slots 0, 1, ... in plane order
WV_FN void write_slots8(Lds lds, const Layout& L, const RawBlock8& b, uint32_t m)
{
	if (m & 15u)
		write_slots_fast(lds, L, half_of(b, false), 4, m & 15u, 0);
	if (m >> 4)
		write_slots_fast(lds, L, half_of(b, true), 4, m >> 4, (uint32_t)__builtin_popcount(m & 15u));
}
// Distinct hash keys among the first 80 values (all the reference's early-stop test sees, lz_precheck_values(8)): lanes
// 0..19 hold them, value k of a lane in round k.  As lz_distinct_keys_fast, with a key of its own (below).
WV_FN uint32_t lz_distinct_keys_fast8(Lds lds, const Layout& L, const RawBlock8& b)
{
	uint32_t distinct = 0;
	lanes_below(lz_precheck_values(8) / 4, [&](const Pred& in) {
		const U32 lane = lane_id();
		const U32 lo[4] = { b.a.x, b.a.z, b.b.x, b.b.z }, hi[4] = { b.a.y, b.a.w, b.b.y, b.b.w };
		U32 addr[4];
		for (int k = 0; k < 4; ++k) {
			// (any function of the value serves: a value can only match behind an equal one, which has the same key under every
			// hash, so the values that are the first with their key cannot match whatever table the reference keeps -- the bound
			// stays a lower bound of what the reference produces; the xor of the value's bytes costs five cheap instructions, the
			// reference's 64-bit multiplication three slow ones)
			U32 t = lo[k] ^ hi[k];
			t = t ^ (t >> 16);
			t = (t ^ (t >> 8)) & 0xFFu;
			addr[k] = U32(L.tab) + t * 4u;
			lds_st32(lds, addr[k], lane + U32(64u * (uint32_t)k), in);
		}
		wave_sync();
		for (int k = 0; k < 4; ++k)
			distinct += (uint32_t)__builtin_popcountll(ballot(in & (lds_ld32(lds, addr[k]) == lane + U32(64u * (uint32_t)k))));
	});
	wave_sync();
	return first_lane_value(distinct);
}
// Second rejection test of the mini-LZ for a block of bytesoftype 8 in registers (as lz_repeats_reject): a value can only
// match when an equal value precedes it; one bit per 13-bit hash of the 8-byte value bounds the number of such values from
// above, and if even with all of them matching the stream exceeds max_size the reference fails after doing all the work
// (lz_compress.h:221-223).  True: the attempt is pointless.  (Incompressible blocks always pass the first test -- their
// max_size is large -- and end here.)
WV_FN bool lz_repeats_reject8(Lds lds, const Layout& L, const RawBlock8& b, uint32_t max_size)
{
	const U32 lane = lane_id();
	U128 z;
	z.x = z.y = z.z = z.w = U32(0u);
	lds_st128(lds, U32(L.tab) + lane * 16u, z, pred_all(true));
	wave_sync();
	const U32 lo[4] = { b.a.x, b.a.z, b.b.x, b.b.z }, hi[4] = { b.a.y, b.a.w, b.b.y, b.b.w };
	uint32_t maybe = 0;
	for (int k = 0; k < 4; ++k) {
		const U32 h = ((lo[k] ^ (hi[k] * 0x9E3779B1u)) * 0x85EBCA6Bu) >> 19;
		const U32 bit = U32(1u) << (h & 31u);
		const U32 old = lds_or_rtn32(lds, U32(L.tab) + (h >> 5) * 4u, bit);
		maybe += (uint32_t)__builtin_popcountll(ballot((old & bit) != U32(0u)));
	}
	wave_sync();
	return 256 / 8 + 256 * 8 - maybe * 7 > max_size;
}
// What lane 16*s + r has to know about pass q of a block whose planes `act` are not constant (fields: tools/gen_shape_tables.py)
WV_FN U32 shape_lane_entry8(uint32_t act, uint32_t q)
{
	return gld32((const uint8_t*)SHAPE_LANES_T8 + (act * 2u + q) * 256u, lane_id() * 4u, pred_all(true));
}
// inclusive sums of the plane sizes over the slots of a pass; *total: the sum over its n slots
WV_FN U32 slot_pass_sizes(const SlotRows& R, uint32_t n, uint32_t* total)
{
	U32 incl = R.size() + scan_source(R.size(), 4, 0u);
	incl = incl + scan_source(incl, 5, 0u);
	*total = n ? readlane(incl, 16u * n - 1u) : 0u;
	return incl;
}
// One pass of a block of bytesoftype 8 into the zeroed image: e = shape_lane_entry8, bbase = the block's start in the image,
// before = bytes of the non-constant planes of the pass in front, incl = slot_pass_sizes.
WV_FN void slot_rows_emit8(Lds lds, const Layout& L, const SlotRows& R, const U32& e, const SameScan8& sc, uint32_t bbase, uint32_t before, const U32& incl)
{
	const U32 lane = lane_id();
	Lds out = lds + L.out;
	const U32 own = U32(slot2_area(L) - L.out) + lane * 16u;
	const Pred valid = (e & 8u) != U32(0u);
	// planes before mine: SAME ones take a byte each (k - j of them), the others are the slots before mine, in this pass and the one in front
	const U32 pbase = U32(bbase + header_bytes(8) + before) + ((e >> 4) & 7u) + (incl - R.size());
	{
		// lane 0 of a slot: the plane's type nibble; lanes 1..7: bytes of the SAME planes that follow the plane directly;
		// lanes 8..15 of the block's first slot: the SAME planes in front of it (block_compress.h:747-750, 1246-1257)
		const U32 role = (e >> 7) & 3u;
		const U32 f = sel((e & 0x200u) != U32(0u), U32(sc.first_hi), U32(sc.first_lo));
		const U32 byte = (f >> ((e >> 10) & 31u)) & 0xFFu;
		const U32 from = sel(role == U32(2u), pbase + R.size(), U32(bbase));
		put_small(out, from * 8u + ((e >> 15) & 127u), sel(role == U32(1u), R.type(), byte), role != U32(0u), own);
	}
	slot_rows_emit_rows(lds, L, R, valid, pbase);
}

} // namespace codec
