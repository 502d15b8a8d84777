// superblock_codec.h -- what one wavefront does around block_codec.h: moving blocks between HBM and
// its LDS scratch, the partial (tail) block, walking the blocks of one superblock on decode, and the
// byte-granular copies that assemble the frame.  Written in the wavevec.h vocabulary, so the host
// emulation in tests/emul exercises the same code the gfx950 kernels run.
//
// Reference behaviour restated here: block_compress / block_decompress outer loops
// (stenos/internal/block_compress.h:1152-1298, 1817-1878) and block_compress_partial (:947-1020).
#pragma once
#include "block_codec.h"

namespace codec {

// ---- HBM <-> LDS copies by one wave ---------------------------------------------------------------

// g must be 16-byte aligned; never reads past g + n
WV_FN void copy_g2l(Lds lds, uint32_t ldsoff, const uint8_t* g, uint32_t n)
{
	const U32 lane = lane_id();
	const uint32_t full = n & ~15u;
	for (uint32_t o = 0; o < full; o += 1024) {
		U32 off = U32(o) + lane * 16u;
		Pred p = off < U32(full);
		lds_st128(lds, U32(ldsoff) + off, gld128(g, off, p), p);
	}
	Pred t = lane < U32(n - full);
	lds_st8(lds, U32(ldsoff + full) + lane, gld8(g, U32(full) + lane, t), t);
}
// any alignment of g
WV_FN void copy_g2l_bytes(Lds lds, uint32_t ldsoff, const uint8_t* g, uint32_t n)
{
	const U32 lane = lane_id();
	for (uint32_t o = 0; o < n; o += 64) {
		Pred p = (U32(o) + lane) < U32(n);
		lds_st8(lds, U32(ldsoff + o) + lane, gld8(g, U32(o) + lane, p), p);
	}
}
WV_FN void load_block(Lds lds, uint32_t ldsoff, const uint8_t* g, uint32_t n)
{
	if ((((uintptr_t)g) & 15u) == 0)
		copy_g2l(lds, ldsoff, g, n);
	else
		copy_g2l_bytes(lds, ldsoff, g, n);
}
// LDS image -> HBM, never writes past g + n
WV_FN void store_block(uint8_t* g, Lds lds, uint32_t ldsoff, uint32_t n)
{
	const U32 lane = lane_id();
	if ((((uintptr_t)g) & 15u) == 0) {
		const uint32_t full = n & ~15u;
		for (uint32_t o = 0; o < full; o += 1024) {
			U32 off = U32(o) + lane * 16u;
			Pred p = off < U32(full);
			gst128(g, off, lds_ld128(lds, U32(ldsoff) + sel(p, off, U32(0u))), p);
		}
		Pred t = lane < U32(n - full);
		gst8(g, U32(full) + lane, lds_ld8(lds, U32(ldsoff + full) + sel(t, lane, U32(0u))), t);
	}
	else
		for (uint32_t o = 0; o < n; o += 64) {
			Pred p = (U32(o) + lane) < U32(n);
			gst8(g, U32(o) + lane, lds_ld8(lds, U32(ldsoff + o) + sel(p, lane, U32(0u))), p);
		}
}

// HBM -> HBM copy of n bytes by one wave; any alignment on both sides.  Source words are read from
// the 4-byte aligned addresses that contain the bytes (never below src & ~3, never at or past the
// aligned word that holds the last byte + 1), destination words are written aligned.
WV_FN void copy_g2g(uint8_t* dst, const uint8_t* src, uint32_t n)
{
	const U32 lane = lane_id();
	const uint32_t head = (uint32_t)((4u - ((uintptr_t)dst & 3u)) & 3u); // bytes until dst is 4-byte aligned
	const uint32_t h = head < n ? head : n;
	{
		Pred p = lane < U32(h);
		gst8(dst, lane, gld8(src, lane, p), p);
	}
	const uint32_t words = (n - h) >> 2;
	const uint32_t smis = (uint32_t)((uintptr_t)(src + h) & 3u);
	const uint8_t* sbase = src + h - smis; // 4-byte aligned
	const uint32_t sh = smis * 8u;
	for (uint32_t o = 0; o < words; o += 64) {
		U32 k = U32(o) + lane;
		Pred p = k < U32(words);
		U32 lo = gld32(sbase, k * 4u, p);
		U32 v = lo;
		if (sh) { // the upper word still holds at least one byte of [src, src + n)
			U32 hi = gld32(sbase, k * 4u + 4u, p);
			v = (lo >> U32(sh)) | (hi << U32(32u - sh));
		}
		gst32(dst + h, k * 4u, v, p);
	}
	const uint32_t done = h + words * 4;
	{
		Pred p = lane < U32(n - done);
		gst8(dst + done, lane, gld8(src + done, lane, p), p);
	}
}

// ---- encode side -----------------------------------------------------------------------------------

// One full block: HBM -> LDS -> encoded image -> 16-byte aligned slot.
WV_FN BlockInfo encode_block_job(Lds lds, const Layout& L, uint32_t T, const uint8_t* src, uint8_t* slot, bool allow_lz)
{
	load_block(lds, L.in, src, 256 * T);
	wave_sync();
	BlockInfo r = encode_full_block(lds, L, T, allow_lz);
	store_block(slot, lds, L.out, (r.size + 15u) & ~15u); // slots are padded to 16 bytes
	return r;
}

// ---- a run of consecutive blocks -> one contiguous byte stream ------------------------------------
//
// A wavefront that encodes consecutive blocks appends their images to a contiguous stream in HBM (its share
// of a superblock payload).  The stream is written in aligned 16-byte groups only.  The bytes of a block that do
// not fill a group wait, padded with zeros, in the 16 bytes in front of the LDS image (Layout::out - 16 .. out);
// the next block is encoded behind them (image offset pos % 16, block_codec.h image_reset), so every append is a
// plain aligned copy.
struct RunStream {
	uint8_t* base; // 16-byte aligned
	uint32_t pos;  // bytes appended so far; the last pos % 16 of them are still in LDS
};

// append the n bytes that were encoded at image offset rs.pos % 16 (the image starts with the waiting bytes)
WV_FN void stream_append(RunStream& rs, Lds lds, uint32_t out, uint32_t n)
{
	const U32 lane = lane_id();
	const uint32_t r = rs.pos & 15u;
	const uint32_t groups = (r + n) >> 4;
	uint8_t* g = rs.base + (rs.pos - r);
	for (uint32_t o = 0; o < groups; o += 64) {
		U32 k = U32(o) + lane;
		Pred p = k < U32(groups);
		gst128(g, k * 16u, lds_ld128(lds, U32(out) + sel(p, k, U32(0u)) * 16u), p);
	}
	// the group behind them (its bytes past the encoding are zero) waits in front of the image
	Pred t = lane < U32(4u);
	U32 a = sel(t, lane, U32(0u)) * 4u;
	U32 v = lds_ld32(lds, U32(out + groups * 16u) + a);
	lds_st32(lds, U32(out - 16u) + a, v, t);
	wave_sync();
	rs.pos += n;
}
WV_FN void stream_flush(const RunStream& rs, Lds lds, uint32_t out)
{
	const U32 lane = lane_id();
	const uint32_t r = rs.pos & 15u;
	Pred t = lane < U32(r);
	gst8(rs.base + (rs.pos - r), lane, lds_ld8(lds, U32(out - 16u) + sel(t, lane, U32(0u))), t);
}

// The same for a block of bytesoftype 2 or 4 whose non-constant planes have been analysed in slots slot_lo, slot_lo + 1, ...
// (L.plinfo[slot] = type | size << 8).  raw: the block in HBM when L.in does not hold it yet (only the mini-LZ reads L.in),
// null when it does.  *dirty: the slot images may have been overwritten by an LZ attempt (they share its scratch); set
// here when this block's attempt does so.
WV_FN BlockInfo finish_block(Lds lds, const Layout& L, uint32_t T, bool allow_lz, uint32_t base, const SameScan& sc, uint32_t slot_lo,
			     const uint8_t* raw, bool* dirty)
{
	uint32_t tab[4] = { 0u, 0u, 0u, 0u }; // per plane: type | offset << 8 | slot << 24
	SlotMap sm;
	uint32_t full, need;
	if (T <= 2) {
		// two planes: scalars are the shorter way
		const U32 pinfo = lds_ld32(lds, U32(L.plinfo) + (lane_id() & 3u) * 4u); // lane q: the entry of slot q
		uint32_t slot = slot_lo, slots[2] = { 0u, 0u };
		for (uint32_t k = 0; k < T; ++k) {
			if ((sc.act >> k) & 1u) {
				tab[k] = readlane(pinfo, slot);
				slots[k] = slot;
				++slot;
			}
			else
				tab[k] = PLANE_SAME | (1u << 8);
		}
		WV_MARK("plane_offsets");
		uint32_t nib = 0;
		for (uint32_t k = 0; k < T; ++k)
			nib |= (tab[k] & 0xFu) << (4 * k);
		full = plane_offsets_small(T, true, 16, tab, &need);
		for (uint32_t k = 0; k < T; ++k)
			tab[k] |= slots[k] << 24;
		sm.first = sc.first;
		sm.nib = U32(nib);
	}
	else {
		// The plane table is built on lanes (lane k of every quad: plane k) and only its results become scalars: scalar
		// instructions cost an issue slot like vector ones, and tables indexed by run-time values would leave the registers.
		const U32 lane = lane_id();
		const U32 k = lane & 3u;
		const Pred inblock = k < U32(T);
		const Pred active = inblock & (((U32(sc.act) >> k) & 1u) == U32(1u));
		const U32 slotv = U32(slot_lo) + popc(U32(sc.act) & ((U32(1u) << k) - 1u)); // slots follow the plane order
		U32 pinfo = lds_ld32(lds, U32(L.plinfo) + sel(active, slotv, U32(0u)) * 4u);
		pinfo = sel(active, pinfo, sel(inblock, U32(PLANE_SAME | (1u << 8)), U32(0u)));
		const U32 type = pinfo & 0xFFu, size = pinfo >> 8;
		WV_MARK("plane_offsets");
		U32 incl = size + sel(k >= U32(1u), row_shr(size, 1, 0), U32(0u)); // prefix sums inside the quad
		incl = incl + sel(k >= U32(2u), row_shr(incl, 2, 0), U32(0u));
		const uint32_t hs = header_bytes(T);
		const U32 off = U32(hs) + incl - size;
		full = readlane(incl, 3);
		// capacity the plane loop needs (block_compress.h:1241): see plane_offsets
		const uint32_t m = readlane(quad_max(sel(inblock & (type != U32(PLANE_RAW)), off + size + 16u, U32(0u))), 0);
		need = m > hs + full ? m : hs + full;
		const U32 tabv = type | (off << 8) | (sel(active, slotv, U32(0u)) << 24);
		for (uint32_t j = 0; j < 4; ++j)
			tab[j] = readlane(tabv, j);
		sm.first = sc.first;
		sm.nib = quad_add(sel(inblock, type << (k << 2), U32(0u)));
	}
	const bool eligible = T % 4 == 0 && full * 3 > 256 * T; // (:1210)
	BlockInfo r;
	r.info = full | (need << 15) | (eligible ? 1u << 30 : 0u);
	bool have_raw = raw == nullptr;
	if (allow_lz && eligible) {
		if (!have_raw) {
			load_block(lds, L.in, raw, 256 * T);
			wave_sync();
			have_raw = true;
		}
		uint32_t n = lz_try(lds, L, T, full, base, dirty);
		if (n) {
			r.size = n + 1;
			r.info |= 1u << 31;
			return r;
		}
	}
	if (*dirty) { // rare: the planes of this block go back into their slots
		if (!have_raw) {
			load_block(lds, L.in, raw, 256 * T);
			wave_sync();
		}
		write_slots(lds, L, load_plane_regs(lds, L.in, T, 0), T, sc.act, slot_lo);
		wave_sync();
	}
	WV_MARK("image_reset");
	image_reset(lds, L, base, header_bytes(T) + full);
	PlaneRegs none;
	none.valid = false;
	none.g = 0;
	emit_planes(lds, L, T, base, 16, none, tab, &sm);
	r.size = header_bytes(T) + full;
	return r;
}

// `nblocks` full blocks at src -> their encodings, back to back, at stage (16-byte aligned, room for
// nblocks * max_block_bytes(T) + 16).  Ample capacity is assumed (the mini-LZ is always tried), which is
// what the reference does for every superblock but the ones at the very end of a tight buffer.
WV_FN uint32_t encode_run(Lds lds, const Layout& L, uint32_t T, const uint8_t* src, uint32_t nblocks, uint8_t* stage)
{
	RunStream rs;
	rs.base = stage;
	rs.pos = 0;
	if (T == 2 || T == 4) {
		// Planes in slots (block_codec.h, analyse_slots): a block whose non-constant planes leave two slots free is
		// analysed together with its successor when that one fits into the rest.
		const uint32_t bs = 256 * T;
		uint32_t i = 0;
		while (i < nblocks) {
			const uint8_t* a = src + (uint64_t)i * bs;
			const uint8_t* b = a + bs;
			WV_MARK("load_block");
			// both blocks are requested at once, straight into registers; when the second one is not paired after all,
			// its load has at least brought it closer for the next round
			const bool fast = (((uintptr_t)a) & 15u) == 0;
			const bool has_b = fast && i + 1 < nblocks;
			PlaneRegs ra;
			RawBlock eb;
			if (fast) {
				const RawBlock ea = load_raw_block(a, T);
				if (has_b)
					eb = load_raw_block(b, T);
				store_raw_block(lds, L.in, ea, T);
				ra = plane_regs_of(ea, T);
			}
			else {
				load_block(lds, L.in, a, bs);
				wave_sync();
				ra = load_plane_regs(lds, L.in, T, 0);
			}
			WV_MARK("block_begin");
			const SameScan sa = scan_same(ra, T);
			write_slots(lds, L, ra, T, sa.act, 0);
			SameScan sb = sa;
			bool pair = false;
			if (has_b && sa.nact <= 2) {
				const PlaneRegs rb = plane_regs_of(eb, T);
				sb = scan_same(rb, T);
				if (sa.nact + sb.nact <= 4) {
					write_slots(lds, L, rb, T, sb.act, sa.nact);
					pair = true;
				}
			}
			const uint32_t nslots = sa.nact + (pair ? sb.nact : 0u);
			wave_sync();
			if (nslots)
				analyse_slots(lds, L, nslots);
			// the blocks of the batch, one after the other
			const uint32_t nblk = pair ? 2u : 1u;
			bool dirty = false;
			for (uint32_t k = 0; k < nblk; ++k) {
				SameScan sc;
				sc.act = k ? sb.act : sa.act;
				sc.nact = k ? sb.nact : sa.nact;
				sc.first = k ? sb.first : sa.first;
				const BlockInfo r = finish_block(lds, L, T, true, rs.pos & 15u, sc, k ? sa.nact : 0u, k ? b : nullptr, &dirty);
				WV_MARK("stream_append");
				stream_append(rs, lds, L.out, r.size);
				WV_MARK("block_end");
			}
			i += pair ? 2u : 1u;
		}
		stream_flush(rs, lds, L.out);
		return rs.pos;
	}
	for (uint32_t i = 0; i < nblocks; ++i) {
		load_block(lds, L.in, src + (uint64_t)i * (256 * T), 256 * T);
		wave_sync();
		BlockInfo r = encode_full_block(lds, L, T, true, rs.pos & 15u);
		stream_append(rs, lds, L.out, r.size);
	}
	stream_flush(rs, lds, L.out);
	return rs.pos;
}

// HBM -> HBM copy of n bytes by one wave with 16-byte stores; src may be read up to 31 bytes past src + n
// (and down to src & ~15), so it is only used on staging buffers that carry that slack.
WV_FN void copy_g2g_wide(uint8_t* dst, const uint8_t* src, uint32_t n)
{
	const U32 lane = lane_id();
	const uint32_t head = (uint32_t)((16u - ((uintptr_t)dst & 15u)) & 15u);
	const uint32_t h = head < n ? head : n;
	{
		Pred p = lane < U32(h);
		gst8(dst, lane, gld8(src, lane, p), p);
	}
	const uint32_t groups = (n - h) >> 4;
	const uint32_t smis = (uint32_t)((uintptr_t)(src + h) & 15u);
	const uint8_t* sbase = src + h - smis; // 16-byte aligned
	const uint32_t dsel = smis >> 2, sh = (smis & 3u) * 8u;
	for (uint32_t o = 0; o < groups; o += 64) {
		U32 k = U32(o) + lane;
		Pred p = k < U32(groups);
		U128 a = gld128(sbase, k * 16u, p);
		U128 v = a;
		if (smis) {
			U128 b = gld128(sbase, k * 16u + 16u, p);
			// dwords dsel .. dsel + 4 of {a, b}
			U32 t0 = dsel == 0 ? a.x : dsel == 1 ? a.y : dsel == 2 ? a.z : a.w;
			U32 t1 = dsel == 0 ? a.y : dsel == 1 ? a.z : dsel == 2 ? a.w : b.x;
			U32 t2 = dsel == 0 ? a.z : dsel == 1 ? a.w : dsel == 2 ? b.x : b.y;
			U32 t3 = dsel == 0 ? a.w : dsel == 1 ? b.x : dsel == 2 ? b.y : b.z;
			U32 t4 = dsel == 0 ? b.x : dsel == 1 ? b.y : dsel == 2 ? b.z : b.w;
			if (sh) {
				v.x = (t0 >> U32(sh)) | (t1 << U32(32u - sh));
				v.y = (t1 >> U32(sh)) | (t2 << U32(32u - sh));
				v.z = (t2 >> U32(sh)) | (t3 << U32(32u - sh));
				v.w = (t3 >> U32(sh)) | (t4 << U32(32u - sh));
			}
			else {
				v.x = t0;
				v.y = t1;
				v.z = t2;
				v.w = t3;
			}
		}
		gst128(dst + h, k * 16u, v, p);
	}
	const uint32_t done = h + groups * 16;
	{
		Pred p = lane < U32(n - done);
		gst8(dst + done, lane, gld8(src + done, lane, p), p);
	}
}

// The tail of a superblock payload: n < 256*T bytes -> [254] + partial block (block_compress.h:1277-1293).
// info.full is unused for tails; info.need is the capacity requirement counted from the 254 byte.
WV_FN BlockInfo encode_tail_job(Lds lds, const Layout& L, uint32_t T, const uint8_t* src, uint32_t n, uint8_t* slot)
{
	const U32 lane = lane_id();
	load_block(lds, L.in, src, n);
	wave_sync();
	// pad with the last byte up to a whole block (:967-968)
	U32 last = lds_ld8(lds, U32(L.in + n - 1));
	for (uint32_t o = n; o < 256 * T; o += 64)
		lds_st8(lds, U32(L.in + o) + lane, last, (U32(o) + lane) < U32(256 * T));
	wave_sync();
	const uint32_t lines = n / (16 * T);
	uint32_t need;
	uint32_t size = encode_partial_lines(lds, L, T, lines, &need);
	// bytes after the last complete line stay raw (:1011-1018)
	const uint32_t rem = n - lines * 16 * T;
	for (uint32_t o = 0; o < rem; o += 64) {
		Pred p = (U32(o) + lane) < U32(rem);
		U32 b = lds_ld8(lds, U32(L.in + lines * 16 * T + o) + sel(p, lane, U32(0u)));
		lds_put_bits(lds + L.out, (U32(size + o) + lane) * 8u, b, p);
	}
	wave_sync();
	size += rem;
	if (size > need) // dst + remaining > dst_end (:1013)
		need = size;
	store_block(slot, lds, L.out, (size + 15u) & ~15u);
	BlockInfo r;
	r.size = size;
	r.info = need << 15;
	return r;
}

// ---- decode side -----------------------------------------------------------------------------------

WV_HD uint32_t max_block_bytes(uint32_t T) { return 256 * T + header_bytes(T) + 1; }
WV_HD uint32_t max_tail_bytes(uint32_t T) { return 280 * T + header_bytes(T) + 2; }
// window size: room for the largest block plus a few KiB so that small blocks are decoded several per refill
WV_HD uint32_t window_bytes(uint32_t T) { return align16(max_tail_bytes(T) + 2048 + 32); }

WV_HD DecLayout make_dec_layout(uint32_t T)
{
	DecLayout L;
	L.win = 0;
	L.img = window_bytes(T) + 32;
	L.total = align16(L.img + 256 * T + 32);
	return L;
}

// Decode the payload of one BLOCK superblock (code 1): `csize` compressed bytes at src -> `dsize`
// bytes at dst.  Returns dsize, or DEC_ERROR on a malformed / truncated stream.
WV_FN uint32_t decode_superblock(Lds lds, const DecLayout& L, uint32_t T, const uint8_t* src, uint32_t csize, uint8_t* dst, uint32_t dsize)
{
	const U32 lane = lane_id_plain();
	const uint32_t bs = 256 * T, hs = header_bytes(T);
	if (dsize == 0 || csize == 0)
		return 0;
	const uint32_t nblocks = dsize / bs;
	if (csize < hs + T && nblocks) // block_compress.h:1813-1815
		return DEC_ERROR;
	const uint32_t wcap = window_bytes(T);
	const uint32_t mis = (uint32_t)((uintptr_t)src & 15u); // the window is filled from the 16-byte aligned address below src
	const uint8_t* abase = src - mis;
	uint32_t wstart = 0, wfill = 0; // window holds abase[wstart, wstart + wfill)
	uint32_t consumed = 0;          // payload bytes consumed so far

	auto ensure = [&](uint32_t need) {
		// make payload bytes [consumed, consumed + need) resident (need already clipped to the payload)
		uint32_t a = consumed + mis; // offset from abase
		if (a >= wstart && a + need <= wstart + wfill)
			return;
		const uint32_t nstart = a & ~15u;
		const uint32_t endoff = csize + mis;
		const uint32_t nfill = endoff - nstart < wcap ? endoff - nstart : wcap;
		uint32_t keep = 0;
		if (wfill && nstart >= wstart && nstart < wstart + wfill && ((wstart + wfill) & 15u) == 0) {
			// the bytes already in the window that are still needed slide to its start (LDS to LDS, ascending
			// 1 KiB chunks: a chunk is read completely before it is written and never overlaps a later source),
			// so every compressed byte is fetched from HBM once
			keep = wstart + wfill - nstart;
			const uint32_t d = nstart - wstart;
			for (uint32_t o = 0; o < keep; o += 1024) {
				U32 off = U32(o) + lane * 16u;
				Pred p = off < U32(keep);
				U128 v = lds_ld128(lds, U32(L.win + d) + sel(p, off, U32(0u)));
				wave_sync();
				lds_st128(lds, U32(L.win) + off, v, p);
				wave_sync();
			}
		}
		wstart = nstart;
		wfill = nfill;
		if (nfill > keep)
			copy_g2l(lds, L.win + keep, abase + wstart + keep, nfill - keep);
		wave_sync();
	};

	for (uint32_t b = 0; b < nblocks; ++b) {
		uint32_t left = csize - consumed;
		uint32_t need = left < max_block_bytes(T) ? left : max_block_bytes(T);
		ensure(need);
		uint32_t n = decode_block(lds, L, T, consumed + mis - wstart, need, 16, true);
		if (n == DEC_ERROR)
			return DEC_ERROR;
		store_block(dst + (size_t)b * bs, lds, L.img, bs);
		wave_sync();
		consumed += n;
	}
	const uint32_t tail = dsize - nblocks * bs;
	if (tail) { // [254] + partial block (:1862-1876, 1749-1795)
		if (consumed == csize)
			return DEC_ERROR;
		uint32_t left = csize - consumed;
		uint32_t need = left < max_tail_bytes(T) ? left : max_tail_bytes(T);
		ensure(need);
		uint32_t cur = consumed + mis - wstart;
		if (win_u8(lds + L.win, cur) != BLOCK_PARTIAL)
			return DEC_ERROR;
		const uint32_t lines = tail / (16 * T);
		uint32_t n = 0;
		if (lines) {
			n = decode_block(lds, L, T, cur + 1, need - 1, lines, false);
			if (n == DEC_ERROR)
				return DEC_ERROR;
		}
		const uint32_t rem = tail - lines * 16 * T;
		if (1 + n + rem > need)
			return DEC_ERROR;
		for (uint32_t o = 0; o < rem; o += 64) {
			Pred p = (U32(o) + lane) < U32(rem);
			U32 v = lds_ld8(lds + L.win, U32(cur + 1 + n + o) + sel(p, lane, U32(0u)));
			lds_st8(lds, U32(L.img + lines * 16 * T + o) + lane, v, p);
		}
		wave_sync();
		store_block(dst + (size_t)nblocks * bs, lds, L.img, tail);
		consumed += 1 + n + rem;
	}
	return dsize;
}

} // namespace codec
