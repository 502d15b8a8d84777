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

// `nblocks` full blocks at src -> their encodings, back to back, at stage (16-byte aligned, room for
// nblocks * max_block_bytes(T) + 16).  Ample capacity is assumed (the mini-LZ is always tried), which is
// what the reference does for every superblock but the ones at the very end of a tight buffer.
// slots: bytesoftype 2 and 4 go through the plane slots (two blocks per pass); false: the plane-group loop at the bottom,
// which is the shorter way for int32 data whose blocks have three or four non-constant planes (kernels.hip, probe_planes).
WV_FN uint32_t encode_run(Lds lds, const Layout& L, uint32_t T, const uint8_t* src, uint32_t nblocks, uint8_t* stage, bool slots = true)
{
	RunStream rs;
	rs.base = stage;
	rs.pos = 0;
	if (slots && (T == 2 || T == 4)) {
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
			SameScan sa;
			uint32_t keys0 = 0, keys1 = 0; // distinct hash keys at the head of each block (first rejection test of the mini-LZ)
			if (fast) {
				const RawBlock ea = load_raw_block(a, T);
				if (has_b)
					eb = load_raw_block(b, T);
				WV_MARK("block_begin");
				ra = plane_regs_of(ea, T);
				sa = T == 4 ? scan_same_raw(ea, T) : scan_same(ra, T); // two planes: one test each is the shorter way
				if (T == 4 && sa.nact >= 2) // with fewer non-constant planes the block is too small for the mini-LZ (:1210)
					keys0 = lz_distinct_keys_regs(lds, L, ea.e);
			}
			else {
				load_block(lds, L.in, a, bs);
				wave_sync();
				if (T == 4)
					keys0 = lz_distinct_keys(lds, L, T);
				ra = load_plane_regs(lds, L.in, T, 0);
				sa = scan_same(ra, T);
			}
			write_slots(lds, L, ra, T, sa.act, 0);
			SameScan sb = sa;
			bool pair = false;
			if (has_b && sa.nact <= 2) {
				PlaneRegs rb;
				if (T == 4)
					sb = scan_same_raw(eb, T);
				else {
					rb = plane_regs_of(eb, T);
					sb = scan_same(rb, T);
				}
				if (sa.nact + sb.nact <= 4) {
					if (T == 4)
						rb = plane_regs_of(eb, T);
					write_slots(lds, L, rb, T, sb.act, sa.nact);
					pair = true;
					if (T == 4 && sb.nact >= 2)
						keys1 = lz_distinct_keys_regs(lds, L, eb.e);
				}
			}
			const uint32_t nslots = sa.nact + (pair ? sb.nact : 0u);
			wave_sync();
			if (nslots)
				analyse_slots(lds, L, nslots);
			const uint32_t nblk = pair ? 2u : 1u, hs = header_bytes(T);
			const uint32_t act1 = pair ? sb.act : 0u;
			BatchPlan P = plan_batch(lds, L, T, sa.act, act1, nblk);
			const uint32_t full0 = P.full[0], full1 = P.full[1];
			bool replan = false; // an LZ attempt ran in between: the plan is rebuilt rather than kept in registers across it
			const uint32_t size0 = hs + full0, size1 = pair ? hs + full1 : 0u;
			uint32_t pending = pair ? 3u : 1u; // blocks whose planes still have to be written
			// Blocks that try the mini-LZ (block_compress.h:1210-1221): those that pass its first rejection test (most do
			// not; the key counts were taken while the blocks were in registers).  Its table is the image, so the second
			// block's attempt has to wait until the first block is out; otherwise both blocks are written in one pass.
			uint32_t lzq = 0;
			if (T == 4) {
				if (full0 * 3 > bs && lz_precheck_passes(T, keys0, full0))
					lzq |= 1u;
				if (pair && full1 * 3 > bs && lz_precheck_passes(T, keys1, full1))
					lzq |= 2u;
			}
			bool dirty = false; // an attempt has overwritten the slot images (they share its scratch)
			for (;;) {
				const uint32_t blk = (lzq & 1u) ? 0u : 1u;
				if (lzq && !(blk == 1u && (pending & 1u))) {
					lzq &= ~(1u << blk);
					load_block(lds, L.in, blk ? b : a, bs); // only the mini-LZ reads L.in
					wave_sync();
					replan = true;
					const uint32_t n = lz_try(lds, L, T, blk ? full1 : full0, rs.pos & 15u, &dirty);
					if (n) {
						stream_append(rs, lds, L.out, n + 1);
						pending &= ~(1u << blk);
					}
					continue;
				}
				if (!pending)
					break;
				// write planes: both blocks in one pass, back to back in the image, when nothing stands in between
				uint32_t mask = (pending & 1u) ? 1u : 2u;
				if (pending == 3u && !lzq && (rs.pos & 15u) + size0 + size1 + 32u <= out_capacity(T))
					mask = 3u;
				if (dirty) { // rare: the planes go back into their slots
					for (uint32_t q = 0; q < nblk; ++q) {
						load_block(lds, L.in, q ? b : a, bs);
						wave_sync();
						write_slots(lds, L, load_plane_regs(lds, L.in, T, 0), T, q ? sb.act : sa.act, q ? sa.nact : 0u);
						wave_sync();
					}
					dirty = false;
				}
				const uint32_t base = rs.pos & 15u;
				const uint32_t bytes = ((mask & 1u) ? size0 : 0u) + ((mask & 2u) ? size1 : 0u);
				if (replan || T == 4) { // int32: rebuilding it is cheaper than the registers it would hold meanwhile
					P = plan_batch(lds, L, T, sa.act, act1, nblk);
					replan = false;
				}
				WV_MARK("image_reset");
				image_reset(lds, L, base, bytes);
				emit_batch(lds, L, T, P, sa.first, sb.first, mask, base, (mask & 1u) ? base + size0 : base, sa.nact, nslots);
				WV_MARK("stream_append");
				stream_append(rs, lds, L.out, bytes);
				pending &= ~mask;
			}
			WV_MARK("block_end");
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

// The decoder checks the bytes a block consumed once per block, not before every read: a block whose first byte is in
// the window reads at most hs + T*(8 + 18 + 16*18) + 16 bytes from there whatever the stream contains, so that much LDS
// has to follow the window's buffer (the image and some padding behind it; stale bytes are harmless).
WV_HD uint32_t max_block_reach(uint32_t T) { return header_bytes(T) + T * 314 + 32; }
WV_HD DecLayout make_dec_layout(uint32_t T)
{
	DecLayout L;
	L.win = 0;
	L.img = window_bytes(T) + 32;
	const uint32_t after = 256 * T + 32;
	L.total = align16(L.img + (after > max_block_reach(T) ? after : max_block_reach(T)));
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
	uint32_t wstart = 0, wfill = 0, wend = 0; // window holds abase[wstart, wend), wend = wstart + wfill
	uint32_t consumed = 0;          // payload bytes consumed so far

	auto ensure = [&](uint32_t need) {
		// make payload bytes [consumed, consumed + need) resident (need already clipped to the payload)
		uint32_t a = consumed + mis; // offset from abase; never below wstart: the window only moves forward with it
		if (a + need <= wend)
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
		wend = nstart + nfill;
		if (nfill > keep)
			copy_g2l(lds, L.win + keep, abase + wstart + keep, nfill - keep);
		wave_sync();
	};

	for (uint32_t b = 0; b < nblocks; ++b) {
		WV_MARK("dec_block_begin");
		uint32_t left = csize - consumed;
		uint32_t need = left < max_block_bytes(T) ? left : max_block_bytes(T);
		ensure(need);
		WV_MARK("dec_block");
		uint32_t n = decode_block(lds, L, T, consumed + mis - wstart, need, 16, true);
		if (n == DEC_ERROR)
			return DEC_ERROR;
		WV_MARK("dec_block_store");
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
