// superblock_codec.h -- what one wavefront does around block_codec.h: moving blocks between HBM and
// its LDS scratch, the partial (tail) block, walking the blocks of one superblock on decode, and the
// byte-granular copies that assemble the frame.  Written in the wavevec.h vocabulary, so the host
// emulation in tests/emul exercises the same code the gfx950 kernels run.
//
// Reference behaviour restated here: block_compress / block_decompress outer loops
// (stenos/internal/block_compress.h:1152-1298, 1817-1878) and block_compress_partial (:947-1020).
#pragma once
#include "slot_codec.h"
#include <type_traits>

namespace codec {

#ifdef WV_HOST_EMULATION
inline uint64_t& emul_group4_count()
{
	static uint64_t n = 0;
	return n;
}
inline uint64_t& emul_group_any_count()
{
	static uint64_t n = 0;
	return n;
}
#endif

// ---- HBM <-> LDS copies by one wave ---------------------------------------------------------------

#ifdef WV_PREDICATE_BRANCHES // (the encoders: predicates as branches, wavevec.h)
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
WV_FN void load_block(Lds lds, uint32_t ldsoff, const uint8_t* g, uint32_t n)
{
	const U32 lane = lane_id();
	const uint32_t full = n & ~15u;
	for (uint32_t o = 0; o < full; o += 1024) {
		U32 off = U32(o) + lane * 16u;
		Pred p = off < U32(full);
		lds_st128(lds, U32(ldsoff) + off, gld128_unaligned(g, off, p), p);
	}
	Pred t = lane < U32(n - full);
	lds_st8(lds, U32(ldsoff + full) + lane, gld8(g, U32(full) + lane, t), t);
}
// LDS image -> HBM, never writes past g + n
WV_FN void store_block(uint8_t* g, Lds lds, uint32_t ldsoff, uint32_t n)
{
	const U32 lane = lane_id();
	const uint32_t full = n & ~15u;
	for (uint32_t o = 0; o < full; o += 1024) {
		U32 off = U32(o) + lane * 16u;
		Pred p = off < U32(full);
		gst128_unaligned(g, off, lds_ld128(lds, U32(ldsoff) + sel(p, off, U32(0u))), p);
	}
	Pred t = lane < U32(n - full);
	gst8(g, U32(full) + lane, lds_ld8(lds, U32(ldsoff + full) + sel(t, lane, U32(0u))), t);
}

#else
// The copies below take no predicate: a lane beyond the end repeats the last group (or byte) -- the same bytes to the same
// place -- so every access is an ordinary one the compiler schedules and waits for as usual, and there is no divergent
// branch around it (wavevec.h, "Predicated memory accesses without a branch").

// g must be 16-byte aligned; never reads past g + n
WV_FN void copy_g2l(Lds lds, uint32_t ldsoff, const uint8_t* g, uint32_t n)
{
	const U32 lane = lane_id();
	const uint32_t full = n & ~15u;
	for (uint32_t o = 0; o < full; o += 1024) {
		const U32 off = umin(U32(o) + lane * 16u, U32(full - 16u));
		lds_st128(lds, U32(ldsoff) + off, gld128(g, off, pred_all(true)), pred_all(true));
	}
	if (n - full) {
		const U32 k = U32(full) + umin(lane, U32(n - full - 1u));
		lds_st8(lds, U32(ldsoff) + k, gld8(g, k, pred_all(true)), pred_all(true));
	}
}
// any alignment of g
WV_FN void load_block(Lds lds, uint32_t ldsoff, const uint8_t* g, uint32_t n)
{
	const U32 lane = lane_id();
	const uint32_t full = n & ~15u;
	for (uint32_t o = 0; o < full; o += 1024) {
		const U32 off = umin(U32(o) + lane * 16u, U32(full - 16u));
		lds_st128(lds, U32(ldsoff) + off, gld128_unaligned(g, off, pred_all(true)), pred_all(true));
	}
	if (n - full) {
		const U32 k = U32(full) + umin(lane, U32(n - full - 1u));
		lds_st8(lds, U32(ldsoff) + k, gld8(g, k, pred_all(true)), pred_all(true));
	}
}
// LDS image -> HBM, never writes past g + n
WV_FN void store_block(uint8_t* g, Lds lds, uint32_t ldsoff, uint32_t n)
{
	const U32 lane = lane_id();
	const uint32_t full = n & ~15u;
	for (uint32_t o = 0; o < full; o += 1024) {
		const U32 off = umin(U32(o) + lane * 16u, U32(full - 16u));
		gst128_unaligned(g, off, lds_ld128(lds, U32(ldsoff) + off), pred_all(true));
	}
	if (n - full) {
		const U32 k = U32(full) + umin(lane, U32(n - full - 1u));
		gst8(g, k, lds_ld8(lds, U32(ldsoff) + k), pred_all(true));
	}
}

#endif

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
	{ // the first 64 groups without a loop around them: that is all of them unless the blocks hardly compress
		Pred p = lane < U32(groups);
		gst128(g, lane * 16u, lds_ld128(lds, U32(out) + sel(p, lane, U32(0u)) * 16u), p);
	}
	for (uint32_t o = 64; o < groups; o += 64) {
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
// nothing waits at the start of a stream
WV_FN void stream_begin(Lds lds, uint32_t out)
{
	const U32 lane = lane_id();
	Pred t = lane < U32(4u);
	lds_st32(lds, U32(out - 16u) + sel(t, lane, U32(0u)) * 4u, U32(0u), t);
	wave_sync();
}
WV_FN void stream_flush(const RunStream& rs, Lds lds, uint32_t out)
{
	const U32 lane = lane_id();
	const uint32_t r = rs.pos & 15u;
	Pred t = lane < U32(r);
	gst8(rs.base + (rs.pos - r), lane, lds_ld8(lds, U32(out - 16u) + sel(t, lane, U32(0u))), t);
}

// Where the encodings of consecutive blocks go.  A sink names the image the next block is written into (at), the byte of
// that image it starts at (base, < 16) and takes the bytes over once they are written (append).
struct StreamSink { // a contiguous stream in HBM (RunStream)
	RunStream rs;
	bool writes; // false: sizes only, nothing is written (a superblock that is expected to end up as a copy, kernels.hip)
	// A superblock that is only measured because it will probably be stored as a copy: where its raw bytes would go in that
	// case (kernels.hip, speculative copy); the blocks are stored there from the registers they were loaded into anyway.
	// nullptr: nowhere.  Every path of encode_blocks_to stores the blocks it takes (the row-lane passes and groups from the
	// registers they loaded them into, the plane-group loop from its LDS copy): kernels.hip relies on that when it skips the copy.
	uint8_t* raw_to = nullptr;
	// Whether the last blocks the wave saw had the shape that groups of four want (encode_blocks_to): kept by the caller from
	// run to run, so that a run of noise does not begin with four loads for nothing.  nullptr: every run tries once.
	bool* group_hint = nullptr;
	WV_MFN void raw8(const RawBlock8& b, uint32_t block)
	{
		gst128_through(raw_to + (uint64_t)block * 2048u, lane_id() * 32u, b.a);
		gst128_through(raw_to + (uint64_t)block * 2048u, lane_id() * 32u + 16u, b.b);
	}
	WV_MFN void raw(const RawBlock& b, uint32_t T, uint32_t block)
	{
		if (T == 2)
			gst64_through(raw_to + (uint64_t)block * 512u, lane_id() * 8u, b.e.x, b.e.y);
		else
			gst128_through(raw_to + (uint64_t)block * 1024u, lane_id() * 16u, b.e);
	}
	WV_MFN Layout at(const Layout& L) const { return L; }
	WV_MFN uint32_t base() const { return rs.pos & 15u; }
	WV_MFN void append(Lds lds, const Layout& L, uint32_t n)
	{
		if (writes)
			stream_append(rs, lds, L.out, n);
		else
			rs.pos += n;
	}
};

// `nblocks` full blocks at src -> their encodings, back to back, into the sink.  Ample capacity is assumed (the
// mini-LZ is always tried), which is what the reference does for every superblock but the ones at the very end of a
// tight buffer.
// slots: bytesoftype 2 and 4 go through the plane slots (slot_codec.h, up to two blocks per pass); false: the plane-group
// loop at the bottom.
// When to try the noise proof of a pass (slot_codec.h, slot_rows_analyse): a failed attempt costs a dozen vector instructions
// that a pass of coded planes pays for nothing, so after one the next seven passes of that position go without, then one tries
// again (data changes character seldom, and a pass of noise that is not tried is merely analysed in full).
struct RawProofHint {
	uint32_t wait = 0;
	WV_MFN bool want()
	{
		if (wait == 0)
			return true;
		--wait;
		return false;
	}
	WV_MFN void tried(bool all_raw) { wait = all_raw ? 0u : 7u; }
};
struct NoPassHook {
	WV_MFN void operator()() const {}
};
// hook(): called after every pass of the slot loop (a wave may have other work waiting: kernels.hip, early store)
template <class Sink, class Hook = NoPassHook>
WV_FN void encode_blocks_to(Sink& sink, Lds lds, const Layout& L, uint32_t T, const uint8_t* src, uint32_t nblocks, bool slots, Hook hook = Hook())
{
	if (slots && (T == 2 || T == 4 || T == 8))
		slot_write_rle_lut(lds, L);
	if (slots && (T == 2 || T == 4)) {
		// Planes in slots (slot_codec.h): the non-constant planes of a block, and of its successor when they fit into
		// the four slots together, are analysed and written by row lanes in one pass.
		const uint32_t bs = 256 * T, hs = header_bytes(T);
		uint32_t i = 0;
		// After a wide batch the next pass probably is one too and asks for its four blocks at once: passes that start below
		// this block number do (0: none).
		uint32_t wide_ahead = 0;
		// Four blocks with the same two planes that are not constant, taken plane by plane (slot_codec.h, "groups of four
		// blocks").  False: the blocks are not of that shape, or one of them may go to the mini-LZ -- nothing has been written,
		// the pass below takes them.  try_group: the last blocks seen had the shape (or nothing is known yet).
		bool try_group = sink.group_hint ? *sink.group_hint : true;
		RawProofHint proof0, proof1; // (the first and the second pass of a group; the passes below use the first)
		auto group4 = [&]() __attribute__((always_inline)) -> bool {
			const uint8_t* a = src + (uint64_t)i * bs;
			WV_MARK("g4_load");
			const RawBlock e0 = load_raw_block(a, T), e1 = load_raw_block(a + bs, T), e2 = load_raw_block(a + 2 * bs, T), e3 = load_raw_block(a + 3 * bs, T);
			const Layout M = sink.at(L);
			WV_MARK("g4_front");
			const SameScan s0 = scan_same_fast(e0, T);
			if (s0.nact != 2) {
				try_group = false;
				return false;
			}
			const SameScan s1 = scan_same_fast(e1, T), s2 = scan_same_fast(e2, T), s3 = scan_same_fast(e3, T);
			if (s1.act != s0.act || s2.act != s0.act || s3.act != s0.act) {
				try_group = false;
				return false;
			}
			const uint32_t k0 = (uint32_t)__builtin_ctz(s0.act), k1 = 31u - (uint32_t)__builtin_clz(s0.act);
			uint32_t keys0 = 0, keys1 = 0, keys2 = 0, keys3 = 0; // (first rejection test of the mini-LZ, as in the pass below)
			if (T == 4) {
				keys0 = lz_distinct_keys_fast<2>(lds, M, e0.e);
				keys1 = lz_distinct_keys_fast<2>(lds, M, e1.e);
				keys2 = lz_distinct_keys_fast<2>(lds, M, e2.e);
				keys3 = lz_distinct_keys_fast<2>(lds, M, e3.e);
			}
			// plane k0 of block q -> slot q, plane k1 -> slot 4 + q
			write_slots_fast(lds, M, e0, T, s0.act, 0, 4);
			write_slots_fast(lds, M, e1, T, s0.act, 1, 4);
			write_slots_fast(lds, M, e2, T, s0.act, 2, 4);
			write_slots_fast(lds, M, e3, T, s0.act, 3, 4);
			// an element of its block for the sixteen lanes of each block (the bytes of the constant planes)
			const U32 firstv = T == 4 ? row_select4v(e0.e.x, e1.e.x, e2.e.x, e3.e.x) : U32(0u);
			if (sink.raw_to) { // (measured only, probably a copy: the raw bytes go where the copy would put them)
				sink.raw(e0, T, i);
				sink.raw(e1, T, i + 1);
				sink.raw(e2, T, i + 2);
				sink.raw(e3, T, i + 3);
			}
			wave_sync();
			SlotRows R;
			WV_MARK("g4_pass0");
			const bool try0 = proof0.want();
			const bool raw0 = slot_rows_analyse(lds, M, R, 0, 4, try0);
			if (try0)
				proof0.tried(raw0);
			const U32 ts0 = R.ts, hm0 = R.hm, pm0 = R.pm;
			const Pred emitmin0 = R.emitmin, eq0 = R.eq;
			WV_MARK("g4_pass1");
			const bool try1 = proof1.want();
			const bool raw1 = slot_rows_analyse(lds, M, R, 1024, 4, try1);
			if (try1)
				proof1.tried(raw1);
			WV_MARK("g4_sizes");
			const U32 bsz = (ts0 & 0xFFFFu) + R.size() + U32(hs + T - 2u); // the block of the lane's slot
			U32 incl = bsz + scan_source(bsz, 4, 0u);
			incl = incl + scan_source(incl, 5, 0u);
			const uint32_t total = readlane(incl, 63);
			if (T == 4) {
				// a block the mini-LZ may take (block_compress.h:1210-1221): the pass below decides and encodes it
				const uint32_t f0 = readlane(bsz, 0) - hs, f1 = readlane(bsz, 16) - hs, f2 = readlane(bsz, 32) - hs, f3 = readlane(bsz, 48) - hs;
				if ((f0 * 3 > bs && lz_precheck_passes(T, keys0, f0)) || (f1 * 3 > bs && lz_precheck_passes(T, keys1, f1)) ||
				    (f2 * 3 > bs && lz_precheck_passes(T, keys2, f2)) || (f3 * 3 > bs && lz_precheck_passes(T, keys3, f3))) {
					try_group = false; // (the groups of any shape look at such blocks again; then the pass)
					return false;
				}
			}
			if (sink.writes) {
				WV_MARK("g4_emit1");
				image_reset_group4(lds, M);
				const U32 bbase = U32(sink.base()) + incl - bsz;
				// the second pass first: its rows are in registers; the slots of that pass are the scratch of both emissions
				slot_rows_emit_rows(lds, M, R, pred_all(true), bbase + U32(hs + k1 - 1u) + (ts0 & 0xFFFFu), 1024, T == 2);
				group4_emit_heads(lds, M, T, k0, k1, bbase, ts0, R.ts, firstv, 1024);
				WV_MARK("g4_emit0");
				slot_rows_reload(lds, M, R, 0, !raw0);
				R.ts = ts0, R.hm = hm0, R.pm = pm0;
				R.emitmin = emitmin0, R.eq = eq0;
				slot_rows_emit_rows(lds, M, R, pred_all(true), bbase + U32(hs + k0), 1024, T == 2);
			}
			WV_MARK("g4_append");
			sink.append(lds, M, total);
			WV_MARK("g4_end");
#ifdef WV_HOST_EMULATION
			++emul_group4_count(); // (tests/emul: the tests assert that the inputs meant for this path take it)
#endif
			i += 4;
			hook();
			return true;
		};
		// Blocks of any shapes whose planes that are not constant fill the eight slots, plane-major (slot_codec.h, "groups of any
		// shape").  False: fewer than two blocks fit, or one of them may go to the mini-LZ -- nothing has been written, the pass
		// below takes them.
		bool try_any = true;
		auto group_any = [&]() __attribute__((always_inline)) -> bool {
			constexpr uint32_t NBMAX = 8;
			const uint32_t NB = T == 2 ? 8u : 4u;
			const uint32_t avail = nblocks - i < NB ? nblocks - i : NB;
			const uint8_t* a = src + (uint64_t)i * bs;
			WV_MARK("ga_load");
			RawBlock e[NBMAX];
#pragma unroll
			for (uint32_t q = 0; q < NBMAX; ++q)
				if (q < NB && q < avail)
					e[q] = load_raw_block(a + (uint64_t)q * bs, T);
			const Layout M = sink.at(L);
			WV_MARK("ga_front");
			// the blocks that fit: their masks packed T bits a block, their first elements in the table
			uint32_t acts = 0, nb = 0, total = 0, keys[4] = { 0, 0, 0, 0 };
			bool open = true;
#pragma unroll
			for (uint32_t q = 0; q < NBMAX; ++q)
				if (q < NB && q < avail && open) {
					const SameScan sq = scan_same_fast(e[q], T);
					if (total + sq.nact <= 8) {
						acts |= sq.act << (T * q);
						total += sq.nact;
						nb = q + 1;
						lds_st32(lds, U32(M.grp + GRP_FIRST + 4 * q), U32(sq.first), lane_id() == U32(0u));
						if (T == 4 && sq.nact >= 2)
							keys[q & 3u] = lz_distinct_keys_fast<2>(lds, M, e[q].e);
					}
					else
						open = false;
				}
			if (nb < 2) {
				try_any = false;
				return false;
			}
			// slots in plane-major order: the planes of lower number of all blocks, then this plane of the blocks in front
			{
				const uint32_t every = T == 2 ? 0x5555u : 0x1111u;
#pragma unroll
				for (uint32_t q = 0; q < NBMAX; ++q)
					if (q < NB && q < nb) {
						const uint32_t actq = (acts >> (T * q)) & ((1u << T) - 1u);
						uint32_t slot_of[4] = { 0, 0, 0, 0 }, k = 0;
						for (uint32_t jj = 0; jj < T; ++jj)
							if ((actq >> jj) & 1u) {
								const uint32_t lower = every * ((1u << jj) - 1u), mine = every << jj;
								slot_of[k++] = (uint32_t)__builtin_popcount(acts & lower) + (uint32_t)__builtin_popcount(acts & mine & ((1u << (T * q + jj)) - 1u));
							}
						write_slots_at(lds, M, e[q], T, actq, slot_of);
					}
			}
			if (sink.raw_to) { // (measured only, probably a copy: the raw bytes go where the copy would put them)
#pragma unroll
				for (uint32_t q = 0; q < NBMAX; ++q)
					if (q < NB && q < nb)
						sink.raw(e[q], T, i + q);
			}
			wave_sync();
			const uint32_t n0 = total < 4 ? total : 4, n1 = total - n0;
			SlotRows R;
			U32 ts0(0u), hm0(0u), pm0(0u);
			Pred emitmin0 = pred_all(false), eq0 = pred_all(false);
			bool raw0 = false;
			const U32 lane = lane_id();
			if (n0) {
				WV_MARK("ga_pass0");
				const bool try0 = proof0.want();
				raw0 = slot_rows_analyse(lds, M, R, 0, n0, try0);
				if (try0)
					proof0.tried(raw0);
				lds_st32(lds, U32(M.grp + GRP_TS) + (lane >> 4) * 4u, R.ts, ((lane & 15u) == U32(0u)) & ((lane >> 4) < U32(n0)));
			}
			if (n1) {
				ts0 = R.ts, hm0 = R.hm, pm0 = R.pm;
				emitmin0 = R.emitmin, eq0 = R.eq;
				WV_MARK("ga_pass1");
				const bool try1 = proof1.want();
				const bool raw1 = slot_rows_analyse(lds, M, R, 1024, n1, try1);
				if (try1)
					proof1.tried(raw1);
				lds_st32(lds, U32(M.grp + GRP_TS + 16) + (lane >> 4) * 4u, R.ts, ((lane & 15u) == U32(0u)) & ((lane >> 4) < U32(n1)));
			}
			wave_sync();
			WV_MARK("ga_place");
			const GroupPlace G = group_place(lds, M, T, nb, acts, sink.base());
			if (T == 4) {
				// a block the mini-LZ may take (block_compress.h:1210-1221): the pass below decides and encodes it
				bool cand = false;
#pragma unroll
				for (uint32_t q = 0; q < 4; ++q)
					if (q < nb && !cand) {
						const uint32_t f = readlane(G.bsize, 4 * q) - hs;
						if (f * 3 > bs && lz_precheck_passes(T, keys[q], f)) {
							// (the count over 40 values turns most blocks away; one it does not is looked at again, as in the pass below: the
							// test that turns noise away, then the count over all 80 values)
							const RawBlock again = load_raw_block(a + (uint64_t)q * bs, T);
							cand = !lz_repeats_reject(lds, M, again.e, f) && lz_precheck_passes(T, lz_distinct_keys_fast<4>(lds, M, again.e), f);
						}
					}
				if (cand) {
					try_any = false;
					return false;
				}
			}
			if (sink.writes) {
				WV_MARK("ga_emit");
				image_reset_group4(lds, M);
				group_emit_heads(lds, M, T, G, 1024);
				// the rows of the pass that is in registers, then (two passes) the first one's, read again from their slots;
				// the second KiB of the slot area is the scratch of both emissions
				const U32 sl = lane >> 4;
				if (n1) {
					const U32 pb1 = lds_ld32(lds, U32(M.grp + GRP_POS + 16) + sl * 4u);
					slot_rows_emit_rows(lds, M, R, sl < U32(n1), pb1, 1024, T == 2);
					slot_rows_reload(lds, M, R, 0, !raw0);
					R.ts = ts0, R.hm = hm0, R.pm = pm0;
					R.emitmin = emitmin0, R.eq = eq0;
				}
				if (n0) {
					const U32 pb0 = lds_ld32(lds, U32(M.grp + GRP_POS) + sl * 4u);
					slot_rows_emit_rows(lds, M, R, sl < U32(n0), pb0, 1024, T == 2);
				}
			}
			WV_MARK("ga_append");
			sink.append(lds, M, G.total);
			// (blocks of one shape with two planes each are the groups of four's: cheaper placement)
			try_group = nb == 4 && total == 8 && ((acts ^ (acts >> T)) & ((1u << (3 * T)) - 1u)) == 0;
#ifdef WV_HOST_EMULATION
			++emul_group_any_count();
#endif
			i += nb;
			hook();
			return true;
		};
		// One pass.  has_b (a type: known where the pass is compiled): a second block follows in the run -- every pass but a
		// last single block, which gets a copy of the pass without the tests for it.
		auto pass = [&](auto has_b_t) __attribute__((always_inline)) {
			constexpr bool has_b = decltype(has_b_t)::value;
			const uint8_t* a = src + (uint64_t)i * bs;
			const uint8_t* b = a + bs;
			WV_MARK("load_block");
			uint32_t nblk = 1;
			uint32_t lzq = 0; // blocks of the batch that try the mini-LZ (block_compress.h:1210-1221)
			{
				// both blocks are requested at once, straight into registers; when the second one is not paired after all,
				// its load has at least brought it closer for the next round
				const RawBlock ea = load_raw_block(a, T);
				RawBlock eb, ec, ed;
				if (has_b)
					eb = load_raw_block(b, T);
				const bool early = has_b && i < wide_ahead;
				if (early) {
					ec = load_raw_block(b + bs, T);
					ed = load_raw_block(b + 2 * bs, T);
				}
				wide_ahead = 0;
				WV_MARK("block_begin");
				const Layout M = sink.at(L);
				SlotBatch B;
				uint32_t keys0 = 0, keys1 = 0; // distinct hash keys among the first 40 values of each block (first rejection test of the mini-LZ)
				{
					const SameScan sa = scan_same_fast(ea, T);
					B.act[0] = sa.act;
					B.first[0] = sa.first;
					B.nact0 = B.nslots = sa.nact;
					B.act[1] = B.first[1] = 0;
				}
				if (has_b) {
					WV_NESTED();
					if (B.nact0 <= 2) {
						const SameScan sb = scan_same_fast(eb, T);
						if (B.nact0 + sb.nact <= 4) {
							B.act[1] = sb.act;
							B.first[1] = sb.first;
							B.nslots = B.nact0 + sb.nact;
							nblk = 2;
						}
					}
				}
								if (T == 4 && B.nact0 >= 2) // with fewer non-constant planes the block is too small for the mini-LZ (:1210)
					keys0 = lz_distinct_keys_fast<2>(lds, M, ea.e);
				if (T == 4 && nblk > 1 && B.nslots - B.nact0 >= 2)
					keys1 = lz_distinct_keys_fast<2>(lds, M, eb.e);
				write_slots_fast(lds, M, ea, T, B.act[0], 0);
				if (nblk > 1)
					write_slots_fast(lds, M, eb, T, B.act[1], B.nact0);
				B.nblk = nblk;
				// (the next four blocks probably look like these two: worth a try as a group, below)
				try_group = nblk == 2 && B.act[0] == B.act[1] && B.nact0 == 2;
				try_any = true;
				// Two blocks with at most one non-constant plane each leave slots free: the blocks behind them move in while
				// they have at most one such plane themselves (a wide batch, slot_codec.h).
				SlotBatch4 W;
				W.nblk = 0;
				if (has_b && B.nslots <= 2) { WV_NESTED(); if (nblk == 2 && B.nact0 <= 1 && B.nslots - B.nact0 <= 1 && i + 2 < nblocks) {
					const bool has_d = i + 3 < nblocks;
					if (!early) {
						ec = load_raw_block(b + bs, T);
						if (has_d)
							ed = load_raw_block(b + 2 * bs, T);
					}
					const SameScan sc = scan_same_fast(ec, T);
					if (sc.nact <= 1) {
						W.act[0] = B.act[0], W.act[1] = B.act[1], W.act[2] = sc.act, W.act[3] = 0;
						W.first[0] = B.first[0], W.first[1] = B.first[1], W.first[2] = sc.first, W.first[3] = 0;
						write_slots_fast(lds, M, ec, T, sc.act, B.nslots);
						W.nslots = B.nslots + sc.nact;
						W.nblk = 3;
						if (has_d) {
							const SameScan sd = scan_same_fast(ed, T);
							if (sd.nact <= 1) {
								W.act[3] = sd.act;
								W.first[3] = sd.first;
								write_slots_fast(lds, M, ed, T, sd.act, W.nslots);
								W.nslots += sd.nact;
								W.nblk = 4;
							}
						}
						nblk = W.nblk;
						wide_ahead = first_lane_value(nblocks > 3 ? nblocks - 3 : 0); // (i + 3 < nblocks; said to be uniform: it is assigned under a condition the compiler takes for divergent)
					}
				} }
				if (sink.raw_to) { // (measured only, probably a copy: the raw bytes go where the copy would put them)
					sink.raw(ea, T, i);
					if (nblk > 1)
						sink.raw(eb, T, i + 1);
					if (nblk > 2)
						sink.raw(ec, T, i + 2);
					if (nblk > 3)
						sink.raw(ed, T, i + 3);
				}
				wave_sync();
				if (W.nblk) {
					SlotRows R;
					if (W.nslots)
					{
						const bool tryw = proof0.want();
						const bool raww = slot_rows_analyse(lds, M, R, 0, W.nslots, tryw);
						if (tryw)
							proof0.tried(raww);
					}
					else {
						for (int k = 0; k < 4; ++k)
							R.sb[k] = R.sd[k] = U32(0u);
						R.hm = R.pm = R.ts = U32(0u);
						R.emitmin = R.eq = pred_all(false);
					}
					uint32_t base[4];
					U32 bbase;
					const SlotPlace4 P = slot_rows_place4(R, W, T, hs, sink.base(), base, &bbase);
					const uint32_t total = base[3] + (W.nblk > 3 ? hs + W.full[3] : 0u) - base[0];
					if (sink.writes) {
						image_reset_fixed(lds, M, T);
						slot_rows_emit4(lds, M, T, R, P, W, bbase, base);
					}
					sink.append(lds, M, total);
					i += nblk;
					hook();
					return;
				}
				SlotRows R;
				if (B.nslots)
				{
					const bool tryb = proof0.want();
					const bool rawb = slot_rows_analyse(lds, M, R, 0, B.nslots, tryb);
					if (tryb)
						proof0.tried(rawb);
				}
				else { // only constant planes: nothing to measure
					for (int k = 0; k < 4; ++k)
						R.sb[k] = R.sd[k] = U32(0u);
					R.hm = R.pm = R.ts = U32(0u);
					R.emitmin = R.eq = pred_all(false);
				}
				WV_MARK("plane_offsets");
				uint32_t p0;
				const U32 incl = slot_rows_sizes(R, B, T, &p0);
				if (T == 4) {
					// Those that pass the rejection tests.  The key counts cover the first 40 values (most blocks fail there); a block
					// they do not turn away is looked at again: first the test that turns noise and floats away (values that hardly
					// repeat), then the key count over all 80 values.
					// (constant indices: a loop over the blocks would send the batch's scalars through memory)
					if (B.full[0] * 3 > bs && lz_precheck_passes(T, keys0, B.full[0])) {
						const RawBlock e = load_raw_block(a, T);
						if (!lz_repeats_reject(lds, M, e.e, B.full[0]) && lz_precheck_passes(T, lz_distinct_keys_fast<4>(lds, M, e.e), B.full[0]))
							lzq |= 1u, lds_st32(lds, U32(M.plinfo), U32(B.full[0]), lane_id() == U32(0u));
					}
					if (nblk > 1 && B.full[1] * 3 > bs && lz_precheck_passes(T, keys1, B.full[1])) {
						const RawBlock e = load_raw_block(b, T);
						if (!lz_repeats_reject(lds, M, e.e, B.full[1]) && lz_precheck_passes(T, lz_distinct_keys_fast<4>(lds, M, e.e), B.full[1]))
							lzq |= 2u, lds_st32(lds, U32(M.plinfo + 4), U32(B.full[1]), lane_id() == U32(0u));
					}
				}
				if (!lzq) {
					const uint32_t base = sink.base(), size0 = hs + B.full[0], size1 = nblk > 1 ? hs + B.full[1] : 0u;
					if (sink.writes) { // (a pass that is only measured needs the sizes, not the places)
						// what the row lanes need to know about a pass of this shape: one table load -- it hits the first-level cache, the
						// shapes of a run hardly change, and asking for it in front of the analysis would cost a register there
						const SlotPlace P = slot_rows_place(R, B, T, shape_lane_entry(T, B.act[0] | (B.act[1] << 4)), incl, p0);
						WV_MARK("image_reset");
						image_reset_fixed(lds, M, T);
						slot_rows_emit(lds, M, T, R, P, B, base, base + size0);
					}
					WV_MARK("stream_append");
					sink.append(lds, M, size0 + size1);
					WV_MARK("block_end");
					i += nblk;
					hook();
					return;
				}
			}
			try_group = try_any = false; // (blocks the mini-LZ may take are encoded one at a time)
			// a mini-LZ attempt, one block at a time (the raw bytes of a measured superblock have been put in place already,
			// above).  The candidates' sizes without the mini-LZ are known from the pass -- it left them in the plane table's
			// place, so that nothing of this rare path is alive in the pass --: the attempt needs nothing else, and the general
			// encoder only runs for a block whose attempt fails or that was not a candidate.
			wave_sync();
			const uint32_t lzfull0 = readlane(lds_ld32(lds, U32(L.plinfo)), 0), lzfull1 = readlane(lds_ld32(lds, U32(L.plinfo + 4)), 0);
			for (uint32_t q = 0; q < nblk; ++q) {
				load_block(lds, L.in, q ? b : a, bs);
				wave_sync();
				const Layout M = sink.at(L);
				const uint32_t n = ((lzq >> q) & 1u) ? lz_try(lds, M, T, q ? lzfull1 : lzfull0, sink.base()) : 0u;
				if (n) {
					sink.append(lds, M, n + 1);
					continue;
				}
				const BlockInfo r = encode_full_block(lds, M, T, false, sink.base());
				sink.append(lds, M, r.size);
			}
			i += nblk;
		};
		while (i + 1 < nblocks) {
			if (try_group && i + 3 < nblocks) {
				WV_NESTED();
				if (group4())
					continue;
			}
			// (bytesoftype 2 only: for 32-bit elements the blocks of one shape have their groups of four, and a third copy of the
			// row-lane machinery in that kernel costs it 34 spilled registers -- measured: every int32 workload 10-25 % slower)
			if (T == 2 && try_any) {
				WV_NESTED();
				if (group_any())
					continue;
			}
			pass(std::true_type());
		}
		if (i < nblocks)
			pass(std::false_type());
		if (sink.group_hint)
			*sink.group_hint = try_group;
		return;
	}
	if (slots && T == 8) {
		// bytesoftype 8 (slot_codec.h): one block per batch, its non-constant planes in one or two passes of four slots.
		const uint32_t bs = 2048, hs = header_bytes(8);
		RawProofHint proof0, proof1; // (the block's first and second pass)
		// (requesting the next block as soon as the current one has left its registers for the slots was measured: +1 % on double
		// sine -- the eight registers it holds across the passes cost more than the wait it removes)
		for (uint32_t i = 0; i < nblocks; ++i) {
			const uint8_t* a = src + (uint64_t)i * bs;
			WV_MARK("load_block");
			const RawBlock8 eb = load_raw_block8(a);
			const Layout M = sink.at(L);
			WV_MARK("block_begin");
			const SameScan8 sc = scan_same8(eb);
			// distinct hash keys among the first 80 values: the mini-LZ's first rejection test (with fewer than three planes that are
			// not constant the block is too small for an attempt, block_compress.h:1210)
			const uint32_t keys = sc.nact >= 3 ? lz_distinct_keys_fast8(lds, M, eb) : 0u;
			if (sink.raw_to) // (measured only, probably a copy: the raw bytes go where the copy would put them)
				sink.raw8(eb, i);
			uint32_t mask0 = sc.act, mask1 = 0; // planes of the first pass: the four lowest set bits
			if (sc.nact > 4) {
				uint32_t m = sc.act;
				for (int q = 0; q < 4; ++q)
					m &= m - 1u;
				mask1 = m;
				mask0 = sc.act & ~m;
			}
			const uint32_t n0 = sc.nact > 4 ? 4u : sc.nact, n1 = sc.nact - n0;
			SlotRows R0, R1;
			U32 incl0(0u), incl1(0u);
			uint32_t pt0 = 0, pt1 = 0;
			if (n0) {
				write_slots8(lds, M, eb, mask0);
				wave_sync();
				const bool try0 = proof0.want();
				const bool raw0 = slot_rows_analyse(lds, M, R0, 0, n0, try0);
				if (try0)
					proof0.tried(raw0);
				incl0 = slot_pass_sizes(R0, n0, &pt0);
			}
			if (n1) {
				write_slots8(lds, M, eb, mask1);
				wave_sync();
				const bool try1 = proof1.want();
				const bool raw1 = slot_rows_analyse(lds, M, R1, 0, n1, try1);
				if (try1)
					proof1.tried(raw1);
				incl1 = slot_pass_sizes(R1, n1, &pt1);
			}
			const uint32_t full = pt0 + pt1 + (8 - sc.nact);
			if (full * 3 > bs && lz_precheck_passes(8, keys, full) && !lz_repeats_reject8(lds, M, eb, full)) { // a mini-LZ attempt: the general block encoder
				load_block(lds, L.in, a, bs);
				wave_sync();
				const uint32_t n = lz_try(lds, M, 8, full, sink.base()); // (its size without the mini-LZ is known: nothing else is needed)
				if (n) {
					sink.append(lds, M, n + 1);
					continue;
				}
				const BlockInfo r = encode_full_block(lds, M, 8, false, sink.base());
				sink.append(lds, M, r.size);
				continue;
			}
			if (sink.writes) {
				const uint32_t base = sink.base();
				image_reset(lds, M, base, hs + full);
				if (n0)
					slot_rows_emit8(lds, M, R0, shape_lane_entry8(sc.act, 0), sc, base, 0, incl0);
				if (n1)
					slot_rows_emit8(lds, M, R1, shape_lane_entry8(sc.act, 1), sc, base, pt0, incl1);
				if (sc.act == 0) { // only constant planes: their bytes (the type nibbles are all 0)
					const U32 lane = lane_id();
					const U32 byte = (sel(lane >= U32(4u), U32(sc.first_hi), U32(sc.first_lo)) >> ((lane & 3u) << 3)) & 0xFFu;
					put_small(lds + M.out, (U32(base + hs) + lane) * 8u, byte, lane < U32(8u), U32(slot2_area(M) - M.out) + lane * 16u);
					wave_sync();
				}
			}
			sink.append(lds, M, hs + full);
		}
		return;
	}
	for (uint32_t i = 0; i < nblocks; ++i) {
		load_block(lds, L.in, src + (uint64_t)i * (256 * T), 256 * T);
		wave_sync();
		if (sink.raw_to) // (measured only, probably a copy: the raw bytes go where the copy would put them)
			for (uint32_t o = 0; o < 256 * T; o += 1024) {
				const U32 off = umin(U32(o) + lane_id() * 16u, U32(256 * T - 16u)); // (lanes beyond the end repeat the last group)
				gst128_through(sink.raw_to + (uint64_t)i * (256 * T), off, lds_ld128(lds, U32(L.in) + off));
			}
		const Layout M = sink.at(L);
		const BlockInfo r = encode_full_block(lds, M, T, true, sink.base());
		sink.append(lds, M, r.size);
	}
}

// ... into a staging stream at stage (16-byte aligned, room for nblocks * max_block_bytes(T) + 16); returns the bytes
// (stage == nullptr: nothing is written, only the bytes are counted)
// raw_to (only without a stage): the blocks' raw bytes are stored there on the way (StreamSink::raw_to)
template <class Hook = NoPassHook>
WV_FN uint32_t encode_run(Lds lds, const Layout& L, uint32_t T, const uint8_t* src, uint32_t nblocks, uint8_t* stage, bool slots = true, Hook hook = Hook(),
			  uint8_t* raw_to = nullptr, bool* group_hint = nullptr)
{
	StreamSink sink;
	sink.group_hint = group_hint;
	sink.rs.base = stage;
	sink.rs.pos = 0;
	sink.writes = stage != nullptr;
	sink.raw_to = stage ? nullptr : raw_to;
	if (sink.writes)
		stream_begin(lds, L.out);
	encode_blocks_to(sink, lds, L, T, src, nblocks, slots, hook);
	if (sink.writes)
		stream_flush(sink.rs, lds, L.out);
	return sink.rs.pos;
}
// HBM -> HBM copy of n bytes by one wave: the destination is written in aligned 16-byte groups (bytes in front of the
// first and behind the last), the source read with 16-byte loads from whatever byte address that makes -- gfx9 and later
// serve unaligned global accesses in hardware, so no lane reads a byte outside [src, src + n) and nothing has to be
// shifted together from two aligned groups.  COPY_ROUNDS rounds of 64 groups at a time: all their loads are requested
// before the first store, so the rounds cost one memory round trip.
constexpr uint32_t COPY_ROUNDS = 4;
#ifdef WV_PREDICATE_BRANCHES
template <uint32_t ROUNDS = COPY_ROUNDS>
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
	uint8_t* d = dst + h;
	const uint8_t* s = src + h;
	for (uint32_t o = 0; o < groups; o += 64 * ROUNDS) {
		U128 a[ROUNDS];
		for (uint32_t q = 0; q < ROUNDS; ++q) {
			const U32 k = U32(o + 64 * q) + lane;
			a[q] = gld128_unaligned(s, k * 16u, k < U32(groups));
		}
		for (uint32_t q = 0; q < ROUNDS; ++q) {
			const U32 k = U32(o + 64 * q) + lane;
			gst128(d, k * 16u, a[q], k < U32(groups));
		}
	}
	const uint32_t done = h + groups * 16;
	{
		Pred p = lane < U32(n - done);
		gst8(dst + done, lane, gld8(src + done, lane, p), p);
	}
}

#else
template <uint32_t ROUNDS = COPY_ROUNDS>
WV_FN void copy_g2g_wide(uint8_t* dst, const uint8_t* src, uint32_t n)
{
	const U32 lane = lane_id();
	const uint32_t head = (uint32_t)((16u - ((uintptr_t)dst & 15u)) & 15u);
	const uint32_t h = head < n ? head : n;
	if (h) // (loads: lanes beyond the end repeat the last byte / group, see copy_g2l; stores: predicated, not waited for)
		gst8_streamed(dst, lane, gld8(src, umin(lane, U32(h - 1u)), pred_all(true)), lane < U32(h));
	const uint32_t groups = (n - h) >> 4;
	uint8_t* d = dst + h;
	const uint8_t* s = src + h;
	for (uint32_t o = 0; o < groups; o += 64 * ROUNDS) {
		U128 a[ROUNDS];
		for (uint32_t q = 0; q < ROUNDS; ++q) // (a round that lies beyond the end altogether is a scalar branch away)
			if (o + 64 * q < groups) {
				const U32 k = umin(U32(o + 64 * q) + lane, U32(groups - 1u));
				a[q] = gld128_unaligned(s, k * 16u, pred_all(true));
			}
		for (uint32_t q = 0; q < ROUNDS; ++q)
			if (o + 64 * q < groups) {
				const U32 k = U32(o + 64 * q) + lane;
				gst128_streamed(d, k * 16u, a[q], k < U32(groups));
			}
	}
	const uint32_t done = h + groups * 16;
	if (n - done)
		gst8_streamed(dst + done, lane, gld8(src + done, umin(lane, U32(n - done - 1u)), pred_all(true)), lane < U32(n - done));
}

#endif
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
	r.info = 0;
	r.need = need;
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
// lz_decode_256 (block_codec.h) packs a group's window offset into 16 bits and keeps a 256-byte table behind the image's
// first 256 * T bytes, in the slack make_dec_layout leaves there: both hold for the element sizes that use it (4 and 8), and
// a change to the window or to the reach of a block that broke them would fail here, not corrupt a table.
static_assert(256 * 4 + 32 <= 65535 && 256 * 8 + 32 <= 65535, "");
WV_HD constexpr uint32_t dec_image_slack(uint32_t T) // bytes between the decoded block (256 * T) and the tables behind the image
{
	return ((256 * T + 32 > ((T + 1) / 2 + T * 314 + 32) ? 256 * T + 32 : ((T + 1) / 2 + T * 314 + 32)) + 15) / 16 * 16 - 256 * T;
}
static_assert(dec_image_slack(4) >= 256 && dec_image_slack(8) >= 256, "lz_decode_256 keeps a 256-byte table behind the decoded block");
WV_HD DecLayout make_dec_layout(uint32_t T)
{
	DecLayout L;
	L.win = 0;
	L.img = window_bytes(T) + 32;
	const uint32_t after = 256 * T + 32;
	L.lut = align16(L.img + (after > max_block_reach(T) ? after : max_block_reach(T)));
	L.total = L.lut + 64 + 128; // (the two tables of dec_write_lut)
	return L;
}

// Decode the payload of one BLOCK superblock (code 1): `csize` compressed bytes at src -> `dsize`
// bytes at dst.  Returns dsize, or DEC_ERROR on a malformed / truncated stream.
// regs: T is 2, 4 or 8 and known at compile time: blocks made of planes go to HBM from registers (decode_planes_to)
WV_FN uint32_t decode_superblock(Lds lds, const DecLayout& L, uint32_t T, const uint8_t* src, uint32_t csize, uint8_t* dst, uint32_t dsize, bool regs = false)
{
	const U32 lane = lane_id_plain();
	const uint32_t bs = 256 * T, hs = header_bytes(T);
	if (dsize == 0 || csize == 0)
		return 0;
	const uint32_t nblocks = dsize / bs;
	if (csize < hs + T && nblocks) // block_compress.h:1813-1815
		return DEC_ERROR;
	dec_write_lut(lds, L);
	const uint32_t wcap = window_bytes(T);
	const uint32_t mis = (uint32_t)((uintptr_t)src & 15u); // the window is filled from the 16-byte aligned address below src
	const uint8_t* abase = src - mis;
	uint32_t wstart = 0, wfill = 0, wend = 0; // window holds abase[wstart, wend), wend = wstart + wfill
	uint32_t consumed = 0;          // payload bytes consumed so far
	const bool to_hbm = regs;

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

	uint8_t* to = dst; // where the next block goes
	for (uint32_t b = 0; b < nblocks; ++b) {
		WV_MARK("dec_block_begin");
		uint32_t left = csize - consumed;
		uint32_t need = left < max_block_bytes(T) ? left : max_block_bytes(T);
		ensure(need);
		WV_MARK("dec_block");
		bool direct = false;
		uint32_t n = decode_block(lds, L, T, consumed + mis - wstart, need, 16, true, to_hbm ? to : nullptr, to_hbm ? &direct : nullptr);
		if (n == DEC_ERROR)
			return DEC_ERROR;
		WV_MARK("dec_block_store");
		if (!to_hbm) { // (known at compile time: with a place in HBM the block is stored when decode_block returns)
			store_block(to, lds, L.img, bs);
			wave_sync();
		}
		consumed += n;
		to += bs;
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
