// pipeline.h -- frame assembly around the block encoder, one wavefront per unit of work:
//   plan_superblock   payload size, BLOCK/COPY decision and capacity requirement of one superblock
//   resolve_capacity  exact replay of the reference's capacity rules for the superblocks the parallel
//                     plan could not clear (normally none, or the last one)
//   pack_block        copy one block (or its raw bytes) and the headers into the frame
// Written in the wavevec.h vocabulary so the host emulation runs the same logic.
//
// Capacity rules being reproduced (serial path of the reference): superblock k is compressed into
// what is left of the caller's buffer, C_k = dst_size - offset_k - 4 (stenos.cpp:895, 608); inside it
// the mini-LZ is only tried when C_k > d + full + 8*T + 2 (block_compress.h:1214) and the block loop
// gives up (-> COPY, stenos.cpp:609-610) when a "dst + x > dst_end" test fails (:1225, 1241, 1284,
// block_compress_partial :984, 994, 1013); COPY itself needs C_k >= bytes (stenos.cpp:366-367).
// The blocks are first encoded with the LZ allowed; plan_superblock derives the smallest C for
// which that result is what the reference would produce and compares it with a lower bound of C_k
// that needs no prefix sum.  Only superblocks that fail this test are replayed sequentially.
#pragma once
#include "superblock_codec.h"

namespace codec {

enum : uint32_t { ENCODE_STATUS_DST_OVERFLOW = 1, ENCODE_STATUS_CHAIN_TIMEOUT = 2 };

struct FrameJob {
	const uint8_t* src;
	uint8_t* dst;
	uint64_t dst_size; // logical capacity of dst (the caller's dst_size)
	uint8_t* slots;
	uint32_t* bsize;   // encoded size of every block (tail block last)
	uint32_t* binfo;   // BlockInfo::info of every block
	uint32_t* bneed;   // BlockInfo::need of every block
	uint32_t* boff;    // offset of every block inside its superblock payload
	uint32_t* sb_csize;
	uint8_t* sb_code;
	uint32_t* sb_need; // capacity requirement per superblock (0: none)
	uint64_t* sb_off;  // nsb + 1 header offsets inside the frame
	uint64_t* total;
	uint32_t* status;
	uint32_t* first_flagged; // smallest superblock index whose capacity test failed (0xFFFFFFFF: none)
	const uint8_t* override_payload;
	uint64_t nfull;       // full blocks in the input
	uint64_t nsb;         // superblocks
	uint64_t total_bytes; // input bytes
	uint32_t tail_bytes;  // bytes of the trailing partial block (0: none)
	uint32_t bps;         // full blocks per full superblock
	uint32_t sb_bytes;
	uint32_t slot_stride;
	uint32_t T;
	uint32_t shift_byte;   // frame byte 0; 0xFFFFFFFF: no frame header (private single-superblock API)
	uint32_t header_bytes; // frame header bytes (0, 8 or 12)
	uint32_t force_copy;   // level 0
	uint32_t tiny_last;    // the last superblock is shorter than 128 bytes: finished by the host (zstd)
	uint32_t override_code; // set by the host for the tiny last superblock before pack
	uint32_t check_total;   // pack: compare *total with dst_size first (not needed where the plan proved that everything fits)
	// levels >= 2 / bytesoftype 1 (stenos.cpp:546-547): every superblock is block-compressed into a buffer of exactly
	// its own size, so the capacity rules need no prefix sum and are replayed inside plan_superblock; superblocks
	// whose block stream fails or exceeds the input get code 0 / csize 0 (the host then takes the zstd strategies),
	// and qprod[s] receives the bytes produced when the reference checks its target ratio (block_compress.h:1267)
	uint32_t fixed_capacity;
	uint32_t* qprod;
	// bytesoftype above 64 (kernels_wide.hip): device scratch the workgroups share out, stenos_kw_scratch_stride(T) bytes each
	uint8_t* wide_scratch;
	uint64_t wide_scratch_bytes;
};

WV_HD uint32_t superblock_bytes(const FrameJob& j, uint64_t s)
{
	const uint64_t begin = s * (uint64_t)j.sb_bytes;
	return (uint32_t)((j.total_bytes - begin) < j.sb_bytes ? (j.total_bytes - begin) : j.sb_bytes);
}
WV_HD uint32_t superblock_blocks(const FrameJob& j, uint64_t s) // full blocks + the tail block
{
	const uint64_t first = s * j.bps;
	const uint64_t last = first + j.bps < j.nfull ? first + j.bps : j.nfull;
	return (uint32_t)(last - first) + ((s == j.nsb - 1 && j.tail_bytes) ? 1u : 0u);
}

// capacity a block needs, counted from the start of the payload, for its fast-path encoding to be
// what the reference produces; a = offset of the block in the payload
WV_FN U32 block_requirement(const U32& a, const U32& info, const U32& need, const Pred& is_tail, uint32_t T)
{
	U32 full = info & 0x3FFFFFFFu;
	Pred lz = ((info >> 31) & 1u) == U32(1u);
	U32 lzreq = a + U32(header_bytes(T)) + full + U32(8u * T + 3u); // C > d + full + 8T + 2
	return sel(!is_tail & lz, lzreq, a + need);
}

WV_FN bool replay_superblock(Lds lds, const Layout& L, const FrameJob& j, uint64_t s, uint64_t C, uint32_t* payload);

WV_FN void plan_superblock(Lds lds, const Layout& L, const FrameJob& j, uint64_t s)
{
	const U32 lane = lane_id();
	const uint64_t first = s * j.bps;
	const uint32_t sbytes = superblock_bytes(j, s);
	const uint32_t count = superblock_blocks(j, s);
	const bool has_tail = s == j.nsb - 1 && j.tail_bytes;
	uint32_t run = 0, need = 0;
	if (!j.force_copy)
		for (uint32_t o = 0; o < count; o += 64) {
			U32 i = U32(o) + lane;
			Pred p = i < U32(count);
			U32 sz = gld32((const uint8_t*)(j.bsize + first), i * 4u, p);
			U32 info = gld32((const uint8_t*)(j.binfo + first), i * 4u, p);
			U32 bneed = gld32((const uint8_t*)(j.bneed + first), i * 4u, p);
			U32 incl = wave_incl_scan(sz);
			U32 a = U32(run) + incl - sz;
			gst32((uint8_t*)(j.boff + first), i * 4u, a, p);
			Pred is_tail = pred_all(has_tail) & (i == U32(count - 1));
			uint32_t m = wave_max(sel(p, block_requirement(a, info, bneed, is_tail, j.T), U32(0u)));
			need = m > need ? m : need;
			run += readlane(incl, 63);
		}
	uint32_t code = 1, csize = run;
	if (j.force_copy || run > sbytes) { // result > bytes -> memcpy (stenos.cpp:609-610); equal is kept
		code = 6;
		csize = sbytes;
		need = sbytes; // a copy stays a copy under any capacity, it only has to fit (stenos.cpp:366-367)
	}
	if (j.tiny_last && s == j.nsb - 1) { // zstd / copy decided by the host, which also checks the capacity
		code = 6;
		csize = sbytes;
		need = 0;
	}
	if (j.fixed_capacity) {
		if (code == 1 && need > sbytes) { // capacity = the superblock's own size: exact replay right here
			uint32_t payload = 0;
			if (replay_superblock(lds, L, j, s, sbytes, &payload) && payload <= sbytes)
				csize = payload;
			else
				code = 6;
		}
		if (code != 1) {
			code = 0;
			csize = 0;
		}
		// bytes produced once 1/16 of the input is consumed (full blocks only)
		const uint32_t bs = 256 * j.T;
		const uint32_t nfullb = count - (has_tail ? 1u : 0u);
		uint32_t q = 0;
		if (code == 1 && nfullb) {
			uint32_t bq = (sbytes / 16 + bs - 1) / bs; // smallest b with b*bs >= sbytes/16, i.e. block index bq-1
			bq = bq == 0 ? 0 : bq - 1;
			if (bq < nfullb)
				q = gload_uniform(j.boff + first + bq) + gload_uniform(j.bsize + first + bq);
		}
		gstore_uniform(j.qprod + s, q);
		gst8(j.sb_code + s, lane, U32(code), lane == U32(0u));
		gst32((uint8_t*)(j.sb_csize + s), U32(0u), U32(csize), lane == U32(0u));
		gst32((uint8_t*)(j.sb_need + s), U32(0u), U32(0u), lane == U32(0u));
		return;
	}
	// lower bound of this superblock's capacity: every earlier superblock stored as a copy
	const uint64_t worst_off = (uint64_t)j.header_bytes + s * ((uint64_t)j.sb_bytes + 4) + 4;
	const bool flagged = j.dst_size < worst_off || j.dst_size - worst_off < need;
	gst8(j.sb_code + s, lane, U32(code), lane == U32(0u));
	gst32((uint8_t*)(j.sb_csize + s), U32(0u), U32(csize), lane == U32(0u));
	gst32((uint8_t*)(j.sb_need + s), U32(0u), U32(need), lane == U32(0u));
	if (flagged)
		gmin32(j.first_flagged, (uint32_t)s);
}

// Exact replay for one superblock whose capacity C is below its requirement.  Returns false when the
// block loop fails (the superblock becomes a copy).  *payload receives the new payload size.
WV_FN bool replay_superblock(Lds lds, const Layout& L, const FrameJob& j, uint64_t s, uint64_t C, uint32_t* payload)
{
	const uint64_t first = s * j.bps;
	const uint32_t count = superblock_blocks(j, s);
	const bool has_tail = s == j.nsb - 1 && j.tail_bytes;
	const uint32_t T = j.T, hs = header_bytes(T), bs = 256 * T;
	uint64_t a = 0;
	for (uint32_t i = 0; i < count; ++i) {
		const uint64_t b = first + i;
		uint32_t info = gload_uniform(j.binfo + b);
		uint32_t bneed = gload_uniform(j.bneed + b);
		uint32_t size = gload_uniform(j.bsize + b);
		const bool is_tail = has_tail && i == count - 1;
		if (!is_tail && info_lz_ok(info) && !(C > a + hs + info_full(info) + 8 * T + 2)) {
			// the reference would not have tried the LZ here (block_compress.h:1214): encode the planes instead
			BlockInfo r = encode_block_job(lds, L, T, j.src + b * (uint64_t)bs, j.slots + b * (uint64_t)j.slot_stride, false);
			size = r.size;
			info = r.info;
			bneed = r.need;
			gstore_uniform(j.bsize + b, size);
			gstore_uniform(j.binfo + b, info);
			gstore_uniform(j.bneed + b, bneed);
		}
		const bool kept_lz = !is_tail && info_lz_ok(info);
		if (!kept_lz && a + bneed > C) // one of the reference's dst_end tests fails
			return false;
		gstore_uniform(j.boff + b, (uint32_t)a);
		a += size;
	}
	*payload = (uint32_t)a;
	return true;
}

// One wavefront walks the superblocks from the first flagged one to the end with their exact offsets.
WV_FN void resolve_capacity(Lds lds, const Layout& L, const FrameJob& j)
{
	const uint32_t k0 = gload_uniform(j.first_flagged);
	if (k0 >= j.nsb)
		return;
	uint64_t off = gload_uniform64(j.sb_off + k0);
	for (uint64_t s = k0; s < j.nsb; ++s) {
		if (j.dst_size < off + 4) { // no room for the superblock header (stenos.cpp:427-429)
			gstore_uniform(j.status, ENCODE_STATUS_DST_OVERFLOW);
			return;
		}
		const uint64_t C = j.dst_size - off - 4;
		const uint32_t sbytes = superblock_bytes(j, s);
		uint32_t code = gload_uniform8(j.sb_code + s);
		uint32_t csize = gload_uniform(j.sb_csize + s);
		const uint32_t need = gload_uniform(j.sb_need + s);
		const bool tiny = j.tiny_last && s == j.nsb - 1;
		if (!tiny && code == 1 && C < need) {
			uint32_t payload = 0;
			if (replay_superblock(lds, L, j, s, C, &payload) && payload <= sbytes)
				csize = payload;
			else {
				code = 6;
				csize = sbytes;
			}
			gstore_uniform8(j.sb_code + s, code);
			gstore_uniform(j.sb_csize + s, csize);
		}
		if (!tiny && code == 6 && C < sbytes) { // compress_memcpy: dst_size < bytes + 4 (stenos.cpp:366-367)
			gstore_uniform(j.status, ENCODE_STATUS_DST_OVERFLOW);
			return;
		}
		gstore_uniform64(j.sb_off + s, off);
		off += 4 + (uint64_t)csize;
	}
	gstore_uniform64(j.sb_off + j.nsb, off);
	gstore_uniform64(j.total, off);
}

// ---- fused path: superblocks made of full blocks that certainly have room -------------------------
//
// FUSED_WAVES wavefronts of one workgroup share a superblock: each encodes a run of consecutive blocks into a
// contiguous staging stream (encode_run), the workgroup learns the superblock's frame offset (sb_off[s], written
// by the scanner wavefront of kernels.hip) and every wave copies its own run to its place in the frame.  No
// per-block table and no separate pack pass.
constexpr uint32_t FUSED_WAVES = 4;
WV_HD uint32_t fused_run_blocks(uint32_t bps) { return (bps + FUSED_WAVES - 1) / FUSED_WAVES; }
WV_HD uint32_t fused_run_capacity(uint32_t bps, uint32_t T) { return align16(fused_run_blocks(bps) * max_block_bytes(T)) + 64; }

// Number of leading superblocks whose capacity is large enough for any encoding, whatever the sizes of the
// superblocks before them (every one stored as a copy): none of the reference's dst_end tests can fire there.
WV_HD uint64_t safe_superblocks(uint64_t dst_size, uint64_t header, uint32_t bps, uint32_t T, uint64_t sb_bytes, uint64_t nsb)
{
	const uint64_t need_max = (uint64_t)(bps + 1) * (256 * T + (T + 1) / 2) + 288 * T + 64;
	const uint64_t fixed = header + 4 + need_max;
	uint64_t n = dst_size >= fixed ? (dst_size - fixed) / (sb_bytes + 4) + 1 : 0;
	return n > nsb ? nsb : n;
}

// blocks [*b0, *b1) of a superblock are wave w's run
WV_HD void fused_run_range(uint32_t bps, uint32_t w, uint32_t* b0, uint32_t* b1)
{
	const uint32_t per = fused_run_blocks(bps);
	*b0 = w * per < bps ? w * per : bps;
	*b1 = *b0 + per < bps ? *b0 + per : bps;
}

// BLOCK or COPY and the bytes the superblock takes in the frame (header included), from the run sizes
WV_HD uint32_t fused_superblock_size(const FrameJob& j, const uint32_t* run_size, uint32_t* code)
{
	uint32_t csize = 0;
	for (uint32_t w = 0; w < FUSED_WAVES; ++w)
		csize += run_size[w];
	*code = csize > j.sb_bytes ? 6u : 1u; // result > bytes -> memcpy (stenos.cpp:609-610); equal is kept
	return 4 + (*code == 1 ? csize : j.sb_bytes);
}

// wave w's share of writing superblock s at frame offset off.  payload_in_place: a copy whose raw bytes stand there already
// (kernels.hip, speculative copy): only the headers are left to write.
WV_FN void fused_store(const FrameJob& j, uint64_t s, uint32_t w, uint64_t off, const uint32_t* run_size, const uint8_t* stage_w, bool payload_in_place = false)
{
	const U32 lane = lane_id();
	uint32_t code;
	const uint32_t csize = fused_superblock_size(j, run_size, &code) - 4;
	uint8_t* base = j.dst + off;
	if (w == 0) {
		if (s == 0 && j.shift_byte != 0xFFFFFFFFu) { // frame header: [shift][bytes:7 LE] (+ [superblock size:4 LE]), stenos.cpp:862-874
			const uint64_t v = (uint64_t)j.shift_byte | (j.total_bytes << 8);
			gst8(j.dst, lane, (U32((uint32_t)v) >> ((lane & 3u) << 3)), lane < U32(4u));
			gst8(j.dst, lane, (U32((uint32_t)(v >> 32)) >> ((lane & 3u) << 3)), (lane >= U32(4u)) & (lane < U32(8u)));
			if (j.shift_byte == 255)
				gst8(j.dst + 8, lane, U32(j.sb_bytes) >> ((lane & 3u) << 3), lane < U32(4u));
		}
		// superblock header [code][csize:3 LE] (stenos.cpp:613-615)
		gst8(base, lane, U32(code | (csize << 8)) >> ((lane & 3u) << 3), lane < U32(4u));
	}
	if (code == 1) {
		uint32_t before = 0;
		for (uint32_t k = 0; k < w; ++k)
			before += run_size[k];
		copy_g2g_wide(base + 4 + before, stage_w, run_size[w]);
	}
	else if (!payload_in_place) {
		uint32_t b0, b1;
		fused_run_range(j.bps, w, &b0, &b1);
		const uint32_t bs = 256 * j.T;
		copy_g2g_wide(base + 4 + (uint64_t)b0 * bs, j.src + (s * j.bps + b0) * (uint64_t)bs, (b1 - b0) * bs);
	}
}

// Pack step: PACK_WAVES wavefronts per superblock, each writing one contiguous quarter of the superblock's
// bytes in the frame.  The work is laid out over the DESTINATION: every lane produces aligned destination
// dwords (fully coalesced stores) and finds the block slot its bytes come from with a short forward scan
// of the block-offset table held in the wave's LDS, so all loads of an iteration are independent.
constexpr uint32_t PACK_WAVES = 4;
WV_HD uint32_t pack_lds_bytes(uint32_t bps) { return align16((bps + 3) * 4) + 16; }

// source byte p of the payload of a BLOCK superblock: block bi holds payload bytes [off[bi], off[bi+1])
WV_FN U32 pack_src_byte(const FrameJob& j, Lds tab, uint64_t first, const U32& p, U32 bi, uint32_t count, const Pred& valid)
{
	// advance to the block that contains p (blocks are at least header_bytes + T bytes long)
	for (;;) {
		Pred adv = valid & (bi + 1u < U32(count)) & (p >= lds_ld32(tab, (bi + 1u) * 4u));
		if (!any(adv))
			break;
		bi = sel(adv, bi + 1u, bi);
	}
	U32 rel = p - lds_ld32(tab, bi * 4u);
	return gld8(j.slots + first * (uint64_t)j.slot_stride, bi * j.slot_stride + rel, valid);
}

WV_FN void pack_superblock(Lds lds, const FrameJob& j, uint64_t s, uint32_t w)
{
	const U32 lane = lane_id();
	if (gload_uniform(j.status))
		return;
	if (j.check_total && gload_uniform64(j.total) > j.dst_size) // never write past the caller's buffer; the host reports DST_OVERFLOW
		return;
	const uint32_t count = superblock_blocks(j, s);
	const uint64_t first = s * j.bps;
	const bool has_tail = s == j.nsb - 1 && j.tail_bytes;
	uint8_t* base = j.dst + gload_uniform64(j.sb_off + s);
	const uint32_t code = gload_uniform8(j.sb_code + s);
	const uint32_t csize = gload_uniform(j.sb_csize + s);
	if (s == 0 && w == 0 && j.shift_byte != 0xFFFFFFFFu) { // frame header: [shift][bytes:7 LE] (+ [superblock size:4 LE]), stenos.cpp:862-874
		const uint64_t v = (uint64_t)j.shift_byte | (j.total_bytes << 8);
		gst8(j.dst, lane, (U32((uint32_t)v) >> ((lane & 3u) << 3)), lane < U32(4u));
		gst8(j.dst, lane, (U32((uint32_t)(v >> 32)) >> ((lane & 3u) << 3)), (lane >= U32(4u)) & (lane < U32(8u)));
		if (j.shift_byte == 255)
			gst8(j.dst + 8, lane, U32(j.sb_bytes) >> ((lane & 3u) << 3), lane < U32(4u));
	}
	if (w == 0) // superblock header [code][csize:3 LE] (stenos.cpp:613-615)
		gst8(base, lane, U32(code | (csize << 8)) >> ((lane & 3u) << 3), lane < U32(4u));
	if (code == 0) // levels >= 2: no block stream for this superblock (the host uses a zstd strategy)
		return;
	uint8_t* pay = base + 4;
	// this wave's share of the payload, cut at destination dword boundaries
	const uint32_t mis = (uint32_t)((uintptr_t)pay & 3u);
	const uint32_t ndw = (mis + csize + 3) >> 2;                  // aligned destination dwords touched by the payload
	const uint32_t per = (ndw + PACK_WAVES - 1) / PACK_WAVES;
	const uint32_t d0 = w * per < ndw ? w * per : ndw, d1 = d0 + per < ndw ? d0 + per : ndw;
	if (d0 >= d1)
		return;
	if ((j.tiny_last && s == j.nsb - 1) || code != 1) {
		// one contiguous source: the host-prepared payload, or the raw input bytes of a COPY superblock
		const uint8_t* srcp = (j.tiny_last && s == j.nsb - 1) ? j.override_payload : j.src + first * (uint64_t)(256 * j.T);
		const uint32_t b0 = d0 * 4 > mis ? d0 * 4 - mis : 0, b1 = d1 * 4 - mis < csize ? d1 * 4 - mis : csize;
		copy_g2g_wide(pay + b0, srcp + b0, b1 - b0);
		return;
	}
	(void)has_tail;
	// block offsets of this superblock -> LDS (count + 1 entries, the last one is the payload size)
	for (uint32_t o = 0; o < count; o += 64) {
		Pred p = (U32(o) + lane) < U32(count);
		lds_st32(lds, (U32(o) + lane) * 4u, gld32((const uint8_t*)(j.boff + first), (U32(o) + lane) * 4u, p), p);
	}
	lds_st32(lds, U32(count * 4u), U32(csize), lane == U32(0u));
	wave_sync();
	// Lanes own 16 consecutive destination bytes per iteration (one 16-byte store, five independent aligned
	// loads).  Groups that lie inside one block take that path; groups cut by a block boundary or by the
	// ends of the payload fall back to dwords, and dwords cut the same way to bytes.
	const uint32_t mis16 = (uint32_t)((uintptr_t)pay & 15u);
	uint8_t* abase = pay - mis16; // 16-byte aligned
	const uint32_t b0 = d0 * 4 > mis ? d0 * 4 - mis : 0;                       // payload bytes [b0, b1) belong to this wave
	const uint32_t b1 = d1 * 4 - mis < csize ? d1 * 4 - mis : csize;
	const uint32_t g0 = (mis16 + b0) >> 4, g1 = (mis16 + b1 + 15) >> 4;        // 16-byte groups touched
	uint32_t lo = 0;
	for (uint32_t o = 0; o < count; o += 64) { // block that holds payload byte b0
		Pred in = (U32(o) + lane) < U32(count);
		uint64_t m = ballot(in & (lds_ld32(lds, (U32(o) + lane + 1u) * 4u) > U32(b0)));
		if (m) {
			lo = o + (uint32_t)__builtin_ctzll(m);
			break;
		}
	}
	const uint8_t* slot0 = j.slots + first * (uint64_t)j.slot_stride;
	const uint32_t steps = 1; // consecutive bytes: blocks are never empty, so the block index grows by at most 1 per byte
	for (uint32_t g = g0; g < g1; g += 64) {
		U32 gi = U32(g) + lane;
		Pred act = gi < U32(g1);
		// payload range of this group clipped to the wave's share: [q0, q1)
		U32 gp = gi * 16u;                                       // offset from abase
		U32 q0 = umax(gp, U32(mis16 + b0)) - U32(mis16);
		U32 q1 = umin(gp + 16u, U32(mis16 + b1)) - U32(mis16);
		Pred whole = act & (q1 - q0 == U32(16u));
		U32 bi(lo);
		for (;;) { // forward scan to the block of byte q0
			Pred adv = act & (bi + 1u < U32(count)) & (q0 >= lds_ld32(lds, (bi + 1u) * 4u));
			if (!any(adv))
				break;
			bi = sel(adv, bi + 1u, bi);
		}
		U32 bstart = lds_ld32(lds, bi * 4u), bend = lds_ld32(lds, (bi + 1u) * 4u);
		Pred fast = whole & (q0 + 16u <= bend);
		{
			// two aligned 16-byte loads cover the 16 source bytes (slots are 16-byte aligned and padded), then a
			// byte funnel: two wide requests per lane instead of five dword requests
			U32 soff = bi * j.slot_stride + (q0 - bstart);
			U32 sa = soff & ~15u;
			U128 a = gld128(slot0, sa, fast), b = gld128(slot0, sa + 16u, fast & ((soff & 15u) != U32(0u)));
			U32 wsel = (soff >> 2) & 3u; // first source dword inside the 32-byte window
			U32 sh = (soff & 3u) << 3, ish = U32(32u) - sh;
			// W[wsel .. wsel+4] out of {a.x a.y a.z a.w b.x b.y b.z b.w}
			U32 t0 = sel(wsel == U32(0u), a.x, sel(wsel == U32(1u), a.y, sel(wsel == U32(2u), a.z, a.w)));
			U32 t1 = sel(wsel == U32(0u), a.y, sel(wsel == U32(1u), a.z, sel(wsel == U32(2u), a.w, b.x)));
			U32 t2 = sel(wsel == U32(0u), a.z, sel(wsel == U32(1u), a.w, sel(wsel == U32(2u), b.x, b.y)));
			U32 t3 = sel(wsel == U32(0u), a.w, sel(wsel == U32(1u), b.x, sel(wsel == U32(2u), b.y, b.z)));
			U32 t4 = sel(wsel == U32(0u), b.x, sel(wsel == U32(1u), b.y, sel(wsel == U32(2u), b.z, b.w)));
			Pred al = sh == U32(0u);
			U128 v;
			v.x = sel(al, t0, (t0 >> sh) | (t1 << ish));
			v.y = sel(al, t1, (t1 >> sh) | (t2 << ish));
			v.z = sel(al, t2, (t2 >> sh) | (t3 << ish));
			v.w = sel(al, t3, (t3 >> sh) | (t4 << ish));
			gst128(abase, gp, v, fast);
		}
		// groups cut by a block boundary or by the ends of this wave's share: byte by byte, but with all
		// table look-ups and loads of the 16 bytes issued before the first store (no dependent round trips)
		Pred slow = act & !fast;
		if (any(slow)) {
			U32 vals[16];
			U32 bk = bi;
			for (uint32_t c = 0; c < 16; ++c) {
				U32 ap = gp + U32(c); // offset from abase
				Pred ok = slow & (ap >= U32(mis16 + b0)) & (ap < U32(mis16 + b1));
				U32 p = sel(ok, ap - U32(mis16), q0);
				// the block of byte p is at most `steps` blocks after the block of the previous byte
				for (uint32_t st = 0; st < steps; ++st) {
					Pred adv = ok & (bk + 1u < U32(count)) & (p >= lds_ld32(lds, (bk + 1u) * 4u));
					bk = sel(adv, bk + 1u, bk);
				}
				vals[c] = gld8(slot0, bk * j.slot_stride + (p - lds_ld32(lds, bk * 4u)), ok);
			}
			for (uint32_t c = 0; c < 16; ++c) {
				U32 ap = gp + U32(c);
				Pred ok = slow & (ap >= U32(mis16 + b0)) & (ap < U32(mis16 + b1));
				gst8(abase, ap, vals[c], ok);
			}
		}
		lo = readlane(bi, 0); // the next iteration starts at or after this lane's block
	}
}

} // namespace codec
