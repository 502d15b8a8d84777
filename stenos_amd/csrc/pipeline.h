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

enum : uint32_t { ENCODE_STATUS_DST_OVERFLOW = 1 };

struct FrameJob {
	const uint8_t* src;
	uint8_t* dst;
	uint64_t dst_size; // logical capacity of dst (the caller's dst_size)
	uint8_t* slots;
	uint32_t* bsize;   // encoded size of every block (tail block last)
	uint32_t* binfo;   // BlockInfo::info of every block
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
WV_FN U32 block_requirement(const U32& a, const U32& info, const Pred& is_tail, uint32_t T)
{
	U32 full = info & 0x7FFFu, need = (info >> 15) & 0x7FFFu;
	Pred lz = ((info >> 31) & 1u) == U32(1u);
	U32 lzreq = a + U32(header_bytes(T)) + full + U32(8u * T + 3u); // C > d + full + 8T + 2
	return sel(!is_tail & lz, lzreq, a + need);
}

WV_FN void plan_superblock(const FrameJob& j, uint64_t s)
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
			U32 incl = wave_incl_scan(sz);
			U32 a = U32(run) + incl - sz;
			gst32((uint8_t*)(j.boff + first), i * 4u, a, p);
			Pred is_tail = pred_all(has_tail) & (i == U32(count - 1));
			uint32_t m = wave_max(sel(p, block_requirement(a, info, is_tail, j.T), U32(0u)));
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
		uint32_t size = gload_uniform(j.bsize + b);
		const bool is_tail = has_tail && i == count - 1;
		if (!is_tail && info_lz_ok(info) && !(C > a + hs + info_full(info) + 8 * T + 2)) {
			// the reference would not have tried the LZ here (block_compress.h:1214): encode the planes instead
			BlockInfo r = encode_block_job(lds, L, T, j.src + b * (uint64_t)bs, j.slots + b * (uint64_t)j.slot_stride, false);
			size = r.size;
			info = r.info;
			gstore_uniform(j.bsize + b, size);
			gstore_uniform(j.binfo + b, info);
		}
		const bool kept_lz = !is_tail && info_lz_ok(info);
		if (!kept_lz && a + info_need(info) > C) // one of the reference's dst_end tests fails
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

// Blocks handed to one wavefront of the pack step: their offsets and sizes are fetched with one vector
// load and the copies of consecutive blocks overlap in the memory system.
constexpr uint32_t PACK_BLOCKS = 32;
WV_HD uint32_t pack_waves_per_superblock(uint32_t bps) { return (bps + 1 + PACK_BLOCKS - 1) / PACK_BLOCKS; }

// Wavefront `w` of superblock `s`: headers + payload of blocks [w*PACK_BLOCKS, ...) into the frame.
WV_FN void pack_blocks(const FrameJob& j, uint64_t s, uint32_t w)
{
	const U32 lane = lane_id();
	if (gload_uniform(j.status))
		return;
	const uint64_t total = gload_uniform64(j.total);
	if (total > j.dst_size) // never write past the caller's buffer; the host reports DST_OVERFLOW
		return;
	const uint32_t count = superblock_blocks(j, s);
	const uint32_t k0 = w * PACK_BLOCKS;
	if (k0 >= count)
		return;
	const uint32_t k1 = k0 + PACK_BLOCKS < count ? k0 + PACK_BLOCKS : count;
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
	if (j.tiny_last && s == j.nsb - 1) { // payload prepared by the host (zstd or raw bytes)
		copy_g2g(base + 4, j.override_payload, csize);
		return;
	}
	const uint32_t bs = 256 * j.T;
	if (code != 1) { // copy superblock: the raw input bytes of these blocks in one piece
		const uint64_t begin = (uint64_t)k0 * bs;
		const uint64_t end = (has_tail && k1 == count) ? (uint64_t)(count - 1) * bs + j.tail_bytes : (uint64_t)k1 * bs;
		copy_g2g(base + 4 + begin, j.src + first * (uint64_t)bs + begin, (uint32_t)(end - begin));
		return;
	}
	Pred mine = (U32(k0) + lane) < U32(k1);
	U32 offs = gld32((const uint8_t*)(j.boff + first + k0), lane * 4u, mine);
	U32 sizes = gld32((const uint8_t*)(j.bsize + first + k0), lane * 4u, mine);
	for (uint32_t k = k0; k < k1; ++k)
		copy_g2g(base + 4 + readlane(offs, k - k0), j.slots + (first + k) * (uint64_t)j.slot_stride, readlane(sizes, k - k0));
}

} // namespace codec
