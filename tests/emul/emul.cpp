// tests/emul/emul.cpp -- TEST INFRASTRUCTURE.  Compiles the wave-level codec sources of the product
// (stenos_amd/csrc/*.h) for the host with WV_HOST_EMULATION: 64 lanes executed in lockstep, LDS as a
// plain buffer.  Lets the CPU test-suite diff the exact kernel logic against the oracle without a GPU.
// The shipped library never contains or calls this.
#define WV_HOST_EMULATION 1
#include "../../stenos_amd/csrc/pipeline.h"
#include "../../stenos_amd/csrc/walk.h"

#include <dlfcn.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>

#include <vector>
typedef size_t (*zc_fn)(void*, size_t, const void*, size_t, int);
typedef unsigned (*ze_fn)(size_t);

using namespace codec;

static uint8_t* alloc_lds(uint32_t bytes)
{
	void* p = nullptr;
	if (posix_memalign(&p, 64, bytes + 256))
		return nullptr;
	memset(p, 0xCD, bytes + 256); // LDS is not zero-initialised on the device either
	return (uint8_t*)p;
}

static int g_dec_regs = 1; // decode_superblock's register path for bytesoftype 2, 4, 8 (what the templated kernels use)

extern "C" {

// payload of one BLOCK superblock with unlimited capacity (block_compress.h:1099-1302)
size_t emul_block_compress(const uint8_t* src, size_t T, size_t bytes, uint8_t* dst, int allow_lz)
{
	Layout L = make_layout((uint32_t)T, true);
	uint8_t* lds = alloc_lds(L.total);
	uint8_t* slot = nullptr;
	if (posix_memalign((void**)&slot, 64, out_capacity((uint32_t)T) + 64))
		return (size_t)-3;
	size_t bs = 256 * T, nb = bytes / bs, off = 0;
	// the device reads whole 16-byte chunks of an aligned source: stage blocks in an aligned copy
	uint8_t* stage = nullptr;
	if (posix_memalign((void**)&stage, 64, bs + 64))
		return (size_t)-3;
	for (size_t b = 0; b < nb; ++b) {
		memcpy(stage, src + b * bs, bs);
		uint32_t n = encode_block_job(lds, L, (uint32_t)T, stage, slot, allow_lz != 0).size;
		memcpy(dst + off, slot, n);
		off += n;
	}
	size_t rem = bytes - nb * bs;
	if (rem) {
		memcpy(stage, src + nb * bs, rem);
		uint32_t n = encode_tail_job(lds, L, (uint32_t)T, stage, (uint32_t)rem, slot).size;
		memcpy(dst + off, slot, n);
		off += n;
	}
	free(stage);
	free(slot);
	free(lds);
	return off;
}

// block_decompress of one superblock payload; misalign shifts the source to exercise the window logic
size_t emul_block_decompress(const uint8_t* src, size_t csize, size_t T, size_t dsize, uint8_t* dst, int misalign)
{
	DecLayout L = make_dec_layout((uint32_t)T);
	uint8_t* lds = alloc_lds(L.total);
	uint8_t* in = nullptr;
	uint8_t* out = nullptr;
	if (posix_memalign((void**)&in, 64, csize + 128) || posix_memalign((void**)&out, 64, dsize + 64))
		return (size_t)-3;
	memset(in, 0xEE, csize + 128);
	memcpy(in + 16 + misalign, src, csize);
	uint32_t r = decode_superblock(lds, L, (uint32_t)T, in + 16 + misalign, (uint32_t)csize, out, (uint32_t)dsize, g_dec_regs && (T == 2 || T == 4 || T == 8));
	if (r != DEC_ERROR)
		memcpy(dst, out, dsize);
	free(in);
	free(out);
	free(lds);
	return r == DEC_ERROR ? (size_t)-4 : r;
}

// The whole encode pipeline as capi.cpp enqueues it (encode_blocks, plan_superblocks, scan_superblocks,
// resolve_frame, host zstd for a tiny last superblock, pack_frame), one "workgroup" after the other.

static int g_fused = 1;
void emul_set_dec_regs(int on) { g_dec_regs = on; }
static int g_slots = 1; // 0: the plane-group loop of encode_run also for bytesoftype 2 and 4 (the fused kernel's GROUPS twin)
void emul_set_slots(int on) { g_slots = on; }
// decoder LDS layout (superblock_codec.h): out = { window offset, image offset, total, window capacity, reach of one block }
void emul_dec_layout(size_t T, uint32_t* out)
{
	const DecLayout L = make_dec_layout((uint32_t)T);
	out[0] = L.win;
	out[1] = L.img;
	out[2] = L.total;
	out[3] = window_bytes((uint32_t)T);
	out[4] = max_block_reach((uint32_t)T);
}
static size_t g_last_fused = 0;
void emul_set_fused(int on) { g_fused = on; }
size_t emul_group_any_count(void) { return (size_t)codec::emul_group_any_count(); } // groups of any shape encoded so far
size_t emul_group4_count(void) { return (size_t)codec::emul_group4_count(); } // groups of four blocks encoded plane by plane so far (superblock_codec.h)
size_t emul_plane_runs_count(void) { return (size_t)codec::emul_plane_runs_count(); } // planes made of run-length coded values decoded by the short form so far (block_codec.h)
size_t emul_lz_serial_count(void) { return (size_t)codec::emul_lz_serial_count(); } // mini-LZ blocks decoded by the serial decoder so far (block_codec.h)
size_t emul_last_fused(void) { return g_last_fused; } // superblocks the last frame sent through the fused path

size_t emul_compress_frame(const uint8_t* src_in, size_t T, size_t bytes, uint8_t* dst, size_t dst_size, int level)
{
	const size_t ERR_DST = (size_t)-6, ERR_PARAM = (size_t)-9;
	if (T < 2 || T >= 65535 || level < 0 || level > 1)
		return ERR_PARAM;
	const size_t bs = 256 * T;
	size_t sb = bs > 131072 ? bs : (131072 / bs) * bs;
	if (dst_size < 8)
		return ERR_DST;
	if (bytes == 0) {
		memset(dst, 0, 8);
		return 8;
	}
	// aligned copy of the source (the device path reads 16-byte chunks of an aligned allocation)
	uint8_t* src = nullptr;
	if (posix_memalign((void**)&src, 64, bytes + 64))
		return (size_t)-3;
	memcpy(src, src_in, bytes);
	FrameJob j;
	memset(&j, 0, sizeof(j));
	j.nsb = bytes / sb + (bytes % sb ? 1 : 0);
	j.nfull = bytes / bs;
	j.tail_bytes = (uint32_t)(bytes % bs);
	j.bps = (uint32_t)(sb / bs);
	j.sb_bytes = (uint32_t)sb;
	j.total_bytes = bytes;
	j.T = (uint32_t)T;
	j.shift_byte = 0;
	j.header_bytes = 8;
	j.force_copy = level == 0;
	const uint64_t nblocks = j.nfull + (j.tail_bytes ? 1 : 0);
	const size_t last_bytes = bytes - (j.nsb - 1) * sb;
	j.tiny_last = (level >= 1 && last_bytes < 128) ? 1 : 0;
	j.slot_stride = out_capacity((uint32_t)T);
	uint8_t* slots = nullptr;
	if (posix_memalign((void**)&slots, 64, (nblocks + 1) * (size_t)j.slot_stride + 64))
		return (size_t)-3;
	std::vector<uint32_t> bsize(nblocks + 1), binfo(nblocks + 1), bneed(nblocks + 1), boff(nblocks + 1), sbcsize(j.nsb + 1), sbneed(j.nsb + 1);
	std::vector<uint8_t> sbcode(j.nsb + 1), frame(dst_size + 64, 0xA5), payload(256);
	std::vector<uint64_t> sboff(j.nsb + 2);
	uint64_t total = 0;
	uint32_t status = 0, first_flagged = 0xFFFFFFFFu;
	j.src = src;
	j.dst = frame.data();
	j.dst_size = dst_size;
	j.slots = slots;
	j.bsize = bsize.data();
	j.binfo = binfo.data();
	j.bneed = bneed.data();
	j.boff = boff.data();
	j.sb_csize = sbcsize.data();
	j.sb_code = sbcode.data();
	j.sb_need = sbneed.data();
	j.sb_off = sboff.data();
	j.total = &total;
	j.status = &status;
	j.first_flagged = &first_flagged;
	j.override_payload = payload.data();
	j.check_total = 1;
	Layout L = make_layout((uint32_t)T, true);
	uint8_t* lds = alloc_lds(L.total);
	// fused zone (capi.cpp enqueue_compress): leading superblocks of full blocks with room for any encoding
	uint64_t s_tight = safe_superblocks(dst_size, j.header_bytes, j.bps, j.T, sb, j.nsb);
	if (j.tiny_last && s_tight > j.nsb - 1)
		s_tight = j.nsb - 1;
	uint64_t s_fused = 0, carry = j.header_bytes;
	if (level >= 1 && g_fused)
		s_fused = j.nfull / j.bps < s_tight ? j.nfull / j.bps : s_tight;
	g_last_fused = s_fused;
	if (s_fused) {
		const uint32_t run_cap = fused_run_capacity(j.bps, j.T);
		uint8_t* stage = nullptr;
		if (posix_memalign((void**)&stage, 64, s_fused * FUSED_WAVES * (size_t)run_cap + 64))
			return (size_t)-3;
		uint8_t* wlds = alloc_lds(FUSED_WAVES * L.total);
		bool guess_copy = false;
		for (uint64_t s = 0; s < s_fused; ++s) { // encode_superblocks
			uint32_t run_size[FUSED_WAVES], code = 0, size = 0;
			for (int attempt = 0;; ++attempt) {
				const bool measure = guess_copy && attempt == 0; // after a superblock that became a copy: sizes first
				for (uint32_t w = 0; w < FUSED_WAVES; ++w) {
					uint32_t b0, b1;
					fused_run_range(j.bps, w, &b0, &b1);
					run_size[w] = encode_run(wlds + w * L.total, L, j.T, src + (s * j.bps + b0) * bs, b1 - b0, measure ? nullptr : stage + (s * FUSED_WAVES + w) * (size_t)run_cap,
								 g_slots != 0);
					if (run_size[w] + 48 > run_cap)
						return (size_t)-1;
				}
				size = fused_superblock_size(j, run_size, &code);
				if (measure && code != 6)
					continue;
				guess_copy = code == 6;
				break;
			}
			sboff[s] = carry; // chain_scanner
			for (uint32_t w = 0; w < FUSED_WAVES; ++w)
				fused_store(j, s, w, carry, run_size, stage + (s * FUSED_WAVES + w) * (size_t)run_cap);
			carry += size;
		}
		sboff[s_fused] = total = carry;
		free(wlds);
		free(stage);
	}
	const uint64_t b_first = s_fused * j.bps < j.nfull ? s_fused * j.bps : (s_fused >= j.nsb ? nblocks : j.nfull);
	if (level >= 1)
		for (uint64_t b = b_first; b < nblocks; ++b) { // encode_blocks
			BlockInfo r = b < j.nfull ? encode_block_job(lds, L, j.T, src + b * bs, slots + b * (size_t)j.slot_stride, true)
						 : encode_tail_job(lds, L, j.T, src + j.nfull * bs, j.tail_bytes, slots + j.nfull * (size_t)j.slot_stride);
			bsize[b] = r.size;
			binfo[b] = r.info;
			bneed[b] = r.need;
		}
	for (uint64_t s = s_fused; s < j.nsb; ++s) // plan_superblocks
		plan_superblock(lds, L, j, s);
	if (s_fused < j.nsb) { // scan_superblocks
		uint64_t off = carry;
		for (uint64_t s = s_fused; s < j.nsb; ++s) {
			sboff[s] = off;
			off += 4 + (uint64_t)sbcsize[s];
		}
		sboff[j.nsb] = total = off;
	}
	resolve_capacity(lds, L, j); // resolve_frame
	size_t result = 0;
	if (j.tiny_last) { // host part of enqueue_compress
		uint64_t off_last = sboff[j.nsb - 1];
		if (status || dst_size < off_last + 4)
			result = ERR_DST;
		else {
			static zc_fn zc = nullptr;
			static ze_fn ze = nullptr;
			if (!zc) {
				void* h = dlopen("/opt/conda/lib/libzstd.so.1", RTLD_NOW | RTLD_LOCAL);
				if (!h)
					h = dlopen("libzstd.so.1", RTLD_NOW | RTLD_LOCAL);
				zc = (zc_fn)dlsym(h, "ZSTD_compress");
				ze = (ze_fn)dlsym(h, "ZSTD_isError");
			}
			uint8_t comp[256];
			size_t room = dst_size - off_last - 4;
			size_t cap = room > sizeof(comp) ? sizeof(comp) : room;
			size_t r = zc(comp, cap, src + (bytes - last_bytes), last_bytes, 1);
			uint32_t code = 2, csize = (uint32_t)r;
			const uint8_t* pl = comp;
			if (ze(r) || r > last_bytes) {
				if (room < last_bytes)
					result = ERR_DST;
				code = 6;
				csize = (uint32_t)last_bytes;
				pl = src + (bytes - last_bytes);
			}
			memcpy(payload.data(), pl, csize);
			sbcode[j.nsb - 1] = (uint8_t)code;
			sbcsize[j.nsb - 1] = csize;
			sboff[j.nsb] = total = off_last + 4 + csize;
			j.override_code = code;
		}
	}
	if (!result) {
		uint8_t* plds = alloc_lds(pack_lds_bytes(j.bps));
		for (uint64_t s = s_fused; s < j.nsb; ++s) // pack_frame
			for (uint32_t w = 0; w < PACK_WAVES; ++w)
				pack_superblock(plds, j, s, w);
		free(plds);
		if (status || total > dst_size)
			result = ERR_DST;
		else {
			result = (size_t)total;
			memcpy(dst, frame.data(), total);
		}
		for (size_t i = (status || total > dst_size) ? 0 : total; i < frame.size(); ++i)
			if (i >= dst_size && frame[i] != 0xA5)
				result = (size_t)-1; // wrote past dst_size
	}
	free(lds);
	free(slots);
	free(src);
	return result;
}

void emul_copy_g2g_wide(uint8_t* dst, const uint8_t* src, size_t n) { copy_g2g_wide(dst, src, (uint32_t)n); }

size_t emul_lds_bytes_encode(size_t T) { return make_layout((uint32_t)T, true).total; }
size_t emul_lds_bytes_decode(size_t T) { return make_dec_layout((uint32_t)T).total; }

// a run of full blocks through encode_run (what a wave of the fused kernel does: the slot encoders for bytesoftype 2, 4, 8)
size_t emul_run_compress(const uint8_t* src, size_t T, size_t nblocks, uint8_t* dst)
{
	Layout L = make_layout((uint32_t)T, true);
	uint8_t* lds = alloc_lds(L.total);
	uint8_t* stage = nullptr;
	const size_t cap = nblocks * max_block_bytes((uint32_t)T) + 256;
	if (posix_memalign((void**)&stage, 64, cap))
		return (size_t)-3;
	memset(stage, 0xCD, cap);
	const uint32_t n = encode_run(lds, L, (uint32_t)T, src, (uint32_t)nblocks, stage, g_slots != 0);
	memcpy(dst, stage, n);
	free(stage);
	free(lds);
	return n;
}

// a run that is only measured while its raw bytes are put where a copy would stand (kernels.hip, speculative copy)
size_t emul_measure_run(const uint8_t* src, size_t T, size_t nblocks, uint8_t* raw_out)
{
	Layout L = make_layout((uint32_t)T, true);
	uint8_t* lds = alloc_lds(L.total);
	const uint32_t n = encode_run(lds, L, (uint32_t)T, src, (uint32_t)nblocks, nullptr, true, NoPassHook(), raw_out);
	free(lds);
	return n;
}

// ---- walk.h: the parallel header walk, its kernels replayed lane by lane (walk_kernels.hip) --------------------------
// off[0 .. nsb], *status |= 1 for a truncated chain as the serial walk says.  Returns the number of segments (0: the plan
// chose the serial walk) or -1 when the speculation failed and the serial walk has to run.
int emul_walk_parallel(const uint8_t* frame, uint64_t size, uint64_t first, uint64_t nsb, uint32_t sb_bytes, uint64_t seg_len, uint64_t* off, uint32_t* status)
{
	using namespace walk;
	const Plan P = make_plan(first, size, nsb, sb_bytes, seg_len);
	if (P.nseg == 0)
		return 0;
	std::vector<Segment> seg(P.nseg);
	for (Segment& s : seg) {
		memset(&s, 0xDB, sizeof s); // (stale contents of the device buffer)
	}
	struct Add {
		uint32_t operator()(uint32_t* c) const { return (*c)++; }
	};
	for (uint32_t k = 0; k < P.nseg; ++k) { // walk_speculate
		if (k == 0) {
			follow_first(P, frame, &seg[0]);
			continue;
		}
		if (k + 1 >= P.nseg)
			continue;
		uint64_t roots[LANES];
		uint32_t nroots = 0;
		const uint64_t begin = seg_begin(P, k), wend = begin + P.window;
		for (uint32_t lane = 0; lane < LANES; ++lane)
			for (uint64_t base = begin + lane * 16u; base < wend; base += LANES * 16u)
				scan_window16(P, frame, k, base, roots, &nroots, Add());
		uint32_t live = 0, hops[LANES];
		uint64_t exit[LANES];
		bool alive[LANES] = { false };
		if (nroots <= LANES)
			for (uint32_t lane = 0; lane < nroots; ++lane)
				if ((alive[lane] = follow_root(P, frame, k, roots[lane], &exit[lane], &hops[lane])))
					++live;
		bool ok = live >= 1 && live <= MAX_ROOTS;
		uint32_t lead = 0;
		while (ok && !alive[lead])
			++lead;
		for (uint32_t lane = 0; ok && lane < LANES; ++lane)
			if (alive[lane] && exit[lane] != exit[lead])
				ok = false;
		if (ok) {
			uint32_t i = 0;
			for (uint32_t lane = 0; lane < LANES; ++lane)
				if (alive[lane]) {
					seg[k].root[i] = roots[lane];
					seg[k].hops[i] = hops[lane];
					++i;
				}
			seg[k].exit = exit[lead];
			seg[k].count = 0;
			seg[k].nroots = live;
			seg[k].state = SEG_OK;
		}
		else {
			seg[k].state = SEG_UNRESOLVED;
			if (getenv("WALK_DEBUG")) fprintf(stderr, "seg %u: %u roots, %u live\n", k, nroots, live);
		}
	}
	bool failed = false;
	for (uint32_t k = 0; k < P.nseg; ++k) // walk_verify (its lanes only read what walk_speculate wrote)
		if (!verify_segment(P, frame, k, seg.data())) {
			if (getenv("WALK_DEBUG")) fprintf(stderr, "verify %u failed: prev exit %llu root %llu begin %llu\n", k, (unsigned long long)seg[k-1].exit, (unsigned long long)seg[k].root[0], (unsigned long long)seg_begin(P,k));
			failed = true;
		}
	if (failed)
		return -1;
	for (uint32_t k = 0; k < P.nseg; ++k) { // walk_write
		uint64_t sum = 0;
		for (uint32_t j = 0; j < k; ++j)
			sum += seg[j].count;
		*status |= write_segment(P, frame, k, seg.data(), sum, off, 1u);
	}
	return (int)P.nseg;
}
}
