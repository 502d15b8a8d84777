// capi.cpp -- host side of libstenos.so: the frozen Stenos C ABI (include/stenos.h) and the
// device-pointer entry points (include/stenos_hip.h) on top of the gfx950 kernels of kernels.hip.
//
// What this file restates from the reference (stenos/internal/stenos.cpp): context and setters
// (:81-286), frame header and superblock sizing (:115-185, 862-874), the superblock strategy slice for
// levels 0/1 (:403-450, 606-615, 658-678), decode framing (:1052-1208), stenos_get_info (:1019-1050),
// the private single-superblock API (:768-842) and the timer wrappers (:1232-1257).  The per-chunk
// thread dispatcher (:909-1010, tiny_pool.h) is replaced by the GPU grid: `threads` is accepted and ignored.
//
// There is no CPU codec here.  If no HIP device is usable every codec call returns
// STENOS_ERROR_INVALID_INSTRUCTION_SET (the reference's code for "required instruction set missing").
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <functional>
#include <memory>
#include <new>
#include <thread>
#include <vector>

#include "../../include/stenos_hip.h"
#include "kernels.h"
#include "strategy.h"

namespace {

constexpr size_t kMaxT = STENOS_MAX_BYTESOFTYPE - 1; // stenos.h:65; above STENOS_K_LDS_MAX_T the kernels of kernels_wide.hip take over

inline bool is_err(size_t r) { return r >= STENOS_LAST_ERROR_CODE; }

// ---- zstd through dlopen: only for superblocks < 128 bytes (stenos.cpp:435-437) and for decoding code 2 ----
typedef size_t (*zstd_compress_fn)(void*, size_t, const void*, size_t, int);
typedef size_t (*zstd_decompress_fn)(void*, size_t, const void*, size_t);
typedef unsigned (*zstd_iserror_fn)(size_t);
typedef int (*zstd_maxclevel_fn)(void);
typedef void* (*zstd_createcctx_fn)(void);
typedef size_t (*zstd_freecctx_fn)(void*);
typedef size_t (*zstd_compresscctx_fn)(void*, void*, size_t, const void*, size_t, int);
struct Zstd {
	zstd_maxclevel_fn max_level = nullptr;
	zstd_compress_fn compress_once = nullptr;
	zstd_createcctx_fn create_cctx = nullptr;
	zstd_freecctx_fn free_cctx = nullptr;
	zstd_compresscctx_fn compress_cctx = nullptr;
	// ZSTD_compress allocates and frees a context of several hundred KB per call, which serialises dozens of worker
	// threads in the allocator; each thread keeps one context instead (ZSTD_compressCCtx, what the reference calls,
	// zstd_wrapper.h:81-83: same bytes)
	size_t compress(void* dst, size_t cap, const void* src, size_t n, int level) const
	{
		struct Holder {
			void* c = nullptr;
			zstd_freecctx_fn fr = nullptr;
			~Holder()
			{
				if (c && fr)
					fr(c);
			}
		};
		static thread_local Holder h;
		if (!h.c && create_cctx) {
			h.c = create_cctx();
			h.fr = free_cctx;
		}
		return h.c ? compress_cctx(h.c, dst, cap, src, n, level) : compress_once(dst, cap, src, n, level);
	}
	zstd_decompress_fn decompress = nullptr;
	zstd_iserror_fn is_error = nullptr;
	bool ok = false;
	Zstd()
	{
		const char* names[] = { "/opt/conda/lib/libzstd.so.1", "libzstd.so.1", "libzstd.so", nullptr };
		void* h = nullptr;
		for (int i = 0; names[i] && !h; ++i)
			h = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
		if (!h)
			return;
		compress_once = (zstd_compress_fn)dlsym(h, "ZSTD_compress");
		create_cctx = (zstd_createcctx_fn)dlsym(h, "ZSTD_createCCtx");
		free_cctx = (zstd_freecctx_fn)dlsym(h, "ZSTD_freeCCtx");
		compress_cctx = (zstd_compresscctx_fn)dlsym(h, "ZSTD_compressCCtx");
		if (!create_cctx || !free_cctx || !compress_cctx)
			create_cctx = nullptr;
		decompress = (zstd_decompress_fn)dlsym(h, "ZSTD_decompress");
		is_error = (zstd_iserror_fn)dlsym(h, "ZSTD_isError");
		max_level = (zstd_maxclevel_fn)dlsym(h, "ZSTD_maxCLevel");
		ok = compress_once && decompress && is_error && max_level;
	}
};
Zstd& zstd()
{
	static Zstd z;
	return z;
}

struct DevBuf {
	void* p = nullptr;
	size_t cap = 0;
	bool ensure(size_t n)
	{
		if (n <= cap)
			return true;
		if (p)
			(void)hipFree(p);
		p = nullptr;
		cap = 0;
		size_t want = (n + 4095) & ~(size_t)4095;
		if (hipMalloc(&p, want) != hipSuccess) {
			p = nullptr;
			return false;
		}
		cap = want;
		return true;
	}
	void release()
	{
		if (p)
			(void)hipFree(p);
		p = nullptr;
		cap = 0;
	}
	template <class T>
	T* as() const
	{
		return (T*)p;
	}
};

// Host staging of the strategy layer (levels >= 2): page-locked so the transfers run at link speed; plain malloc
// when the pinned allocation fails.  Kept by the context between calls.
struct HostBuf {
	uint8_t* p = nullptr;
	size_t cap = 0;
	bool pinned = false;
	bool ensure(size_t n)
	{
		if (n <= cap)
			return true;
		release();
		const size_t want = (n + (n >> 3) + 4095) & ~(size_t)4095;
		void* q = nullptr;
		if (hipHostMalloc(&q, want, hipHostMallocDefault) == hipSuccess)
			pinned = true;
		else {
			(void)hipGetLastError();
			q = malloc(want);
			pinned = false;
		}
		if (!q)
			return false;
		p = (uint8_t*)q;
		cap = want;
		return true;
	}
	void release()
	{
		if (p) {
			if (pinned)
				(void)hipHostFree(p);
			else
				free(p);
		}
		p = nullptr;
		cap = 0;
	}
	uint8_t* data() const { return p; }
};

// builds with -DSTENOS_HOST_TRACE: wall-clock of the host phases of the strategy layer on stderr (diagnostics)
// Wall time of the stages of a levels >= 2 call (the host's strategy layer around the GPU passes), summed per stage name
// into the context: stenos_hip_stage_ms() reads them (bench.py reports them); -DSTENOS_HOST_TRACE also prints every mark.
enum StageId { STAGE_GPU_PASS = 0, STAGE_ESTIMATES, STAGE_BLOCKS_TO_HOST, STAGE_ZSTD, STAGE_LAYOUT, STAGE_UPLOAD, STAGE_INFLATE, STAGE_DEVICE_FINISH, STAGE_COUNT };
struct PhaseTrace {
	double* acc; // STAGE_COUNT sums in milliseconds (nullptr: none)
	std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
	explicit PhaseTrace(double* sums = nullptr) : acc(sums) {}
	void mark(const char* what, int stage)
	{
		const auto n = std::chrono::steady_clock::now();
		const double ms = std::chrono::duration<double, std::milli>(n - t).count();
		if (acc)
			acc[stage] += ms;
#ifdef STENOS_HOST_TRACE
		fprintf(stderr, "[stenos] %-28s %8.2f ms\n", what, ms);
#else
		(void)what;
#endif
		t = n;
	}
};

inline void put_le(uint8_t* p, uint64_t v, int n)
{
	for (int i = 0; i < n; ++i)
		p[i] = (uint8_t)(v >> (8 * i));
}
inline uint64_t get_le(const uint8_t* p, int n)
{
	uint64_t v = 0;
	for (int i = 0; i < n; ++i)
		v |= (uint64_t)p[i] << (8 * i);
	return v;
}

// stenos.cpp:71-76
inline size_t base_superblock(size_t block_size)
{
	if (block_size > STENOS_BLOCK_SIZE)
		return block_size;
	return (STENOS_BLOCK_SIZE / block_size) * block_size;
}

} // namespace

struct stenos_context_s {
	// parameters (stenos.cpp:94-106)
	int level = 1;
	int threads = 1;
	uint64_t max_nanoseconds = 0;
	size_t custom_shift = STENOS_NO_BLOCK_SHIFT;

	// device state
	bool probed = false, usable = false;
	DevBuf in, out;                                  // staging for the host-pointer ABI
	DevBuf slots, bsize, binfo, bneed, boff, sbcsize, sbneed, sbcode, sboff; // workspace of the encode pipeline / decode index
	DevBuf chain;                                    // fused path: ticket counter + one chained-scan word per superblock
	DevBuf tmp1, tmp2;                               // device scratch for superblocks that pass through zstd on the host (codes 3-5)
	DevBuf qprod, shuf, mid0, mid1;                  // levels >= 2: ratio checkpoints, shuffled input, plane middles (raw / delta'd)
	DevBuf walk;                                     // segments of the parallel header walk (walk.h)
	DevBuf dslots, dtab;                             // levels >= 2, device destinations: zstd output slots of two batches and their offset / size tables
	bool test_serial_walk = false;                   // (only the test build can set it) stenos_hip_test_walk: frames without an index are walked by one lane
	DevBuf wide;                                     // bytesoftype above 64: scratch of the HBM-resident kernels (kernels_wide.hip)
	DevBuf misc;                                     // [0,8) total, [8,12) decode status, [12,16) encode status, [16,20) first flagged, [20,24) unused, [24,32) scan carry, [64,320) override payload
	HostBuf h_in, h_out, h_blocks, h_shuf, h_mid0, h_mid1, h_stage, h_tab; // host staging of the strategy layer
	uint64_t* h_total = nullptr;                     // pinned copy of misc[0,16) for compress; decode status at +32
	// last asynchronous job
	hipStream_t job_stream = nullptr;
	int job_kind = 0; // 0 none, 1 compress, 2 decompress
	size_t job_dst_size = 0, job_expected = 0;
	bool job_host_codes = false; // the last decode met zstd-based superblocks (finished on the host)
	size_t last_nsb = 0;
	// optional kernel timing (stenos_hip_set_profiling)
	bool profiling = false;
	hipEvent_t ev[4] = { nullptr, nullptr, nullptr, nullptr }; // encode start/stop, decode start/stop
	bool ev_valid[2] = { false, false };
	hipStream_t up_stream = nullptr, main_stream = nullptr; // chunked host-pointer calls: uploads / coding + downloads
	hipStream_t copy_stream = nullptr, upload_stream = nullptr; // levels >= 2: block streams to the host / the frame to the device, beside the host's zstd
	std::vector<hipEvent_t> batch_ev;   // ... one event per batch of superblocks
	std::vector<hipEvent_t> set_ev;     // ... and one per set of zstd output slots on their way to the device
	std::vector<hipStream_t> set_streams; // decode of zstd-based superblocks: one stream per set of inflated batches
	double stage_ms[16] = { 0 }; // levels >= 2: wall time per stage of the strategy layer, summed over the calls (stenos_hip_stage_ms)
	bool warm = false;    // a device call has gone through on this context (buffers, code objects and streams are up)
	int last_devices = 1; // devices the last host-pointer call used
	int hip_devices = 0;  // stenos_hip_set_devices: devices a host-pointer call may spread over (0: STENOS_HIP_DEVICES, else one)
	bool test_lanes_share_device = false; // stenos_hip_test_lanes: the lanes all use the current device (one-GPU test boxes)
	int test_fail_lane = -1;              // stenos_hip_test_lanes: this lane never runs (error-path test)
	// what the last compression was asked to do: a fused launch that gave up waiting is done again without the fused kernel
	const void* job_src = nullptr;
	void* job_dst = nullptr;
	size_t job_T = 0, job_bytes = 0;
	bool no_fused = false;
	int fused_fallbacks = 0;      // times that happened (stenos_hip_fused_fallbacks)
	int inject_chain_timeout = 0; // (only the test build can set it, stenos_hip_test_fused_timeouts) the next n fused launches are treated as if they had given up
	int device = -1; // the device the buffers above live on (the one that was current when they were first needed)
	// host-pointer calls with stenos_set_threads(ctx, n > 1): one child context per further device (or per stand-in lane),
	// used from a host thread of its own (multi_device below)
	std::vector<stenos_context_s*> lanes;

	void release_device_state()
	{
		DevBuf* all[] = { &in, &out, &slots, &bsize, &binfo, &bneed, &boff, &sbcsize, &sbneed, &sbcode, &sboff, &misc, &tmp1, &tmp2, &qprod, &shuf, &mid0, &mid1, &chain, &wide, &walk, &dslots, &dtab };
		for (DevBuf* b : all)
			b->release();
		if (h_total)
			(void)hipHostFree(h_total);
		h_total = nullptr;
		for (hipEvent_t& e : ev)
			if (e) {
				(void)hipEventDestroy(e);
				e = nullptr;
			}
		ev_valid[0] = ev_valid[1] = false;
		for (hipStream_t* s : { &up_stream, &main_stream, &copy_stream, &upload_stream })
			if (*s) {
				(void)hipStreamDestroy(*s);
				*s = nullptr;
			}
		for (hipEvent_t e : batch_ev)
			(void)hipEventDestroy(e);
		batch_ev.clear();
		for (hipEvent_t e : set_ev)
			(void)hipEventDestroy(e);
		set_ev.clear();
		for (hipStream_t st : set_streams)
			(void)hipStreamDestroy(st);
		set_streams.clear();
		last_nsb = 0;
		job_kind = 0;
	}
	bool device_ready()
	{
		int cur = -1;
		if (probed && usable && hipGetDevice(&cur) == hipSuccess && cur != device) {
			// the caller switched devices between calls: buffers of the old device are of no use on this one.  An asynchronous
			// job that is still pending there is waited for first (its stream outlives the switch); its result is lost to
			// stenos_hip_finish, which then reports that there is no job -- not a silent success.
			if (job_kind && job_stream)
				(void)hipStreamSynchronize(job_stream);
			release_device_state();
			probed = false;
			warm = false;
		}
		if (!probed) {
			probed = true;
			int n = 0;
			usable = hipGetDeviceCount(&n) == hipSuccess && n > 0 && hipGetDevice(&device) == hipSuccess;
			if (usable && hipHostMalloc((void**)&h_total, 64, hipHostMallocDefault) != hipSuccess)
				usable = false;
		}
		return usable;
	}
	~stenos_context_s()
	{
		for (stenos_context_s* l : lanes)
			if (l) {
				l->~stenos_context_s();
				free(l);
			}
		DevBuf* all[] = { &in, &out, &slots, &bsize, &binfo, &bneed, &boff, &sbcsize, &sbneed, &sbcode, &sboff, &misc, &tmp1, &tmp2, &qprod, &shuf, &mid0, &mid1, &chain, &wide, &walk, &dslots, &dtab };
		for (DevBuf* b : all)
			b->release();
		HostBuf* host[] = { &h_in, &h_out, &h_blocks, &h_shuf, &h_mid0, &h_mid1, &h_stage, &h_tab };
		for (HostBuf* b : host)
			b->release();
		if (h_total)
			(void)hipHostFree(h_total);
		for (hipEvent_t e : ev)
			if (e)
				(void)hipEventDestroy(e);
		for (hipStream_t s : { up_stream, main_stream, copy_stream, upload_stream })
			if (s)
				(void)hipStreamDestroy(s);
		for (hipEvent_t e : batch_ev)
			(void)hipEventDestroy(e);
		for (hipEvent_t e : set_ev)
			(void)hipEventDestroy(e);
		for (hipStream_t st : set_streams)
			(void)hipStreamDestroy(st);
	}
	void mark(int idx, hipStream_t stream)
	{
		if (!profiling)
			return;
		if (!ev[idx] && hipEventCreate(&ev[idx]) != hipSuccess)
			return;
		if (hipEventRecord(ev[idx], stream) == hipSuccess && (idx & 1))
			ev_valid[idx >> 1] = true;
	}
};

namespace {

size_t finish_job(stenos_context_s* ctx);

struct FramePlan {
	size_t sb = 0;          // superblock bytes
	uint32_t shift = 0;     // frame byte 0 (255 = custom size follows)
	size_t header = 8;      // frame header bytes
	uint64_t nsb = 0, nfull = 0;
	uint32_t tail = 0, bps = 0;
};

// ctx->prepare + frame geometry (stenos.cpp:115-185, 853-874).  Returns 0 or an error code.
size_t plan_frame(const stenos_context_s* ctx, size_t T, size_t bytes, int level, FramePlan& f)
{
	if (T == 0 || T >= STENOS_MAX_BYTESOFTYPE)
		return STENOS_ERROR_INVALID_BYTESOFTYPE;
	const size_t bs = T * 256;
	if (ctx->custom_shift != STENOS_NO_BLOCK_SHIFT) {
		f.sb = bs << ctx->custom_shift;
		f.shift = 255;
		f.header = 12;
	}
	else {
		f.sb = base_superblock(bs);
		f.shift = 0;
		if (bytes > f.sb) {
			f.shift = level ? (uint32_t)(level - 1) / 2 : 0;
			f.sb <<= f.shift;
		}
		f.header = 8;
	}
	if (f.sb < bs || f.sb >= STENOS_MAX_BLOCK_BYTES)
		return STENOS_ERROR_INVALID_PARAMETER;
	f.nsb = bytes / f.sb + (bytes % f.sb ? 1 : 0);
	f.nfull = bytes / bs;
	f.tail = (uint32_t)(bytes % bs);
	f.bps = (uint32_t)(f.sb / bs);
	return 0;
}

// What this build cannot do is refused loudly instead of being routed to a CPU path (nothing at present: every level and
// every bytesoftype the reference accepts has a device path).
size_t check_supported(const stenos_context_s* ctx, size_t T, int level)
{
	(void)ctx;
	(void)level;
	return T > kMaxT ? STENOS_ERROR_INVALID_BYTESOFTYPE : 0;
}
// bytesoftype above 64: scratch for the workgroups of kernels_wide.hip, at most 1 GiB (at least one workgroup's worth)
bool wide_scratch(stenos_context_s* ctx, size_t T, uint64_t units, uint8_t** p, uint64_t* bytes);
bool wide_scratch(stenos_context_s* ctx, size_t T, uint64_t units, uint8_t** p, uint64_t* bytes)
{
	*p = nullptr;
	*bytes = 0;
	if (T <= STENOS_K_LDS_MAX_T)
		return true;
	const uint64_t stride = stenos_kw_scratch_stride((uint32_t)T);
	uint64_t groups = ((uint64_t)1 << 30) / stride;
	groups = groups > units ? units : groups;
	groups = groups > 2048 ? 2048 : (groups < 1 ? 1 : groups);
	if (!ctx->wide.ensure((size_t)(groups * stride)))
		return false;
	*p = ctx->wide.as<uint8_t>();
	*bytes = groups * stride;
	return true;
}
// levels >= 2 and bytesoftype 1 go through the strategy layer (block codec on the GPU + zstd on the host)
inline bool needs_strategy(size_t T, int level) { return level >= 2 || (level == 1 && T == 1); }

// Enqueue the compression of `bytes` device bytes into a frame (or, with frame_header == false, into
// the bare superblock stream used by the private API).  Nothing is waited for except, for a final
// superblock shorter than 128 bytes, the copy of those bytes to the host.
size_t enqueue_compress(stenos_context_s* ctx, const uint8_t* d_src, size_t T, size_t bytes, uint8_t* d_dst, size_t dst_size, int level,
			const FramePlan& f, bool frame_header, hipStream_t stream)
{
	const uint64_t nblocks = f.nfull + (f.tail ? 1 : 0);
	const uint32_t stride = stenos_k_slot_stride((uint32_t)T);
	if (!ctx->bsize.ensure((nblocks + 1) * 4) || !ctx->binfo.ensure((nblocks + 1) * 4) || !ctx->bneed.ensure((nblocks + 1) * 4) || !ctx->boff.ensure((nblocks + 1) * 4) ||
	    !ctx->sbcsize.ensure((f.nsb + 1) * 4) || !ctx->sbneed.ensure((f.nsb + 1) * 4) || !ctx->sbcode.ensure(f.nsb + 1) ||
	    !ctx->sboff.ensure((f.nsb + 8) * 8) || !ctx->misc.ensure(4096))
		return STENOS_ERROR_ALLOC;

	const size_t last_bytes = bytes - (f.nsb - 1) * f.sb;
	const bool tiny_last = level >= 1 && last_bytes < 128; // small input: direct zstd (stenos.cpp:435-437)
	if (tiny_last && !zstd().ok)
		return STENOS_ERROR_ZSTD_INTERNAL;

	uint8_t* misc = ctx->misc.as<uint8_t>();
	codec::FrameJob j;
	j.src = d_src;
	j.dst = d_dst;
	j.dst_size = dst_size;
	j.slots = ctx->slots.as<uint8_t>();
	j.bsize = ctx->bsize.as<uint32_t>();
	j.binfo = ctx->binfo.as<uint32_t>();
	j.bneed = ctx->bneed.as<uint32_t>();
	if (!wide_scratch(ctx, T, nblocks, &j.wide_scratch, &j.wide_scratch_bytes))
		return STENOS_ERROR_ALLOC;
	j.boff = ctx->boff.as<uint32_t>();
	j.sb_csize = ctx->sbcsize.as<uint32_t>();
	j.sb_code = ctx->sbcode.as<uint8_t>();
	j.sb_need = ctx->sbneed.as<uint32_t>();
	j.sb_off = ctx->sboff.as<uint64_t>();
	j.total = (uint64_t*)misc;
	j.status = (uint32_t*)(misc + 12);
	j.first_flagged = (uint32_t*)(misc + 16);
	j.override_payload = misc + 64;
	j.nfull = f.nfull;
	j.nsb = f.nsb;
	j.total_bytes = bytes;
	j.tail_bytes = f.tail;
	j.bps = f.bps;
	j.sb_bytes = (uint32_t)f.sb;
	j.slot_stride = stride;
	j.T = (uint32_t)T;
	j.shift_byte = frame_header ? f.shift : 0xFFFFFFFFu;
	j.header_bytes = frame_header ? (uint32_t)f.header : 0u;
	j.force_copy = level == 0 ? 1u : 0u;
	j.tiny_last = tiny_last ? 1u : 0u;
	j.override_code = 0;

	j.check_total = 0;
	j.fixed_capacity = 0;
	j.qprod = nullptr;
	const uint64_t header = j.header_bytes;

	// Superblocks whose capacity is certainly large enough for any encoding ("safe zone", normally all but the
	// last one or two) need no capacity replay and no overflow check; they are processed in chunks, the
	// pack of a chunk overlapping the encoding of the next one on a second stream.  The remaining tail zone
	// goes through plan / scan / resolve / pack in order.
	uint64_t s_tight = codec::safe_superblocks(dst_size, header, f.bps, (uint32_t)T, f.sb, f.nsb);
	if (tiny_last && s_tight > f.nsb - 1)
		s_tight = f.nsb - 1;
	const uint64_t nblocks_all = f.nfull + (f.tail ? 1 : 0);
	auto first_block = [&](uint64_t sb_index) { // first block of a superblock (nblocks_all for sb_index == nsb)
		const uint64_t b = sb_index * f.bps;
		return sb_index >= f.nsb ? nblocks_all : (b < f.nfull ? b : f.nfull);
	};

	uint64_t* d_carry = (uint64_t*)(misc + 24);
	// Safe superblocks that consist of full blocks go through the fused kernel (encode + offsets + store in one launch).
	// (offset 0 means "not published yet" to the fused kernel, so frames without a header stay on the other path)
	uint64_t s_fused = 0;
	if (level >= 1 && header > 0 && stenos_k_fused_supported((uint32_t)T) && !ctx->no_fused)
		s_fused = f.nfull / f.bps < s_tight ? f.nfull / f.bps : s_tight;
	// One arena serves both: the staging streams of the fused superblocks, then (the fused kernel is done by
	// then) the 16-byte aligned slots of the remaining blocks, addressed by their absolute block number.
	const uint64_t b_unfused = first_block(s_fused);
	if (level >= 1) {
		const size_t stage_bytes = s_fused ? stenos_k_fused_stage_bytes((uint32_t)T, f.bps, s_fused) : 0;
		const size_t slot_bytes = (size_t)(nblocks_all - b_unfused + 1) * stride;
		if (!ctx->slots.ensure(stage_bytes > slot_bytes ? stage_bytes : slot_bytes))
			return STENOS_ERROR_ALLOC;
		j.slots = ctx->slots.as<uint8_t>() - b_unfused * (uint64_t)stride;
	}
	if (s_fused) {
		if (!ctx->chain.ensure((s_fused + 2) * 8))
			return STENOS_ERROR_ALLOC;
		uint64_t* desc = ctx->chain.as<uint64_t>() + 1; // word 0: ticket counter
		if (stenos_k_launch_init(misc, header, ctx->chain.as<uint64_t>(), s_fused + 2, j.sb_off, s_fused + 8, stream) != hipSuccess)
			return STENOS_ERROR_UNDEFINED;
		ctx->mark(0, stream);
		if (stenos_k_launch_encode_fused(j, s_fused, ctx->slots.as<uint8_t>(), desc, ctx->chain.as<uint32_t>(), d_carry, stream) != hipSuccess)
			return STENOS_ERROR_UNDEFINED;
		ctx->mark(1, stream);
	}
	else if (stenos_k_launch_init(misc, header, nullptr, 0, nullptr, 0, stream) != hipSuccess)
		return STENOS_ERROR_UNDEFINED;
	if (s_tight > s_fused) {
		// safe superblocks the fused kernel does not take (bytesoftype too large for its LDS budget, frames without a header):
		// one encode / plan / scan / pack sequence.  (Overlapping the pack of one chunk with the encoding of the next on a
		// second stream was measured on MI355X and gains nothing.)
		if (s_fused == 0)
			ctx->mark(0, stream); // kernel timing: the encode_blocks launch of the safe zone
		if (level >= 1 && stenos_k_launch_encode(j, first_block(s_fused), first_block(s_tight), stream) != hipSuccess)
			return STENOS_ERROR_UNDEFINED;
		if (s_fused == 0)
			ctx->mark(1, stream);
		if (stenos_k_launch_plan(j, s_fused, s_tight, stream) != hipSuccess || stenos_k_launch_scan(j, s_fused, s_tight, d_carry, stream) != hipSuccess ||
		    stenos_k_launch_pack(j, s_fused, s_tight, stream) != hipSuccess)
			return STENOS_ERROR_UNDEFINED;
	}
	// tail zone
	j.check_total = 1;
	if (s_tight < f.nsb) {
		if (s_tight == 0)
			ctx->mark(0, stream);
		if (level >= 1 && stenos_k_launch_encode(j, first_block(s_tight), nblocks_all, stream) != hipSuccess)
			return STENOS_ERROR_UNDEFINED;
		if (s_tight == 0)
			ctx->mark(1, stream);
		if (stenos_k_launch_plan(j, s_tight, f.nsb, stream) != hipSuccess || stenos_k_launch_scan(j, s_tight, f.nsb, d_carry, stream) != hipSuccess ||
		    stenos_k_launch_resolve(j, stream) != hipSuccess)
			return STENOS_ERROR_UNDEFINED;
	}

	if (tiny_last) {
		// The reference hands zstd the rest of the caller's buffer as capacity (stenos.cpp:666, 895), and
		// zstd's result depends on it, so the final offset of this last superblock must be known first.
		uint64_t off_last = 0;
		uint32_t status = 0;
		uint8_t raw[128], comp[256];
		if (hipMemcpyAsync(&off_last, j.sb_off + (f.nsb - 1), 8, hipMemcpyDeviceToHost, stream) != hipSuccess ||
		    hipMemcpyAsync(&status, j.status, 4, hipMemcpyDeviceToHost, stream) != hipSuccess ||
		    hipMemcpyAsync(raw, d_src + (bytes - last_bytes), last_bytes, hipMemcpyDeviceToHost, stream) != hipSuccess ||
		    hipStreamSynchronize(stream) != hipSuccess)
			return STENOS_ERROR_UNDEFINED;
		if (status || dst_size < off_last + 4) // an earlier superblock did not fit / no room for this header (stenos.cpp:427-429)
			return STENOS_ERROR_DST_OVERFLOW;
		const size_t room = dst_size - (size_t)off_last - 4;
		size_t cap = room > sizeof(comp) ? sizeof(comp) : room; // above ZSTD_compressBound(127) the capacity no longer matters
		size_t r = zstd().compress(comp, cap, raw, last_bytes, 1); // zstd level 1 (zstd_wrapper.h:49-56)
		const uint8_t* payload = comp;
		uint32_t code = 2, csize = (uint32_t)r;
		if (zstd().is_error(r) || r > last_bytes) { // -> MEMCPY (stenos.cpp:668-669, 366-367)
			if (room < last_bytes)
				return STENOS_ERROR_DST_OVERFLOW;
			code = 6;
			csize = (uint32_t)last_bytes;
			payload = raw;
		}
		const uint64_t end = off_last + 4 + csize;
		const uint8_t code8 = (uint8_t)code;
		if (hipMemcpyAsync(misc + 64, payload, csize, hipMemcpyHostToDevice, stream) != hipSuccess ||
		    hipMemcpyAsync(j.sb_code + (f.nsb - 1), &code8, 1, hipMemcpyHostToDevice, stream) != hipSuccess ||
		    hipMemcpyAsync(j.sb_csize + (f.nsb - 1), &csize, 4, hipMemcpyHostToDevice, stream) != hipSuccess ||
		    hipMemcpyAsync(j.sb_off + f.nsb, &end, 8, hipMemcpyHostToDevice, stream) != hipSuccess ||
		    hipMemcpyAsync(j.total, &end, 8, hipMemcpyHostToDevice, stream) != hipSuccess ||
		    hipStreamSynchronize(stream) != hipSuccess) // the sources live on this stack frame
			return STENOS_ERROR_UNDEFINED;
		j.override_code = code;
	}
	if (stenos_k_launch_pack(j, s_tight, f.nsb, stream) != hipSuccess)
		return STENOS_ERROR_UNDEFINED;
	// total (8 bytes) and the encode status (4 bytes at +12) travel together
	if (hipMemcpyAsync(ctx->h_total, misc, 16, hipMemcpyDeviceToHost, stream) != hipSuccess)
		return STENOS_ERROR_UNDEFINED;
	ctx->last_nsb = f.nsb;
	return 0;
}

// zstd_from_reduced_level (zstd_wrapper.h:49-56)
int zstd_level_of(int clevel)
{
	if (clevel < 1)
		return 1;
	if (clevel < 9)
		return clevel * 2 - 1;
	return zstd().max_level();
}

// compress_memcpy (stenos.cpp:363-374) into host memory
size_t host_copy_superblock(const uint8_t* src, size_t bytes, uint8_t* dst, size_t room)
{
	if (room < bytes + 4)
		return STENOS_ERROR_DST_OVERFLOW;
	dst[0] = 6;
	put_le(dst + 1, bytes, 3);
	memcpy(dst + 4, src, bytes);
	return bytes + 4;
}

// Host side worker threads for the zstd stages of levels >= 2 (one superblock per task).
#ifndef STENOS_HOST_THREADS_CAP
#define STENOS_HOST_THREADS_CAP 64
#endif
constexpr unsigned HOST_THREADS_DEFAULT_CAP = STENOS_HOST_THREADS_CAP; // workers of the strategy layer unless STENOS_HOST_THREADS says otherwise (at most 256)
unsigned host_threads()
{
	static const unsigned threads = [] { // (read once: no environment look-ups on the call path)
		unsigned n = std::thread::hardware_concurrency();
		// a container's CPU quota (cgroup v2 cpu.max / v1 cfs quota) is what the workers really get: beyond about 1.5 x
		// of it more threads only take time slices from each other (measured on a 16-CPU share of a 256-thread host:
		// 24 workers 13.8 GB/s, 64: 11.5, 256: 4.7 for doubles at level 2)
		{
			double quota = 0, period = 0;
			if (FILE* fp = fopen("/sys/fs/cgroup/cpu.max", "r")) {
				char q[32] = { 0 };
				if (fscanf(fp, "%31s %lf", q, &period) == 2 && strcmp(q, "max") != 0)
					quota = atof(q);
				fclose(fp);
			}
			else if (FILE* fq = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
				if (fscanf(fq, "%lf", &quota) != 1)
					quota = 0;
				fclose(fq);
				if (FILE* fr = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
					if (fscanf(fr, "%lf", &period) != 1)
						period = 0;
					fclose(fr);
				}
			}
			if (quota > 0 && period > 0) {
				const unsigned share = (unsigned)(quota / period * 1.5 + 0.5);
				n = share < n ? (share < 1 ? 1u : share) : n;
			}
		}
		n = n > HOST_THREADS_DEFAULT_CAP ? HOST_THREADS_DEFAULT_CAP : n;
		if (const char* e = getenv("STENOS_HOST_THREADS"))
			if (atoi(e) > 0)
				n = (unsigned)atoi(e);
		return n > 256 ? 256u : n < 1 ? 1u : n;
	}();
	return threads;
}

// Persistent workers (created on first use, joined at unload): a batch of superblocks is a few milliseconds of
// work, too little to pay for 64 thread creations each time.  One job at a time; the caller works too.
class WorkerPool {
	std::vector<std::thread> threads_;
	std::mutex job_mutex_, m_;
	std::condition_variable cv_work_, cv_done_;
	const std::function<void(uint64_t)>* fn_ = nullptr;
	uint64_t cnt_ = 0, generation_ = 0;
	std::atomic<uint64_t> next_{ 0 };
	unsigned busy_ = 0, wanted_ = 0;
	bool stop_ = false;

	void drain()
	{
		for (uint64_t k; (k = next_.fetch_add(1)) < cnt_;)
			(*fn_)(k);
	}
	void loop(unsigned id)
	{
		uint64_t seen = 0;
		std::unique_lock<std::mutex> lk(m_);
		for (;;) {
			cv_work_.wait(lk, [&] { return stop_ || generation_ != seen; });
			if (stop_)
				return;
			seen = generation_;
			if (id >= wanted_)
				continue;
			lk.unlock();
			drain();
			lk.lock();
			if (--busy_ == 0)
				cv_done_.notify_one();
		}
	}

public:
	~WorkerPool()
	{
		{
			std::lock_guard<std::mutex> lk(m_);
			stop_ = true;
		}
		cv_work_.notify_all();
		for (auto& t : threads_)
			t.join();
	}
	void run(uint64_t cnt, const std::function<void(uint64_t)>& fn)
	{
		const unsigned nthreads = host_threads();
		const unsigned helpers = (unsigned)((cnt < nthreads ? cnt : nthreads) - (cnt ? 1 : 0));
		std::lock_guard<std::mutex> job(job_mutex_);
		if (helpers == 0) {
			for (uint64_t k = 0; k < cnt; ++k)
				fn(k);
			return;
		}
		{
			std::lock_guard<std::mutex> lk(m_);
			while (threads_.size() < helpers) {
				const unsigned id = (unsigned)threads_.size();
				threads_.emplace_back([this, id] { loop(id); });
			}
			fn_ = &fn;
			cnt_ = cnt;
			next_ = 0;
			wanted_ = helpers;
			busy_ = helpers;
			++generation_;
		}
		cv_work_.notify_all();
		drain();
		std::unique_lock<std::mutex> lk(m_);
		cv_done_.wait(lk, [&] { return busy_ == 0; });
	}
};

void parallel_for(uint64_t cnt, const std::function<void(uint64_t)>& fn)
{
	static WorkerPool pool;
	pool.run(cnt, fn);
}

// Levels >= 2 and bytesoftype 1: the strategy layer of compress_generic_superblock (stenos.cpp:451-604, 617-678).
// The GPU encodes every superblock with the block codec (capacity = the superblock's own size, as the reference's
// scratch buffer), shuffles the input and prepares the plane middles for the LZ4-dry estimates; the host runs
// the estimator and zstd (third-party entropy coder) and assembles the frame in `h_dst` with the reference's
// serial capacity semantics.  h_src / d_src: host and device copies of the input.
// d_dst (device destinations): the frame is uploaded batch by batch while the next batch is in zstd; h_dst is the staging.
size_t compress_strategy(stenos_context_s* ctx, const uint8_t* h_src, const uint8_t* d_src, size_t T, size_t bytes, uint8_t* h_dst, size_t dst_size,
			 int level, const FramePlan& f, hipStream_t stream, uint8_t* d_dst = nullptr)
{
	if (!zstd().ok)
		return STENOS_ERROR_ZSTD_INTERNAL;
	if (dst_size < f.header)
		return STENOS_ERROR_DST_OVERFLOW;
	h_dst[0] = (uint8_t)f.shift;
	put_le(h_dst + 1, bytes, 7);
	if (f.header == 12)
		put_le(h_dst + 8, f.sb, 4);
	const uint64_t nblocks = f.nfull + (f.tail ? 1 : 0);
	const uint32_t stride = stenos_k_slot_stride((uint32_t)T);
	const size_t tmp_cap = bytes + 4 * (size_t)f.nsb + 64;
	if (!ctx->bsize.ensure((nblocks + 1) * 4) || !ctx->binfo.ensure((nblocks + 1) * 4) || !ctx->bneed.ensure((nblocks + 1) * 4) || !ctx->boff.ensure((nblocks + 1) * 4) ||
	    !ctx->sbcsize.ensure((f.nsb + 1) * 4) || !ctx->sbneed.ensure((f.nsb + 1) * 4) || !ctx->sbcode.ensure(f.nsb + 1) ||
	    !ctx->sboff.ensure((f.nsb + 8) * 8) || !ctx->misc.ensure(4096) || !ctx->qprod.ensure((f.nsb + 1) * 4) ||
	    !ctx->slots.ensure((nblocks + 1) * (size_t)stride) || !ctx->tmp1.ensure(tmp_cap))
		return STENOS_ERROR_ALLOC;
	uint8_t* misc = ctx->misc.as<uint8_t>();
	codec::FrameJob j;
	memset(&j, 0, sizeof(j));
	j.src = d_src;
	j.dst = ctx->tmp1.as<uint8_t>();
	j.dst_size = ~(uint64_t)0 >> 1;
	j.slots = ctx->slots.as<uint8_t>();
	j.bsize = ctx->bsize.as<uint32_t>();
	j.binfo = ctx->binfo.as<uint32_t>();
	j.bneed = ctx->bneed.as<uint32_t>();
	if (!wide_scratch(ctx, T, nblocks, &j.wide_scratch, &j.wide_scratch_bytes))
		return STENOS_ERROR_ALLOC;
	j.boff = ctx->boff.as<uint32_t>();
	j.sb_csize = ctx->sbcsize.as<uint32_t>();
	j.sb_code = ctx->sbcode.as<uint8_t>();
	j.sb_need = ctx->sbneed.as<uint32_t>();
	j.sb_off = ctx->sboff.as<uint64_t>();
	j.total = (uint64_t*)misc;
	j.status = (uint32_t*)(misc + 12);
	j.first_flagged = (uint32_t*)(misc + 16);
	j.override_payload = misc + 64;
	j.nfull = f.nfull;
	j.nsb = f.nsb;
	j.total_bytes = bytes;
	j.tail_bytes = f.tail;
	j.bps = f.bps;
	j.sb_bytes = (uint32_t)f.sb;
	j.slot_stride = stride;
	j.T = (uint32_t)T;
	j.shift_byte = 0xFFFFFFFFu;
	j.fixed_capacity = 1;
	j.qprod = ctx->qprod.as<uint32_t>();
	uint64_t* d_carry = (uint64_t*)(misc + 24);
	// The block codec's verdict per superblock first (sizes only: nothing is packed or moved yet) ...
	if (stenos_k_launch_init(misc, 0, nullptr, 0, nullptr, 0, stream) != hipSuccess || stenos_k_launch_encode(j, 0, nblocks, stream) != hipSuccess || stenos_k_launch_plan(j, 0, f.nsb, stream) != hipSuccess)
		return STENOS_ERROR_UNDEFINED;
	std::vector<uint8_t> code(f.nsb);
	std::vector<uint32_t> csize(f.nsb), qprod(f.nsb);
	std::vector<uint64_t> sboff(f.nsb + 1);
	if (hipMemcpyAsync(code.data(), j.sb_code, f.nsb, hipMemcpyDeviceToHost, stream) != hipSuccess ||
	    hipMemcpyAsync(csize.data(), j.sb_csize, f.nsb * 4, hipMemcpyDeviceToHost, stream) != hipSuccess ||
	    hipMemcpyAsync(qprod.data(), j.qprod, f.nsb * 4, hipMemcpyDeviceToHost, stream) != hipSuccess)
		return STENOS_ERROR_UNDEFINED;
	PhaseTrace trace(ctx->stage_ms);
	// ... and, for an input that lives on the device (h_src == NULL), the part of it the estimator looks at: the first
	// 1/16 of every superblock (stenos.cpp:497-499), one strided copy into a host image of the input.  The rest of a
	// superblock is fetched only if it ends up going through zstd as it is (or as a copy).
	const bool lazy_src = h_src == nullptr;
	if (lazy_src) {
		if (!ctx->h_in.ensure(bytes + 64))
			return STENOS_ERROR_ALLOC;
		uint8_t* img = ctx->h_in.data();
		const uint64_t whole = bytes / f.sb;
		if (whole && f.sb / 16 &&
		    hipMemcpy2DAsync(img, f.sb, d_src, f.sb, f.sb / 16, (size_t)whole, hipMemcpyDeviceToHost, stream) != hipSuccess)
			return STENOS_ERROR_UNDEFINED;
		if (bytes > whole * f.sb && hipMemcpyAsync(img + whole * f.sb, d_src + whole * f.sb, bytes - whole * f.sb, hipMemcpyDeviceToHost, stream) != hipSuccess)
			return STENOS_ERROR_UNDEFINED; // (the last, partial superblock: all of it)
		h_src = img;
	}
	if (hipStreamSynchronize(stream) != hipSuccess)
		return STENOS_ERROR_UNDEFINED;
	trace.mark("verdicts and samples to host", STAGE_GPU_PASS);
	HostBuf& blocks = ctx->h_blocks;
	// transposed views for the estimator and the transposed zstd strategies (levels > 2 only, stenos.cpp:515-537)
	const bool transposed = T > 1 && level > 2;
	HostBuf &shuf = ctx->h_shuf, &mid0 = ctx->h_mid0, &mid1 = ctx->h_mid1;
	if (transposed) {
		if (!ctx->shuf.ensure(bytes + 64) || !ctx->mid0.ensure(bytes + 64) || !ctx->mid1.ensure(bytes + 64))
			return STENOS_ERROR_ALLOC;
		if (stenos_k_launch_shuffle_superblocks(d_src, ctx->shuf.as<uint8_t>(), (uint32_t)T, f.sb, bytes, stream) != hipSuccess ||
		    stenos_k_launch_delta_middles(ctx->shuf.as<uint8_t>(), ctx->mid0.as<uint8_t>(), (uint32_t)T, f.sb, bytes, (uint32_t)level, false, stream) != hipSuccess ||
		    stenos_k_launch_delta_middles(ctx->shuf.as<uint8_t>(), ctx->mid1.as<uint8_t>(), (uint32_t)T, f.sb, bytes, (uint32_t)level, true, stream) != hipSuccess)
			return STENOS_ERROR_UNDEFINED;
		if (!shuf.ensure(bytes + 64) || !mid0.ensure(bytes + 64) || !mid1.ensure(bytes + 64))
			return STENOS_ERROR_ALLOC;
		if (hipMemcpy(shuf.data(), ctx->shuf.p, bytes, hipMemcpyDeviceToHost) != hipSuccess ||
		    hipMemcpy(mid0.data(), ctx->mid0.p, bytes, hipMemcpyDeviceToHost) != hipSuccess ||
		    hipMemcpy(mid1.data(), ctx->mid1.p, bytes, hipMemcpyDeviceToHost) != hipSuccess)
			return STENOS_ERROR_UNDEFINED;
		trace.mark("transposed views to host", STAGE_GPU_PASS);
	}

	int zstd_level = level; // stenos.cpp:441-460
	if (T > 1) {
		zstd_level = level - 1;
		if (zstd_level >= 4)
			++zstd_level;
	}
	const int zl = zstd_level_of(zstd_level);
	const size_t bs = 256 * T;
	// What a superblock becomes is decided first (it does not depend on the room left in the destination):
	//   0 = tiny input, plain zstd level 1 (stenos.cpp:435-437)      1 = block codec + zstd (code 5, or 1)
	//   2/3/4 = zstd over the raw / transposed / transposed+delta bytes (stenos.cpp:548-558)
	auto decide = [&](uint64_t s) -> int {
		const uint8_t* src = h_src + s * f.sb;
		const size_t sbytes = (size_t)((bytes - s * f.sb) < f.sb ? (bytes - s * f.sb) : f.sb);
		if (sbytes < 128)
			return 0;
		double lz_ratio = 1.1, lz_tr = 0, lz_trd = 0;
		if (sbytes >= bs)
			lz_ratio = (double)(sbytes / 16) / (double)strategy::lz4_dry_size(src, sbytes / 16, 10 - level);
		if (T > 1) {
			if (transposed && sbytes >= bs) {
				const size_t step = strategy::middle_step(T, sbytes, level);
				lz_tr = strategy::transposed_ratio(mid0.data() + s * f.sb, T, step, level);
				if (lz_tr > lz_ratio)
					lz_ratio = lz_tr;
				lz_trd = strategy::transposed_ratio(mid1.data() + s * f.sb, T, step, level) * 1.1;
				if (lz_trd > lz_ratio)
					lz_ratio = lz_trd;
				const double factor = 1. + level / 12.;
				lz_tr *= factor;
				lz_trd *= factor;
				lz_ratio *= factor;
			}
		}
		else
			lz_ratio *= 1. + level / 12.;
		// block codec result of the GPU; the reference gives up when, after 1/16 of the input, the running
		// ratio is below the estimate (block_compress.h:1266-1274)
		bool ok = code[s] == 1;
		if (ok && qprod[s]) {
			size_t bq = (sbytes / 16 + bs - 1) / bs;
			bq = bq == 0 ? 0 : bq - 1;
			const double ratio = (double)((bq + 1) * bs) / (double)qprod[s];
			if (ratio < lz_ratio)
				ok = false;
		}
		if (ok)
			return 1;
		int c = 2; // stenos.cpp:548-558
		if (lz_ratio > 1.40) {
			if (lz_ratio == lz_tr)
				c = 3;
			else if (lz_ratio == lz_trd)
				c = 4;
		}
		return c;
	};

	// One superblock -> [code][csize:3][payload] at `out` with `room` bytes of capacity (what the reference
	// hands to its strategies, stenos.cpp:895).  `delta_src`: the GPU's byte delta of the transposed
	// superblock for choice 4.  Returns the bytes written or an error code.
	auto emit = [&](uint64_t s, int choice, const uint8_t* delta_src, uint8_t* out, size_t room) -> size_t {
		const uint8_t* src = h_src + s * f.sb;
		const size_t sbytes = (size_t)((bytes - s * f.sb) < f.sb ? (bytes - s * f.sb) : f.sb);
		size_t r;
		if (choice == 1) {
			const uint8_t* payload = blocks.data() + sboff[s] + 4;
			const size_t cblock = csize[s];
			r = zstd().compress(out + 4, room - 4, payload, cblock, zl); // stenos.cpp:583
			if (zstd().is_error(r) || r > cblock) {                     // NO_ZSTD (:585-596)
				if (room < 4 + cblock)
					return (size_t)STENOS_ERROR_DST_OVERFLOW;
				out[0] = 1;
				put_le(out + 1, cblock, 3);
				memcpy(out + 4, payload, cblock);
				return cblock + 4;
			}
			out[0] = 5;
			put_le(out + 1, r, 3);
			return r + 4;
		}
		const uint8_t* zsrc = choice == 3 ? shuf.data() + s * f.sb : choice == 4 ? delta_src : src;
		r = zstd().compress(out + 4, room - 4, zsrc, sbytes, choice == 0 ? 1 : zl);
		if (zstd().is_error(r) || r > sbytes)
			return host_copy_superblock(src, sbytes, out, room);
		out[0] = (uint8_t)(choice == 0 ? 2 : choice);
		put_le(out + 1, r, 3);
		return r + 4;
	};

	// What every superblock becomes, then only the block streams that are kept are packed and brought to the host: the
	// reference abandons the block codec for a superblock after 1/16 of it when the ratio target fails
	// (block_compress.h:1266-1274) -- here the verdict comes from the sizes, and an abandoned superblock costs neither a
	// pack nor a transfer.
	std::vector<int> all_choice(f.nsb);
	parallel_for(f.nsb, [&](uint64_t s) { all_choice[s] = decide(s); });
	trace.mark("estimates", STAGE_ESTIMATES);
	{
		std::vector<uint8_t> keep(f.nsb);
		uint64_t dropped = 0;
		for (uint64_t s = 0; s < f.nsb; ++s) {
			keep[s] = all_choice[s] == 1;
			dropped += code[s] == 1 && !keep[s];
		}
		if (dropped) {
			if (!ctx->tmp2.ensure(f.nsb + 64) || hipMemcpyAsync(ctx->tmp2.p, keep.data(), f.nsb, hipMemcpyHostToDevice, stream) != hipSuccess ||
			    stenos_k_launch_keep_superblocks(ctx->tmp2.as<uint8_t>(), j.sb_code, j.sb_csize, f.nsb, stream) != hipSuccess)
				return STENOS_ERROR_UNDEFINED;
		}
		if (stenos_k_launch_scan(j, 0, f.nsb, d_carry, stream) != hipSuccess || stenos_k_launch_pack(j, 0, f.nsb, stream) != hipSuccess ||
		    hipMemcpyAsync(sboff.data(), j.sb_off, (f.nsb + 1) * 8, hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess)
			return STENOS_ERROR_UNDEFINED;
		const size_t blocks_size = (size_t)sboff[f.nsb];
		if (!blocks.ensure(blocks_size + 64))
			return STENOS_ERROR_ALLOC;
		trace.mark("pack", STAGE_GPU_PASS);
	}
	// The block streams come to the host batch by batch on a stream of their own, in the order the batches are compressed:
	// the transfer of batch k + 1 runs while batch k is in zstd (the link moves 50 GB/s, sixteen cores' zstd a quarter of
	// that), and so does the upload of the finished part of the frame when the destination is device memory.
	const size_t ample = (4 + f.sb + f.sb / 128 + 1024 + 15) & ~(size_t)15; // (a multiple of 16: the gather kernel reads the slots with aligned 16-byte loads)
	uint64_t batch = ((size_t)256 << 20) / f.sb;
	batch = batch < 64 ? 64 : batch > 1024 ? 1024 : batch;
	const uint64_t nbatch = (f.nsb + batch - 1) / batch;
	if (!ctx->copy_stream && hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking) != hipSuccess)
		return STENOS_ERROR_ALLOC;
	if (d_dst && !ctx->upload_stream && hipStreamCreateWithFlags(&ctx->upload_stream, hipStreamNonBlocking) != hipSuccess)
		return STENOS_ERROR_ALLOC;
	hipStream_t copy = ctx->copy_stream;                          // device -> host: block streams, in batch order
	hipStream_t up = d_dst ? ctx->upload_stream : ctx->copy_stream; // host -> device: the frame (not queued behind the downloads)
	struct Drain { // (whatever way this function is left, no transfer of this call is still in flight)
		hipStream_t a, b;
		~Drain()
		{
			(void)hipStreamSynchronize(a);
			(void)hipStreamSynchronize(b);
		}
	} drain = { copy, up };
	while (ctx->batch_ev.size() < nbatch) {
		hipEvent_t e;
		if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess)
			return STENOS_ERROR_ALLOC;
		ctx->batch_ev.push_back(e);
	}
	{
		if (lazy_src) { // raw bytes of the superblocks that go through zstd as they are (runs of neighbours in one copy), first
			uint8_t* img = ctx->h_in.data();
			for (uint64_t s = 0; s < f.nsb;) {
				if (all_choice[s] == 1) {
					++s;
					continue;
				}
				uint64_t e = s;
				while (e < f.nsb && all_choice[e] != 1)
					++e;
				const size_t b0 = (size_t)(s * f.sb), b1 = (size_t)(e * f.sb < bytes ? e * f.sb : bytes);
				if (hipMemcpyAsync(img + b0, d_src + b0, b1 - b0, hipMemcpyDeviceToHost, copy) != hipSuccess)
					return STENOS_ERROR_UNDEFINED;
				s = e;
			}
		}
		for (uint64_t b = 0; b < nbatch; ++b) {
			const uint64_t s0 = b * batch, s1 = s0 + batch < f.nsb ? s0 + batch : f.nsb;
			const size_t lo = (size_t)sboff[s0], hi = (size_t)sboff[s1];
			if ((hi > lo && hipMemcpyAsync(blocks.data() + lo, j.dst + lo, hi - lo, hipMemcpyDeviceToHost, copy) != hipSuccess) ||
			    hipEventRecord(ctx->batch_ev[b], copy) != hipSuccess)
				return STENOS_ERROR_UNDEFINED;
		}
	}
	// zstd's result depends on the capacity only below ZSTD_compressBound of its input, so the superblocks of a
	// batch are compressed in parallel into roomy scratch buffers and then laid out in order; a superblock that
	// meets less room than that in the caller's buffer (the end of a tight buffer) is redone with the exact
	// capacity, as the reference's serial loop would have seen it.
	// Device destinations: the batch's slots go to the device as they are (one transfer, beside the next batch's zstd) and a
	// kernel puts every superblock at its place in the frame: the host's threads do not touch the bytes again.  Two sets of
	// slots, so that a batch can be compressed while the one before it is on its way.
	std::vector<int> choice;
	const size_t set_slots = (size_t)(batch < f.nsb ? batch : f.nsb) * ample;
	const int nsets = d_dst ? 2 : 1;
	if (!ctx->h_stage.ensure((size_t)nsets * set_slots)) // (kept by the context: a fresh 100 MB allocation per call costs more than the zstd calls)
		return STENOS_ERROR_ALLOC;
	const size_t tab_bytes = ((size_t)batch * 16 + 63) & ~(size_t)63; // per set: offsets, sizes (uint64 each)
	if (d_dst && (!ctx->dslots.ensure(2 * set_slots + 64) || !ctx->dtab.ensure(2 * tab_bytes + 64) || !ctx->h_tab.ensure(2 * tab_bytes + 64)))
		return STENOS_ERROR_ALLOC;
	while (d_dst && ctx->set_ev.size() < 2) {
		hipEvent_t e;
		if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess)
			return STENOS_ERROR_ALLOC;
		ctx->set_ev.push_back(e);
	}
	bool set_busy[2] = { false, false };
	struct { uint8_t* p; uint8_t* get() const { return p; } } scratch = { ctx->h_stage.data() };
	std::vector<size_t> sizes;
	std::vector<uint64_t> dslot; // position of a choice-4 superblock in the batch's delta buffer
	std::vector<size_t> offsets;
	std::vector<uint8_t> deltas;
	size_t off = f.header;
	auto upload = [&](size_t lo, size_t hi) -> bool { // frame bytes [lo, hi) that the host laid out in h_dst
		return !d_dst || hi <= lo || hipMemcpyAsync(d_dst + lo, h_dst + lo, hi - lo, hipMemcpyHostToDevice, up) == hipSuccess;
	};
	uint64_t nb = 0;
	for (uint64_t s0 = 0; s0 < f.nsb; s0 += batch, ++nb) {
		const uint64_t cnt = (s0 + batch < f.nsb ? s0 + batch : f.nsb) - s0;
		choice.assign(all_choice.begin() + (ptrdiff_t)s0, all_choice.begin() + (ptrdiff_t)(s0 + cnt));
		const int set = d_dst ? (int)(nb & 1) : 0;
		scratch.p = ctx->h_stage.data() + (size_t)set * set_slots;
		if (set_busy[set]) { // the slots of this set are still on their way to the device (two batches ago)
			if (hipEventSynchronize(ctx->set_ev[(size_t)set]) != hipSuccess)
				return STENOS_ERROR_UNDEFINED;
			set_busy[set] = false;
			trace.mark("slots to device", STAGE_UPLOAD);
		}
		if (hipEventSynchronize(ctx->batch_ev[s0 / batch]) != hipSuccess) // the batch's block streams (and the raw superblocks) are on the host
			return STENOS_ERROR_UNDEFINED;
		trace.mark("block streams to host", STAGE_BLOCKS_TO_HOST);
#ifdef STENOS_HOST_TRACE
		{
			unsigned h[5] = { 0, 0, 0, 0, 0 };
			for (uint64_t k = 0; k < cnt; ++k)
				++h[choice[k]];
			fprintf(stderr, "[stenos]   superblocks %llu: tiny %u, block codec %u, zstd %u, transposed %u, transposed+delta %u\n", (unsigned long long)cnt, h[0], h[1],
				h[2], h[3], h[4]);
		}
#endif
		// byte delta of the whole transposed superblock on the GPU for the choice-4 ones (stenos.cpp:646)
		dslot.assign(cnt, 0);
		uint64_t nd = 0;
		for (uint64_t k = 0; k < cnt; ++k)
			if (choice[k] == 4)
				dslot[k] = nd++;
		if (nd) {
			if (!ctx->tmp2.ensure(nd * f.sb + 64))
				return STENOS_ERROR_ALLOC;
			deltas.resize(nd * f.sb);
			for (uint64_t k = 0; k < cnt; ++k)
				if (choice[k] == 4) {
					const uint64_t s = s0 + k;
					const size_t sbytes = (size_t)((bytes - s * f.sb) < f.sb ? (bytes - s * f.sb) : f.sb);
					if (stenos_k_launch_delta(ctx->shuf.as<uint8_t>() + s * f.sb, ctx->tmp2.as<uint8_t>() + dslot[k] * f.sb, sbytes, false, stream) !=
					    hipSuccess)
						return STENOS_ERROR_UNDEFINED;
				}
			if (hipMemcpyAsync(deltas.data(), ctx->tmp2.p, nd * f.sb, hipMemcpyDeviceToHost, stream) != hipSuccess ||
			    hipStreamSynchronize(stream) != hipSuccess)
				return STENOS_ERROR_UNDEFINED;
		}
		sizes.assign(cnt, 0);
		parallel_for(cnt, [&](uint64_t k) {
			sizes[k] = emit(s0 + k, choice[k], deltas.data() + dslot[k] * f.sb, scratch.get() + k * ample, ample);
		});
		trace.mark("zstd", STAGE_ZSTD);
		// Layout in order.  When even the last superblock of the batch finds ample room (the usual case), the
		// offsets are a plain prefix sum and the copies run on the worker threads.
		{
			size_t end = off;
			bool plain = true;
			offsets.resize(cnt);
			for (uint64_t k = 0; k < cnt && plain; ++k) {
				offsets[k] = end;
				plain = !is_err(sizes[k]) && dst_size >= end + ample;
				end += plain ? sizes[k] : 0;
			}
			if (plain && d_dst) { // laid out on the device
				uint64_t* tab = (uint64_t*)(ctx->h_tab.data() + (size_t)set * tab_bytes);
				for (uint64_t k = 0; k < cnt; ++k) {
					tab[k] = offsets[k];
					tab[batch + k] = sizes[k];
				}
				uint8_t* d_slots = ctx->dslots.as<uint8_t>() + (size_t)set * set_slots;
				uint64_t* d_tab = (uint64_t*)(ctx->dtab.as<uint8_t>() + (size_t)set * tab_bytes);
				if (hipMemcpyAsync(d_slots, scratch.get(), (size_t)cnt * ample, hipMemcpyHostToDevice, up) != hipSuccess ||
				    hipMemcpyAsync(d_tab, tab, (size_t)batch * 16, hipMemcpyHostToDevice, up) != hipSuccess ||
				    stenos_k_launch_gather_pieces(d_slots, ample, d_tab, d_tab + batch, (uint32_t)cnt, d_dst, up) != hipSuccess ||
				    hipEventRecord(ctx->set_ev[(size_t)set], up) != hipSuccess)
					return STENOS_ERROR_UNDEFINED;
				set_busy[set] = true;
				off = end;
				trace.mark("layout", STAGE_LAYOUT);
				continue;
			}
			if (plain) {
				parallel_for(cnt, [&](uint64_t k) { memcpy(h_dst + offsets[k], scratch.get() + k * ample, sizes[k]); });
				off = end;
				trace.mark("layout", STAGE_LAYOUT);
				continue;
			}
		}
		const size_t batch_begin = off;
		for (uint64_t k = 0; k < cnt; ++k) {
			if (dst_size < off + 4) // stenos.cpp:427-429
				return STENOS_ERROR_DST_OVERFLOW;
			const size_t room = dst_size - off;
			size_t r = sizes[k];
			if (room >= ample) {
				if (!is_err(r))
					memcpy(h_dst + off, scratch.get() + k * ample, r);
			}
			else
				r = emit(s0 + k, choice[k], deltas.data() + dslot[k] * f.sb, h_dst + off, room);
			if (is_err(r))
				return r;
			off += r;
		}
		trace.mark("layout", STAGE_LAYOUT);
		if (!upload(batch_begin, off)) // (a batch the host laid out itself: the end of a tight destination)
			return STENOS_ERROR_UNDEFINED;
	}
	if (d_dst) { // the frame header last; then everything has to be there
		if (!upload(0, f.header) || hipStreamSynchronize(up) != hipSuccess)
			return STENOS_ERROR_UNDEFINED;
		trace.mark("upload", STAGE_UPLOAD);
	}
	return off;
}

size_t compress_device(stenos_context_s* ctx, const void* d_src, size_t T, size_t bytes, void* d_dst, size_t dst_size, hipStream_t stream, bool wait)
{
	if (!ctx->device_ready())
		return STENOS_ERROR_INVALID_INSTRUCTION_SET;
	const int level = ctx->level;
	if (ctx->max_nanoseconds) // the time limit is a feature of the host-pointer ABI (compress_timed); these entry points take none
		return STENOS_ERROR_INVALID_PARAMETER;
	FramePlan f;
	size_t e = plan_frame(ctx, T, bytes, level, f);
	if (is_err(e))
		return e;
	e = check_supported(ctx, T, level);
	if (is_err(e))
		return e;
	if (dst_size < f.header) // stenos.cpp:862-863, 870-871
		return STENOS_ERROR_DST_OVERFLOW;
	ctx->job_kind = 0;
	if (bytes && needs_strategy(T, level)) {
		// the strategy layer needs the input on the host (estimator, zstd): fetch it, assemble the frame there
		const size_t roomy = f.header + f.nsb * 4 + bytes + f.sb / 128 + 4096; // beyond the largest frame (all copies) + ZSTD_compressBound's margin the capacity no longer matters
		const size_t cap = dst_size < roomy ? dst_size : roomy;
		HostBuf& h_out = ctx->h_out;
		if (!h_out.ensure(cap + 64))
			return STENOS_ERROR_ALLOC;
		// (no host copy of the input: the strategy layer fetches what it looks at)
		size_t r = compress_strategy(ctx, nullptr, (const uint8_t*)d_src, T, bytes, h_out.data(), cap, level, f, stream, (uint8_t*)d_dst);
		if (is_err(r))
			return r;
		ctx->last_nsb = 0; // no device-side index for these frames
		ctx->h_total[0] = r;
		ctx->h_total[1] = 0;
		ctx->job_kind = 1;
		ctx->job_stream = stream;
		ctx->job_dst_size = dst_size;
		return wait ? finish_job(ctx) : 0;
	}
	if (bytes == 0) { // stenos.cpp:876-878
		uint8_t h[12];
		h[0] = (uint8_t)f.shift;
		put_le(h + 1, 0, 7);
		put_le(h + 8, f.sb, 4);
		if (hipMemcpyAsync(d_dst, h, f.header, hipMemcpyHostToDevice, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess)
			return STENOS_ERROR_UNDEFINED;
		ctx->last_nsb = 0;
		ctx->h_total[0] = f.header;
		ctx->h_total[1] = 0;
		ctx->job_kind = 1;
		ctx->job_stream = stream;
		ctx->job_dst_size = dst_size;
		return f.header;
	}
	e = enqueue_compress(ctx, (const uint8_t*)d_src, T, bytes, (uint8_t*)d_dst, dst_size, level, f, true, stream);
	if (is_err(e))
		return e;
	ctx->job_kind = 1;
	ctx->job_stream = stream;
	ctx->job_dst_size = dst_size;
	ctx->job_src = d_src;
	ctx->job_dst = d_dst;
	ctx->job_T = T;
	ctx->job_bytes = bytes;
	return wait ? finish_job(ctx) : 0;
}

struct FrameInfo {
	uint64_t total = 0;
	size_t sb = 0, header = 0;
	uint64_t nsb = 0;
};
// frame header checks of stenos_decompress_generic (stenos.cpp:1066-1116); returns 0 or an error code
size_t parse_frame(const uint8_t* h, size_t have, size_t T, size_t dst_size, FrameInfo& fi)
{
	if (T == 0 || T >= STENOS_MAX_BYTESOFTYPE)
		return STENOS_ERROR_INVALID_BYTESOFTYPE;
	if (have < 8)
		return STENOS_ERROR_SRC_OVERFLOW;
	const unsigned shift = h[0];
	if (shift > 4 && shift != 255)
		return STENOS_ERROR_INVALID_INPUT;
	fi.total = get_le(h + 1, 7);
	if (fi.total > dst_size)
		return STENOS_ERROR_DST_OVERFLOW;
	fi.header = 8;
	if (fi.total == 0)
		return 0;
	if (shift == 255) {
		if (have < 12)
			return STENOS_ERROR_SRC_OVERFLOW;
		fi.sb = (size_t)get_le(h + 8, 4);
		fi.header = 12;
		// what the compressor can have written (prepare(): a whole number of blocks' worth, below STENOS_MAX_BLOCK_BYTES);
		// the reference trusts the field (stenos.cpp:1098-1103) and would divide by zero or size buffers from garbage
		if (fi.sb < T * 256 || fi.sb >= STENOS_MAX_BLOCK_BYTES)
			return STENOS_ERROR_INVALID_INPUT;
	}
	else
		fi.sb = base_superblock(T * 256) << shift;
	// Unlike the reference (stenos.cpp:1115-1116, 1131) the last superblock of a frame whose size is an
	// exact multiple of the superblock size is decoded with its full size instead of 0 bytes.
	fi.nsb = fi.total / fi.sb + (fi.total % fi.sb ? 1 : 0);
	if (fi.nsb > 0x7FFFFFFFull) // one workgroup per superblock: beyond the grid limit (256 TiB of int32)
		return STENOS_ERROR_INVALID_PARAMETER;
	return 0;
}

// Finish the superblocks whose payload went through zstd (codes 2-5, decompress_generic_superblock,
// stenos.cpp:694-740): zstd itself runs on the host (third-party entropy coder, dlopen'ed), the byte kernels and
// the block decoder that follow it run on the device.  h_index: nsb + 1 header offsets on the host; h_frame: host
// copy of the frame or NULL (then the headers and payloads are fetched from the device).
size_t finish_host_codes(stenos_context_s* ctx, const uint8_t* d_frame, const uint8_t* h_frame, size_t size, size_t T, const uint64_t* h_index,
			 const FrameInfo& fi, uint8_t* d_dst, hipStream_t stream)
{
	PhaseTrace trace(ctx->stage_ms);
	// A frame that lives on the device comes to the host in pieces, on a stream of its own: the threads inflate the
	// superblocks of the first pieces while the rest is still on the link (the frame of 8 GiB of bytes at level 3 is 4 GB:
	// 80 ms of link time, as much as half the inflation).
	constexpr size_t PIECE = (size_t)64 << 20;
	const size_t pieces = h_frame ? 0 : (size + PIECE - 1) / PIECE;
	size_t pieces_here = 0;
	if (!h_frame) {
		HostBuf& frame_copy = ctx->h_in;
		if (!frame_copy.ensure(size + 64))
			return STENOS_ERROR_ALLOC;
		if (!ctx->copy_stream && hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking) != hipSuccess)
			return STENOS_ERROR_ALLOC;
		while (ctx->set_ev.size() < pieces) {
			hipEvent_t e;
			if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess)
				return STENOS_ERROR_ALLOC;
			ctx->set_ev.push_back(e);
		}
		h_frame = frame_copy.data();
	}
	// (the caller's stream has been waited for: the frame is complete on the device.  Only a few pieces are queued ahead of
	// the one being read: the copies of the inflated batches to the device wait behind whatever the other direction has queued)
	constexpr size_t AHEAD = 4;
	size_t pieces_queued = 0;
	auto queue_pieces = [&](size_t upto) -> bool {
		for (; pieces_queued < pieces && pieces_queued < upto; ++pieces_queued) {
			const size_t at = pieces_queued * PIECE, n = size - at < PIECE ? size - at : PIECE;
			if (hipMemcpyAsync(ctx->h_in.data() + at, d_frame + at, n, hipMemcpyDeviceToHost, ctx->copy_stream) != hipSuccess ||
			    hipEventRecord(ctx->set_ev[pieces_queued], ctx->copy_stream) != hipSuccess)
				return false;
		}
		return true;
	};
	auto frame_here = [&](size_t upto) -> bool { // the first `upto` bytes of the frame are on the host
		while (pieces_here < pieces && pieces_here * PIECE < upto) {
			if (!queue_pieces(pieces_here + 1 + AHEAD) || hipEventSynchronize(ctx->set_ev[pieces_here]) != hipSuccess)
				return false;
			++pieces_here;
		}
		return true;
	};
	struct Item {
		uint64_t s;
		uint32_t code;
		size_t csize, dsize, r;
	};
	std::vector<Item> items;
	uint64_t next_sb = 0;
	// the next (up to) `want` superblocks that went through zstd, in frame order; 0 or an error code
	auto collect = [&](size_t want) -> size_t {
		items.clear();
		for (; next_sb < fi.nsb && items.size() < want; ++next_sb) {
			const uint64_t s = next_sb;
			if (h_index[s] + 4 > size)
				return STENOS_ERROR_SRC_OVERFLOW;
			if (!frame_here(h_index[s] + 4))
				return STENOS_ERROR_UNDEFINED;
			const uint8_t* hd = h_frame + h_index[s];
			const unsigned code = hd[0];
			if (code == 1 || code == 6)
				continue;
			if (code < 2 || code > 5)
				return STENOS_ERROR_INVALID_INPUT;
			const size_t csize = (size_t)get_le(hd + 1, 3);
			const uint64_t begin = s * (uint64_t)fi.sb;
			const size_t dsize = (size_t)((fi.total - begin) < fi.sb ? (fi.total - begin) : fi.sb);
			if (h_index[s] + 4 + csize > size)
				return STENOS_ERROR_INVALID_INPUT;
			if (!frame_here(h_index[s] + 4 + csize))
				return STENOS_ERROR_UNDEFINED;
			items.push_back({ s, code, csize, dsize, 0 });
		}
		return 0;
	};
	auto drain_frame = [&]() {
		if (pieces)
			(void)hipStreamSynchronize(ctx->copy_stream);
	};
	if (!zstd().ok) {
		// (only an error if a superblock needs it)
		size_t e = collect(1);
		drain_frame();
		return e ? e : items.empty() ? 0 : (size_t)STENOS_ERROR_ZSTD_INTERNAL;
	}

	// The superblocks are inflated by the worker threads into one staging buffer per batch (slot k: 12 spare bytes,
	// a [1][size:3] header for code 5, the bytes at +16), moved to the device in one copy and finished there.  Four sets of
	// buffers, each with a stream of its own: a batch is a few hundred superblocks, one wave each in the block decoder, which
	// is far from filling the device -- what a batch costs there is latency, and the batches of different sets overlap (the
	// copy of one beside the kernels of two others) while the threads inflate the next.
	constexpr int NSETS = 4;
	const size_t slot = (((size_t)fi.sb + 64 + 15) & ~(size_t)15) + 16;
	uint64_t batch = ((size_t)128 << 20) / slot;
	batch = batch < 64 ? 64 : batch > 1024 ? 1024 : batch;
	if (batch > fi.nsb)
		batch = fi.nsb;
	const size_t set_bytes = (batch * slot + 63) & ~(size_t)63, set_ids = (batch * 4 + 63) & ~(size_t)63, set_idx = (batch * 8 + 63) & ~(size_t)63;
	if (!ctx->tmp1.ensure(NSETS * set_bytes + 64) || !ctx->tmp2.ensure(NSETS * set_bytes + 64) || !ctx->bsize.ensure(NSETS * set_ids + 64) ||
	    !ctx->binfo.ensure(NSETS * set_idx + 64) || !ctx->misc.ensure(4096))
		return drain_frame(), STENOS_ERROR_ALLOC;
	HostBuf& stage = ctx->h_stage;
	// (behind the sets of slots: the superblock numbers and slot offsets of each batch, page-locked like the slots)
	const size_t tab_off = NSETS * set_bytes + 64;
	if (!stage.ensure(tab_off + NSETS * (set_ids + set_idx) + 64))
		return drain_frame(), STENOS_ERROR_ALLOC;
	while (ctx->batch_ev.size() < NSETS + 1) {
		hipEvent_t e;
		if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess)
			return drain_frame(), STENOS_ERROR_ALLOC;
		ctx->batch_ev.push_back(e);
	}
	// (bytesoftype above 64 decodes through one scratch area, wide_scratch(): its batches stay in line on the caller's stream)
	const bool one_stream = T > STENOS_K_LDS_MAX_T;
	while (!one_stream && ctx->set_streams.size() < NSETS) {
		hipStream_t st;
		if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess)
			return drain_frame(), STENOS_ERROR_ALLOC;
		ctx->set_streams.push_back(st);
	}
	auto sync_all = [&]() {
		drain_frame();
		(void)hipStreamSynchronize(stream);
		for (hipStream_t st : ctx->set_streams)
			(void)hipStreamSynchronize(st);
	};
	// what the caller's stream has queued (the block-coded superblocks of this frame, whatever wrote the frame) comes first
	if (!one_stream) {
		hipEvent_t start = ctx->batch_ev[NSETS];
		if (hipEventRecord(start, stream) != hipSuccess)
			return drain_frame(), STENOS_ERROR_UNDEFINED;
		for (hipStream_t st : ctx->set_streams)
			if (hipStreamWaitEvent(st, start, 0) != hipSuccess)
				return drain_frame(), STENOS_ERROR_UNDEFINED;
	}
	std::vector<uint32_t> ids[NSETS];
	std::vector<uint64_t> idx[NSETS];
	volatile uint32_t* h_status = (volatile uint32_t*)((uint8_t*)ctx->h_total + 40); // (page-locked: the device writes it)
	bool pending[NSETS];
	for (int set = 0; set < NSETS; ++set) {
		h_status[set] = 0;
		pending[set] = false;
	}
	auto settle = [&](int set) -> size_t { // the batch that used this set of buffers is through
		if (!pending[set])
			return 0;
		pending[set] = false;
		if (hipEventSynchronize(ctx->batch_ev[(size_t)set]) != hipSuccess)
			return STENOS_ERROR_UNDEFINED;
		return h_status[set] ? (size_t)STENOS_ERROR_INVALID_INPUT : 0;
	};
	for (size_t nbatch = 0;; ++nbatch) {
		if (size_t e = collect((size_t)batch)) {
			sync_all();
			return e;
		}
		if (items.empty())
			break;
		constexpr size_t i0 = 0;
		const int set = (int)(nbatch % NSETS);
		hipStream_t const qs = one_stream ? stream : ctx->set_streams[(size_t)set];
		const size_t cnt = items.size();
		if (size_t e = settle(set)) {
			sync_all();
			return e;
		}
		trace.mark("device finish", STAGE_DEVICE_FINISH);
		uint8_t* const hs = stage.data() + (size_t)set * set_bytes;
		uint8_t* const t1 = ctx->tmp1.as<uint8_t>() + (size_t)set * set_bytes;
		uint8_t* const t2 = ctx->tmp2.as<uint8_t>() + (size_t)set * set_bytes;
		uint32_t* const d_ids = (uint32_t*)(ctx->bsize.as<uint8_t>() + (size_t)set * set_ids);
		uint64_t* const d_idx = (uint64_t*)(ctx->binfo.as<uint8_t>() + (size_t)set * set_idx);
		uint32_t* const d_status = (uint32_t*)(ctx->misc.as<uint8_t>() + 328 + 4 * set);
		parallel_for(cnt, [&](uint64_t k) {
			Item& it = items[i0 + k];
			// code 5: zstd over the block stream, at most the superblock size (stenos.cpp:732)
			const size_t cap = it.code == 5 ? (size_t)fi.sb + 64 : it.dsize;
			it.r = zstd().decompress(hs + k * slot + 16, cap, h_frame + h_index[it.s] + 4, it.csize);
		});
		trace.mark("zstd inflate", STAGE_INFLATE);
		ids[set].clear();
		idx[set].clear();
		for (size_t k = 0; k < cnt; ++k) {
			const Item& it = items[i0 + k];
			if (zstd().is_error(it.r) || (it.code != 5 && it.code != 2 && it.r != it.dsize)) { // stenos.cpp:696-698, 706-708, 718-720
				sync_all();
				return STENOS_ERROR_INVALID_INPUT;
			}
			if (it.code == 5) { // -> one BLOCK superblock for the block decoder (stenos.cpp:726-740)
				uint8_t* h4 = hs + k * slot + 12;
				h4[0] = 1;
				put_le(h4 + 1, it.r, 3);
				ids[set].push_back((uint32_t)it.s);
				idx[set].push_back(k * slot + 12);
			}
		}
		// Only the part of the slots that is in use goes up: an inflated block stream is about half its 256 KiB slot, and the
		// link is what the device's side of a batch waits for.  One strided copy (rows of the widest item, a slot apart).
		size_t width = 0;
		for (size_t k = 0; k < cnt; ++k) {
			const Item& it = items[i0 + k];
			const size_t w = 16 + (it.code == 5 ? it.r : it.dsize);
			width = w > width ? w : width;
		}
		width = (width + 63) & ~(size_t)63;
		width = width > slot ? slot : width;
		bool ok = hipMemcpy2DAsync(t1, slot, hs, slot, width, cnt, hipMemcpyHostToDevice, qs) == hipSuccess;
		for (size_t k = 0; k < cnt && ok; ++k) {
			const Item& it = items[i0 + k];
			uint8_t* out = d_dst + it.s * (uint64_t)fi.sb;
			const uint8_t* in = t1 + k * slot + 16;
			hipError_t e = hipSuccess;
			if (it.code == 2) // plain zstd
				e = hipMemcpyAsync(out, in, it.dsize, hipMemcpyDeviceToDevice, qs);
			else if (it.code == 3) // zstd on the transposed superblock (stenos.cpp:700-710)
				e = stenos_k_launch_shuffle(in, out, (uint32_t)T, it.dsize, true, qs);
			else if (it.code == 4) { // transposed + byte delta (stenos.cpp:711-725)
				e = stenos_k_launch_delta(in, t2 + k * slot, it.dsize, true, qs);
				if (e == hipSuccess)
					e = stenos_k_launch_shuffle(t2 + k * slot, out, (uint32_t)T, it.dsize, true, qs);
			}
			ok = e == hipSuccess;
		}
		if (ok && !ids[set].empty()) {
			// (the two small tables come from the page-locked buffer: a copy from pageable memory is staged by the runtime and
			// waits for the stream, which would keep the host from inflating the next batch meanwhile)
			uint8_t* h_ids = stage.data() + tab_off + (size_t)set * (set_ids + set_idx);
			uint8_t* h_idx = h_ids + set_ids;
			memcpy(h_ids, ids[set].data(), ids[set].size() * 4);
			memcpy(h_idx, idx[set].data(), idx[set].size() * 8);
			ok = hipMemcpyAsync(d_ids, h_ids, ids[set].size() * 4, hipMemcpyHostToDevice, qs) == hipSuccess &&
			     hipMemcpyAsync(d_idx, h_idx, idx[set].size() * 8, hipMemcpyHostToDevice, qs) == hipSuccess &&
			     hipMemsetAsync(d_status, 0, 4, qs) == hipSuccess;
			DecodeArgs a;
			a.frame = t1;
			a.size = cnt * slot;
			a.sb_off = d_idx;
			a.sb_ids = d_ids;
			a.dst = d_dst;
			a.total_bytes = fi.total;
			a.nsb = ids[set].size();
			a.sb_bytes = (uint32_t)fi.sb;
			a.T = (uint32_t)T;
			a.status = d_status;
			ok = ok && wide_scratch(ctx, T, a.nsb, &a.wide_scratch, &a.wide_scratch_bytes) && stenos_k_launch_decode(a, qs) == hipSuccess &&
			     hipMemcpyAsync((void*)(h_status + set), d_status, 4, hipMemcpyDeviceToHost, qs) == hipSuccess;
		}
		ok = ok && hipEventRecord(ctx->batch_ev[(size_t)set], qs) == hipSuccess;
		if (!ok) {
			sync_all();
			return STENOS_ERROR_UNDEFINED;
		}
		pending[set] = true;
	}
	drain_frame();
	for (int set = 0; set < NSETS; ++set)
		if (size_t e = settle(set)) {
			sync_all();
			return e;
		}
	trace.mark("device finish", STAGE_DEVICE_FINISH);
	return 0;
}

size_t decompress_device(stenos_context_s* ctx, const void* d_src, size_t T, size_t size, void* d_dst, size_t dst_size, const uint64_t* d_index,
			 const uint64_t* h_index, const uint8_t* h_frame, hipStream_t stream, bool wait)
{
	if (!ctx->device_ready())
		return STENOS_ERROR_INVALID_INSTRUCTION_SET;
	uint8_t h[12] = { 0 };
	const size_t have = size < 12 ? size : 12;
	if (have && (hipMemcpyAsync(h, d_src, have, hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess))
		return STENOS_ERROR_UNDEFINED;
	FrameInfo fi;
	size_t e = parse_frame(h, have, T, dst_size, fi);
	if (is_err(e))
		return e;
	ctx->job_kind = 0;
	if (fi.total == 0)
		return 0;
	// (a caller's index may be the context's own, from stenos_hip_last_index / stenos_hip_frame_index: only touch it when none is given)
	if (!ctx->misc.ensure(4096) || (!d_index && !ctx->sboff.ensure((fi.nsb + 2) * 8)))
		return STENOS_ERROR_ALLOC;
	uint32_t* d_status = (uint32_t*)(ctx->misc.as<uint8_t>() + 8);
	if (hipMemsetAsync(d_status, 0, 4, stream) != hipSuccess)
		return STENOS_ERROR_UNDEFINED;
	if (!d_index) {
		d_index = ctx->sboff.as<uint64_t>();
		if (!ctx->walk.ensure(stenos_k_walk_scratch_bytes()))
			return STENOS_ERROR_ALLOC;
		if (stenos_k_launch_walk((const uint8_t*)d_src, size, fi.header, fi.nsb, (uint32_t)fi.sb, ctx->sboff.as<uint64_t>(), d_status,
					 ctx->test_serial_walk ? nullptr : ctx->walk.p, stream) != hipSuccess)
			return STENOS_ERROR_UNDEFINED;
	}
	DecodeArgs a;
	a.frame = (const uint8_t*)d_src;
	a.size = size;
	a.sb_off = d_index;
	a.dst = (uint8_t*)d_dst;
	a.total_bytes = fi.total;
	a.nsb = fi.nsb;
	a.sb_bytes = (uint32_t)fi.sb;
	a.T = (uint32_t)T;
	a.status = d_status;
	if (!wide_scratch(ctx, T, a.nsb, &a.wide_scratch, &a.wide_scratch_bytes))
		return STENOS_ERROR_ALLOC;
	ctx->mark(2, stream);
	if (stenos_k_launch_decode(a, stream) != hipSuccess)
		return STENOS_ERROR_UNDEFINED;
	ctx->mark(3, stream);
	if (hipMemcpyAsync((uint8_t*)ctx->h_total + 32, d_status, 4, hipMemcpyDeviceToHost, stream) != hipSuccess)
		return STENOS_ERROR_UNDEFINED;
	ctx->job_kind = 2;
	ctx->job_stream = stream;
	ctx->job_expected = (size_t)fi.total;
	if (!wait)
		return 0;
	size_t r = finish_job(ctx);
	if (!is_err(r) && ctx->job_host_codes) { // zstd-based superblocks present
		std::vector<uint64_t> idx;
		if (!h_index) {
			idx.resize(fi.nsb + 1);
			if (hipMemcpy(idx.data(), d_index, (fi.nsb + 1) * 8, hipMemcpyDeviceToHost) != hipSuccess)
				return STENOS_ERROR_UNDEFINED;
			h_index = idx.data();
		}
		e = finish_host_codes(ctx, (const uint8_t*)d_src, h_frame, size, T, h_index, fi, (uint8_t*)d_dst, stream);
		return is_err(e) ? e : (size_t)fi.total;
	}
	return r;
}

// ---- host-pointer calls on large inputs -------------------------------------------------------------
// The link is full duplex and the codec is ~30x faster than it, so the call is cut into chunks of whole superblocks:
// a helper thread uploads chunk k+1 on its own stream while the calling thread codes chunk k and downloads the
// result.  A chunk is a frame of its own on the device (superblocks are independent units, stenos.cpp:893-904), the
// caller's frame is the concatenation of the chunks' superblock streams behind one header.
constexpr size_t kHostChunkBytes = 32u << 20; // ~0.6 ms of link time, ~0.25 ms of fixed cost of a device call
constexpr size_t kHostChunkedFrom = 3 * kHostChunkBytes;

class Uploader {
	std::thread th;
	std::mutex m;
	std::condition_variable cv;
	size_t ready = 0;
	bool failed = false;
	std::atomic<bool> cancel{ false };

public:
	bool start(size_t chunks, std::function<bool(size_t)> upload) // false: no thread to be had (the caller takes the single pass)
	{
		int device = 0;
		(void)hipGetDevice(&device);
		try {
			th = std::thread([this, chunks, upload, device] {
			bool ok = hipSetDevice(device) == hipSuccess;
			for (size_t k = 0; k < chunks; ++k) {
				ok = ok && !cancel.load() && upload(k);
				std::lock_guard<std::mutex> l(m);
				failed = !ok;
				ready = ok ? k + 1 : chunks; // nobody waits for ever
				cv.notify_all();
				if (!ok)
					break;
			}
			});
		}
		catch (...) {
			return false;
		}
		return true;
	}
	bool wait_for(size_t k)
	{
		std::unique_lock<std::mutex> l(m);
		cv.wait(l, [&] { return ready > k; });
		return !failed;
	}
	~Uploader()
	{
		cancel = true;
		if (th.joinable())
			th.join();
	}
};

inline bool chunk_streams(stenos_context_s* ctx)
{
	for (hipStream_t* s : { &ctx->up_stream, &ctx->main_stream })
		if (!*s && hipStreamCreateWithFlags(s, hipStreamNonBlocking) != hipSuccess)
			return false;
	return true;
}

// *no_thread: the helper thread could not be started and nothing has been done (the caller takes the single pass)
size_t compress_chunked(stenos_context_s* ctx, const uint8_t* src, size_t T, size_t bytes, uint8_t* out, size_t dst_size, const FramePlan& f, bool* no_thread)
{
	*no_thread = false;
	const size_t chunk = kHostChunkBytes / f.sb * f.sb;
	const size_t chunks = (bytes + chunk - 1) / chunk;
	const size_t worst = f.header + (chunk / f.sb) * 4 + chunk; // a chunk stored as copies
	if (!chunk_streams(ctx) || !ctx->in.ensure(bytes + 64) || !ctx->out.ensure((dst_size < worst ? dst_size : worst) + 64))
		return STENOS_ERROR_ALLOC;
	uint8_t* d_in = ctx->in.as<uint8_t>();
	hipStream_t up_stream = ctx->up_stream, stream = ctx->main_stream;
	Uploader up;
	if (!up.start(chunks, [=](size_t k) {
		    const size_t begin = k * chunk, n = bytes - begin < chunk ? bytes - begin : chunk;
		    return hipMemcpyAsync(d_in + begin, src + begin, n, hipMemcpyHostToDevice, up_stream) == hipSuccess && hipStreamSynchronize(up_stream) == hipSuccess;
	    })) {
		*no_thread = true;
		return STENOS_ERROR_ALLOC;
	}
	out[0] = (uint8_t)f.shift;
	put_le(out + 1, bytes, 7);
	if (f.header == 12)
		put_le(out + 8, f.sb, 4);
	size_t off = f.header;
	for (size_t k = 0; k < chunks; ++k) {
		if (!up.wait_for(k))
			return STENOS_ERROR_UNDEFINED;
		const size_t begin = k * chunk, n = bytes - begin < chunk ? bytes - begin : chunk;
		// the chunk's frame sees the capacity the caller's buffer has left, so every superblock meets the room it would
		// meet in a single pass (stenos.cpp:893-904)
		const size_t room = dst_size - off + f.header;
		const size_t r = compress_device(ctx, d_in + begin, T, n, ctx->out.p, room, stream, true);
		if (is_err(r))
			return r;
		if (hipMemcpyAsync(out + off, ctx->out.as<uint8_t>() + f.header, r - f.header, hipMemcpyDeviceToHost, stream) != hipSuccess ||
		    hipStreamSynchronize(stream) != hipSuccess)
			return STENOS_ERROR_UNDEFINED;
		off += r - f.header;
	}
	return off;
}

// h_index: offsets of the superblock headers in the frame and its end (walked by the caller); only codes 1 and 6 inside.
// Superblocks [sA, sB) of the frame -> their bytes of `out`.
size_t decompress_chunked(stenos_context_s* ctx, const uint8_t* in, size_t T, const FrameInfo& fi, const std::vector<uint64_t>& h_index, uint8_t* out, uint64_t sA,
			  uint64_t sB, bool* no_thread = nullptr)
{
	if (no_thread)
		*no_thread = false;
	const uint64_t per = kHostChunkBytes / fi.sb ? kHostChunkBytes / fi.sb : 1; // superblocks per chunk
	const uint64_t count = sB - sA;
	const size_t chunks = (size_t)((count + per - 1) / per);
	const size_t size = (size_t)(h_index[sB] - h_index[sA]);
	const uint64_t oA = sA * fi.sb, oB = sB * fi.sb < fi.total ? sB * fi.sb : fi.total;
	const size_t H = fi.header; // 8, or 12 with a custom superblock size (repeated in every chunk's header)
	// chunk k on the device: [frame header of its own][its superblocks], 16 bytes further than in the frame per chunk
	// before it so that the headers do not overlap the neighbours; its index in sboff at entry (s0 - sA) + k
	if (!chunk_streams(ctx) || !ctx->in.ensure(size + 16 * (chunks + 1) + H + 64) || !ctx->out.ensure((size_t)(oB - oA) + 64) || !ctx->sboff.ensure((count + chunks + 2) * 8))
		return STENOS_ERROR_ALLOC;
	std::vector<uint64_t> rel(count + chunks);
	std::vector<uint8_t> hdr(12 * chunks);
	for (size_t k = 0; k < chunks; ++k) {
		const uint64_t s0 = sA + k * per, s1 = s0 + per < sB ? s0 + per : sB;
		for (uint64_t s = s0; s <= s1; ++s)
			rel[s - sA + k] = h_index[s] - h_index[s0] + H;
		const uint64_t o0 = s0 * fi.sb, o1 = s1 * fi.sb < fi.total ? s1 * fi.sb : fi.total;
		memcpy(&hdr[12 * k], in, H);
		put_le(&hdr[12 * k + 1], o1 - o0, 7);
	}
	uint8_t* d_in = ctx->in.as<uint8_t>();
	uint64_t* d_rel = ctx->sboff.as<uint64_t>();
	hipStream_t up_stream = ctx->up_stream, stream = ctx->main_stream;
	const uint64_t* idx = h_index.data();
	const uint64_t* relp = rel.data();
	const uint8_t* hdrp = hdr.data();
	auto chunk_frame = [=](size_t k) { return d_in + (idx[sA + k * per] - idx[sA]) + 16 * (k + 1); };
	Uploader up;
	if (!up.start(chunks, [=](size_t k) {
		    const uint64_t s0 = sA + k * per, s1 = s0 + per < sB ? s0 + per : sB;
		    uint8_t* d = chunk_frame(k);
		    return hipMemcpyAsync(d, hdrp + 12 * k, H, hipMemcpyHostToDevice, up_stream) == hipSuccess &&
			   hipMemcpyAsync(d + H, in + idx[s0], idx[s1] - idx[s0], hipMemcpyHostToDevice, up_stream) == hipSuccess &&
			   hipMemcpyAsync(d_rel + (s0 - sA) + k, relp + (s0 - sA) + k, (s1 - s0 + 1) * 8, hipMemcpyHostToDevice, up_stream) == hipSuccess &&
			   hipStreamSynchronize(up_stream) == hipSuccess;
	    })) {
		if (no_thread)
			*no_thread = true;
		return STENOS_ERROR_ALLOC;
	}
	for (size_t k = 0; k < chunks; ++k) {
		if (!up.wait_for(k))
			return STENOS_ERROR_UNDEFINED;
		const uint64_t s0 = sA + k * per, s1 = s0 + per < sB ? s0 + per : sB;
		const uint64_t o0 = s0 * fi.sb, o1 = s1 * fi.sb < fi.total ? s1 * fi.sb : fi.total;
		uint8_t* d_out = ctx->out.as<uint8_t>() + (o0 - oA);
		const size_t r = decompress_device(ctx, chunk_frame(k), T, (size_t)(H + h_index[s1] - h_index[s0]), d_out, (size_t)(o1 - o0), d_rel + (s0 - sA) + k, nullptr, nullptr,
						   stream, true);
		if (is_err(r))
			return r;
		if (r != o1 - o0 || ctx->job_host_codes)
			return STENOS_ERROR_INVALID_INPUT;
		if (hipMemcpyAsync(out + o0, d_out, (size_t)(o1 - o0), hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess)
			return STENOS_ERROR_UNDEFINED;
	}
	return (size_t)(oB - oA);
}

// ---- host-pointer calls on page-locked memory ---------------------------------------------------------
// Memory the caller has page-locked (hipHostMalloc, hipHostRegister, a pinned tensor) is visible to the device: the
// kernels then read the input and write the frame THROUGH the link, both directions at once, with no staging copy on
// either side -- the call is bound by the larger of the two transfers instead of their sum.  Returns the device alias of
// [p, p + n) or NULL (pageable memory, or a range that leaves its registration).
void* device_alias(const void* p, size_t n)
{
	hipPointerAttribute_t a;
	if (!p || hipPointerGetAttributes(&a, p) != hipSuccess) {
		(void)hipGetLastError(); // (pageable memory is "invalid value" to the runtime)
		return nullptr;
	}
	if (a.type != hipMemoryTypeHost || !a.devicePointer)
		return nullptr;
	void* base = nullptr;
	size_t size = 0;
	if (hipMemGetAddressRange((hipDeviceptr_t*)&base, &size, (hipDeviceptr_t)a.devicePointer) != hipSuccess) {
		(void)hipGetLastError();
		return nullptr;
	}
	const uintptr_t lo = (uintptr_t)a.devicePointer, end = (uintptr_t)base + size;
	return lo >= (uintptr_t)base && lo + n <= end ? a.devicePointer : nullptr;
}

// ---- host-pointer calls on several devices ------------------------------------------------------------
// The reference's dispatcher hands superblocks to the threads of stenos_set_threads (stenos.cpp:909-1010, 1151-1202).
// Here a host-pointer call is bound by the PCIe link of the device, not by the codec, so what pays is more DEVICES --
// more links -- per call.  That is opt-in (stenos_hip_set_devices, or STENOS_HIP_DEVICES >= 2 in the environment): the
// devices of a process are not the caller's to take just because it asked for CPU threads.  With it, a call uses
// min(threads, devices) of them: superblocks are independent in both directions, every device takes a
// contiguous range of them through a child context driven by a host thread of its own, and nothing is exchanged between
// devices (no collective: the caller's buffers are the meeting point).
//   compress:   every device uploads and encodes its range (a frame of its own, roomy destination); the sizes meet on
//               the host, a prefix sum gives every range its place and each device downloads straight to it.  Only
//               superblocks whose encoding cannot depend on the room that is left (safe_superblocks) are shared out; the
//               last one or two of a frame -- or all of them under a tight dst_size -- follow on the calling thread's
//               device with the exact room, as in the single-device path.
//   decompress: the host walks the superblock headers anyway; every device gets a range of them and writes its bytes.
// stenos_hip_test_lanes (tests on a one-GPU box) lets the lanes share the current device and makes one of them fail.
constexpr size_t kLanesFrom = (size_t)64 << 20; // below, one link moves the data before a second thread is up

// devices visible to the process (asked once: the answer does not change while the process lives)
int visible_devices()
{
	static const int n = [] {
		int k = 0;
		return hipGetDeviceCount(&k) == hipSuccess && k > 0 ? k : 1;
	}();
	return n;
}
// How many devices a host-pointer call of `bytes` may spread over.  Opt-in: stenos_hip_set_devices(ctx, n >= 2), or the
// environment variable STENOS_HIP_DEVICES >= 2 (read once) for callers that cannot be changed; without either a call stays
// on the calling thread's device whatever stenos_set_threads() says (the reference's CPU-thread knob, stenos.h:140).
int lane_count(stenos_context_s* ctx, size_t bytes)
{
	if (ctx->threads <= 1 || bytes < kLanesFrom)
		return 1;
	static const int env = [] {
		const char* e = getenv("STENOS_HIP_DEVICES");
		return e ? atoi(e) : 0;
	}();
	int n = ctx->hip_devices > 0 ? ctx->hip_devices : env;
	if (n < 2)
		return 1;
	if (!ctx->test_lanes_share_device && n > visible_devices())
		n = visible_devices();
	return n < ctx->threads ? n : ctx->threads;
}
// lane 0 is the context itself (the calling thread's device); lane i > 0 a child on device (current + i) % count
stenos_context_s* lane_context(stenos_context_s* ctx, int i, int* device)
{
	int cur = 0;
	(void)hipGetDevice(&cur);
	*device = ctx->test_lanes_share_device ? cur : (cur + i) % visible_devices();
	if (i == 0)
		return ctx;
	if ((int)ctx->lanes.size() < i)
		ctx->lanes.resize((size_t)i, nullptr);
	stenos_context_s*& l = ctx->lanes[(size_t)i - 1];
	if (!l) {
		void* m = malloc(sizeof(stenos_context_s));
		if (!m)
			return nullptr;
		l = new (m) stenos_context_s();
	}
	l->level = ctx->level;
	l->threads = 1;
	l->max_nanoseconds = 0;
	l->custom_shift = ctx->custom_shift;
	return l;
}
// Runs fn(i) for every lane on a thread of its own (lane 0 on the calling thread) with the lane's device current.
// result[i] must hold an error code on entry: a lane whose thread cannot be started, whose device cannot be made current
// or that is made to fail by the test hook leaves it there, so a lane that never ran is an error, not a result of 0.
bool run_lanes(stenos_context_s* ctx, int n, const std::vector<int>& device, const std::function<void(int)>& fn)
{
	std::vector<std::thread> th;
	bool ok = true;
	const int fail = ctx->test_fail_lane;
	for (int i = 1; i < n; ++i) {
		try {
			th.emplace_back([&, i] {
				if (i != fail && hipSetDevice(device[(size_t)i]) == hipSuccess)
					fn(i);
			});
		}
		catch (...) {
			ok = false;
			break;
		}
	}
	if (ok && fail != 0)
		fn(0);
	for (std::thread& t : th)
		t.join();
	return ok;
}

size_t compress_lanes(stenos_context_s* ctx, const uint8_t* src, size_t T, size_t bytes, uint8_t* out, size_t dst_size, const FramePlan& f, int n)
{
	// superblocks that are coded the same whatever room is left, all of them full: these are shared out
	uint64_t safe = codec::safe_superblocks(dst_size, f.header, f.bps, (uint32_t)T, f.sb, f.nsb);
	const uint64_t whole = f.nfull / f.bps;
	safe = safe < whole ? safe : whole;
	if (safe < (uint64_t)(2 * n))
		return STENOS_ERROR_INVALID_PARAMETER; // (not an error: the caller takes the single-device path)
	std::vector<stenos_context_s*> lane((size_t)n);
	std::vector<int> device((size_t)n);
	for (int i = 0; i < n; ++i)
		if (!(lane[(size_t)i] = lane_context(ctx, i, &device[(size_t)i])))
			return STENOS_ERROR_ALLOC;
	std::vector<size_t> got((size_t)n, (size_t)STENOS_ERROR_UNDEFINED); // (a lane that never runs is an error)
	auto range = [&](int i, uint64_t* a, uint64_t* b) {
		*a = safe * (uint64_t)i / (uint64_t)n;
		*b = safe * (uint64_t)(i + 1) / (uint64_t)n;
	};
	// upload + encode
	if (!run_lanes(ctx, n, device, [&](int i) {
		    stenos_context_s* c = lane[(size_t)i];
		    uint64_t a, b;
		    range(i, &a, &b);
		    const size_t nb = (size_t)(b - a) * f.sb, worst = f.header + (size_t)(b - a) * 4 + nb + 4096;
		    if (!c->device_ready() || !chunk_streams(c) || !c->in.ensure(nb + 64) || !c->out.ensure(worst + 64)) {
			    got[(size_t)i] = STENOS_ERROR_ALLOC;
			    return;
		    }
		    if (hipMemcpyAsync(c->in.p, src + a * f.sb, nb, hipMemcpyHostToDevice, c->main_stream) != hipSuccess) {
			    got[(size_t)i] = STENOS_ERROR_UNDEFINED;
			    return;
		    }
		    got[(size_t)i] = compress_device(c, c->in.p, T, nb, c->out.p, worst, c->main_stream, true);
	    }))
		return STENOS_ERROR_ALLOC;
	std::vector<size_t> off((size_t)n + 1);
	off[0] = f.header;
	for (int i = 0; i < n; ++i) {
		if (is_err(got[(size_t)i]))
			return got[(size_t)i];
		if (got[(size_t)i] < f.header)
			return STENOS_ERROR_UNDEFINED;
		off[(size_t)i + 1] = off[(size_t)i] + got[(size_t)i] - f.header;
	}
	if (off[(size_t)n] > dst_size)
		return STENOS_ERROR_DST_OVERFLOW; // (cannot happen for safe superblocks; never write past the buffer)
	// download, every range to its place
	std::vector<int> bad((size_t)n, 1); // (cleared by the lane once its bytes are in place)
	if (!run_lanes(ctx, n, device, [&](int i) {
		    stenos_context_s* c = lane[(size_t)i];
		    if (hipMemcpyAsync(out + off[(size_t)i], c->out.as<uint8_t>() + f.header, got[(size_t)i] - f.header, hipMemcpyDeviceToHost, c->main_stream) == hipSuccess &&
			hipStreamSynchronize(c->main_stream) == hipSuccess)
			    bad[(size_t)i] = 0;
	    }))
		return STENOS_ERROR_ALLOC;
	for (int b : bad)
		if (b)
			return STENOS_ERROR_UNDEFINED;
	out[0] = (uint8_t)f.shift;
	put_le(out + 1, bytes, 7);
	if (f.header == 12)
		put_le(out + 8, f.sb, 4);
	size_t end = off[(size_t)n];
	if (safe < f.nsb) { // the superblocks that look at the room: one more frame, with exactly the room the caller's buffer has left
		const size_t begin = (size_t)safe * f.sb, rest = bytes - begin;
		const size_t room = dst_size - end + f.header, worst = f.header + (size_t)(f.nsb - safe) * 4 + rest;
		if (!ctx->in.ensure(rest + 64) || !ctx->out.ensure((room < worst ? room : worst) + 64))
			return STENOS_ERROR_ALLOC;
		if (hipMemcpy(ctx->in.p, src + begin, rest, hipMemcpyHostToDevice) != hipSuccess)
			return STENOS_ERROR_UNDEFINED;
		const size_t r = compress_device(ctx, ctx->in.p, T, rest, ctx->out.p, room, nullptr, true);
		if (is_err(r))
			return r;
		if (hipMemcpy(out + end, ctx->out.as<uint8_t>() + f.header, r - f.header, hipMemcpyDeviceToHost) != hipSuccess)
			return STENOS_ERROR_UNDEFINED;
		end += r - f.header;
	}
	return end;
}

size_t decompress_lanes(stenos_context_s* ctx, const uint8_t* in, size_t T, const FrameInfo& fi, const std::vector<uint64_t>& h_index, uint8_t* out, int n)
{
	std::vector<stenos_context_s*> lane((size_t)n);
	std::vector<int> device((size_t)n);
	for (int i = 0; i < n; ++i)
		if (!(lane[(size_t)i] = lane_context(ctx, i, &device[(size_t)i])))
			return STENOS_ERROR_ALLOC;
	std::vector<size_t> got((size_t)n, (size_t)STENOS_ERROR_UNDEFINED); // (a lane that never runs is an error)
	if (!run_lanes(ctx, n, device, [&](int i) {
		    const uint64_t a = fi.nsb * (uint64_t)i / (uint64_t)n, b = fi.nsb * (uint64_t)(i + 1) / (uint64_t)n;
		    stenos_context_s* c = lane[(size_t)i];
		    got[(size_t)i] = !c->device_ready() ? (size_t)STENOS_ERROR_INVALID_INSTRUCTION_SET : (a < b ? decompress_chunked(c, in, T, fi, h_index, out, a, b) : 0);
	    }))
		return STENOS_ERROR_ALLOC;
	for (size_t r : got)
		if (is_err(r))
			return r;
	return (size_t)fi.total;
}

} // namespace

// =====================================================================================================
// exported C ABI
// =====================================================================================================
extern "C" {

stenos_context* stenos_make_context(void)
{
	void* m = malloc(sizeof(stenos_context_s));
	return m ? new (m) stenos_context_s() : nullptr;
}
void stenos_destroy_context(stenos_context* ctx)
{
	if (ctx) {
		ctx->~stenos_context_s();
		free(ctx);
	}
}
void stenos_reset_context(stenos_context* ctx) // stenos.cpp:245-252 (the custom block size is kept, as there)
{
	if (ctx) {
		ctx->level = 1;
		ctx->threads = 1;
		ctx->max_nanoseconds = 0;
	}
}
size_t stenos_set_level(stenos_context* ctx, int level)
{
	ctx->level = level > 9 ? 9 : (level < 0 ? 0 : level);
	return 0;
}
size_t stenos_set_threads(stenos_context* ctx, int threads)
{
	ctx->threads = threads < 1 ? 1 : threads;
	return 0;
}
size_t stenos_set_max_nanoseconds(stenos_context* ctx, uint64_t nanoseconds)
{
	ctx->max_nanoseconds = nanoseconds;
	return 0;
}
size_t stenos_set_block_size(stenos_context* ctx, size_t blocksize_shift)
{
	if (blocksize_shift >= 16 && blocksize_shift != STENOS_NO_BLOCK_SHIFT)
		return STENOS_ERROR_INVALID_PARAMETER;
	ctx->custom_shift = blocksize_shift;
	return 0;
}
size_t stenos_memory_footprint(stenos_context* ctx)
{
	// host bytes of the context; device buffers are reported by stenos_hip_workspace_bytes()
	(void)ctx;
	return sizeof(stenos_context_s);
}
int stenos_has_error(size_t r) { return r >= STENOS_LAST_ERROR_CODE; }
size_t stenos_bound(size_t bytes) { return stenos::compress_bound(bytes); }

// Time-limited compression (stenos_set_max_nanoseconds).  The reference keeps adjusting its level to the time that is left:
// per block inside the block codec, down to blocks stored as they are (block_compress.h:1024-1075, 1158-1176), per
// superblock for the zstd stages (zstd_wrapper.h:118-174, stenos.cpp:471-490), on superblocks sized after the thread
// count (stenos.cpp:126-149), and finishes with plain copies when nothing else fits.  Its output depends on the clock and
// is not reproducible.  Here the unit of adjustment is a slice of whole superblocks (default size, frame byte 0): before
// each slice the host clock and the rates measured so far decide whether the slice goes through zstd on top of the block
// codec (level 2, when the context's level allows it), through the block codec (level 1) or is stored as copies; a slice
// is only compressed when copying everything behind it would still fit the time that is left.  Every frame decodes with
// the ordinary decoder.
size_t compress_timed(stenos_context* ctx, const uint8_t* src, size_t T, size_t bytes, uint8_t* out, size_t dst_size)
{
	using clock = std::chrono::steady_clock;
	const auto start = clock::now();
	const double budget = (double)ctx->max_nanoseconds * 1e-9;
	if (T == 0 || T >= STENOS_MAX_BYTESOFTYPE)
		return STENOS_ERROR_INVALID_BYTESOFTYPE;
	const size_t sb = base_superblock(T * 256);
	if (dst_size < 8)
		return STENOS_ERROR_DST_OVERFLOW;
	out[0] = 0;
	put_le(out + 1, bytes, 7);
	// slices of 1/16 of the input, between 4 and 64 MiB: enough of them to adjust, each large enough for the device
	size_t slice = bytes / 16;
	slice = slice < ((size_t)4 << 20) ? ((size_t)4 << 20) : (slice > ((size_t)64 << 20) ? ((size_t)64 << 20) : slice);
	slice = (slice + sb - 1) / sb * sb;
	const int top = ctx->level > 2 ? 2 : ctx->level; // levels above 2 change the superblock size of a frame: not inside one frame
	double rate[3] = { 6e9, 12e9, 1e9 }; // bytes per second of a slice stored as copies / at level 1 / at level 2: first guesses, then measured
	const int saved_level = ctx->level;
	const uint64_t saved_ns = ctx->max_nanoseconds;
	const size_t saved_shift = ctx->custom_shift; // (the reference's time-limited frames choose their superblock size themselves, too)
	size_t off = 8, pos = 0, result = 0;
	while (pos < bytes) {
		const size_t n = bytes - pos < slice ? bytes - pos : slice;
		const double left = budget - std::chrono::duration<double>(clock::now() - start).count();
		const double rest = (double)(bytes - pos - n) / rate[0]; // what copying everything behind this slice takes
		// the first device call of a context also pays for the runtime's start, the code object and the buffers: a tight
		// budget on a cold context is better spent copying
		const double cold = ctx->warm ? 0.0 : 0.25;
		int level = 0;
		if (top >= 1 && left > 0 && (double)n / rate[1] + rest + cold <= left)
			level = 1;
		if (level == 1 && top >= 2 && zstd().ok && (double)n / rate[2] + rest <= left * 0.5)
			level = 2;
		const auto t0 = clock::now();
		size_t r;
		if (level == 0) {
			const size_t nsb = n / sb + (n % sb ? 1 : 0);
			if (dst_size - off < n + 4 * nsb) {
				result = STENOS_ERROR_DST_OVERFLOW;
				break;
			}
			for (size_t s = 0; s < nsb; ++s) { // compress_memcpy (stenos.cpp:363-374)
				const size_t m = n - s * sb < sb ? n - s * sb : sb;
				out[off] = 6;
				put_le(out + off + 1, m, 3);
				memcpy(out + off + 4, src + pos + s * sb, m);
				off += 4 + m;
			}
			r = 0;
		}
		else {
			// the slice as a frame of its own, written so that its 8-byte header falls on the 8 bytes in front of `off`
			uint8_t keep[8];
			memcpy(keep, out + off - 8, 8);
			ctx->level = level;
			ctx->max_nanoseconds = 0;
			ctx->custom_shift = STENOS_NO_BLOCK_SHIFT;
			r = stenos_compress_generic(ctx, src + pos, T, n, out + off - 8, dst_size - off + 8);
			ctx->level = saved_level;
			ctx->max_nanoseconds = saved_ns;
			ctx->custom_shift = saved_shift;
			memcpy(out + off - 8, keep, 8);
			if (is_err(r)) {
				result = r;
				break;
			}
			off += r - 8;
		}
		const double took = std::chrono::duration<double>(clock::now() - t0).count();
		if (took > 0)
			rate[level] = 0.5 * rate[level] + 0.5 * (double)n / took;
		pos += n;
	}
	return is_err(result) ? result : off;
}

size_t stenos_compress_generic(stenos_context* ctx, const void* src, size_t bytesoftype, size_t bytes, void* dst, size_t dst_size)
{
	FramePlan f;
	size_t e = plan_frame(ctx, bytesoftype, bytes, ctx->level, f);
	if (is_err(e))
		return e;
	if (ctx->max_nanoseconds && bytes && ctx->level)
		return compress_timed(ctx, (const uint8_t*)src, bytesoftype, bytes, (uint8_t*)dst, dst_size);
	e = check_supported(ctx, bytesoftype, ctx->level);
	if (is_err(e))
		return e;
	if (dst_size < f.header)
		return STENOS_ERROR_DST_OVERFLOW;
	uint8_t* out = (uint8_t*)dst;
	if (bytes == 0 || ctx->level == 0) {
		// header only, or plain copies (stenos.cpp:431-433, 363-374): no codec involved, done in place
		const size_t need = f.header + f.nsb * 4 + bytes;
		if (dst_size < need)
			return STENOS_ERROR_DST_OVERFLOW;
		out[0] = (uint8_t)f.shift;
		put_le(out + 1, bytes, 7);
		if (f.header == 12)
			put_le(out + 8, f.sb, 4);
		size_t off = f.header;
		for (uint64_t s = 0; s < f.nsb; ++s) {
			size_t n = (size_t)((bytes - s * f.sb) < f.sb ? (bytes - s * f.sb) : f.sb);
			out[off] = 6;
			put_le(out + off + 1, n, 3);
			memcpy(out + off + 4, (const uint8_t*)src + s * f.sb, n);
			off += 4 + n;
		}
		return off;
	}
	if (!ctx->device_ready())
		return STENOS_ERROR_INVALID_INSTRUCTION_SET;
	// the largest frame there can be: every superblock stored as a copy.  (stenos_bound() assumes superblocks of the
	// default size; with stenos_set_block_size() there can be many more headers.)  Nothing is written past dst_size.
	const size_t worst = f.header + f.nsb * 4 + bytes;
	if (!needs_strategy(bytesoftype, ctx->level)) {
		const int lanes = lane_count(ctx, bytes);
		ctx->last_devices = 1;
		if (lanes > 1) {
			const size_t r = compress_lanes(ctx, (const uint8_t*)src, bytesoftype, bytes, out, dst_size, f, lanes);
			if (r != STENOS_ERROR_INVALID_PARAMETER) { // (that one: too few shareable superblocks, e.g. a tight dst_size)
				ctx->last_devices = lanes;
				return r;
			}
		}
	}
	if (!needs_strategy(bytesoftype, ctx->level)) {
		// page-locked caller memory: no staging on that side (both sides: no copy at all)
		void* a_src = device_alias(src, bytes);
		void* a_dst = device_alias(dst, dst_size);
		// (one side only and a large call: the chunked path below overlaps its upload, coding and download, which a single
		// pass with a blocking copy on the other side would not)
		if ((a_src && a_dst) || ((a_src || a_dst) && bytes < kHostChunkedFrom)) {
			const size_t cap = dst_size < worst ? dst_size : worst;
			if ((!a_src && !ctx->in.ensure(bytes + 64)) || (!a_dst && !ctx->out.ensure(cap + 64)))
				return STENOS_ERROR_ALLOC;
			if (!a_src && hipMemcpy(ctx->in.p, src, bytes, hipMemcpyHostToDevice) != hipSuccess)
				return STENOS_ERROR_UNDEFINED;
			const size_t r = compress_device(ctx, a_src ? a_src : ctx->in.p, bytesoftype, bytes, a_dst ? a_dst : ctx->out.p, dst_size, nullptr, true);
			if (is_err(r))
				return r;
			if (!a_dst && hipMemcpy(dst, ctx->out.p, r, hipMemcpyDeviceToHost) != hipSuccess)
				return STENOS_ERROR_UNDEFINED;
			return r;
		}
	}
	if (bytes >= kHostChunkedFrom && !needs_strategy(bytesoftype, ctx->level)) {
		bool no_thread = false;
		const size_t r = compress_chunked(ctx, (const uint8_t*)src, bytesoftype, bytes, out, dst_size, f, &no_thread);
		if (!no_thread)
			return r;
	}
	if (!ctx->in.ensure(bytes + 64) || !ctx->out.ensure((dst_size < worst ? dst_size : worst) + 64))
		return STENOS_ERROR_ALLOC;
	if (hipMemcpy(ctx->in.p, src, bytes, hipMemcpyHostToDevice) != hipSuccess)
		return STENOS_ERROR_UNDEFINED;
	if (needs_strategy(bytesoftype, ctx->level))
		return compress_strategy(ctx, (const uint8_t*)src, ctx->in.as<uint8_t>(), bytesoftype, bytes, out, dst_size, ctx->level, f, nullptr);
	// the caller's dst_size is the logical capacity (a frame that does not fit is reported, nothing is written past it)
	size_t r = compress_device(ctx, ctx->in.p, bytesoftype, bytes, ctx->out.p, dst_size, nullptr, true);
	if (is_err(r))
		return r;
	if (hipMemcpy(dst, ctx->out.p, r, hipMemcpyDeviceToHost) != hipSuccess)
		return STENOS_ERROR_UNDEFINED;
	return r;
}

size_t stenos_decompress_generic(stenos_context* ctx, const void* src, size_t bytesoftype, size_t size, void* dst, size_t dst_size)
{
	const uint8_t* in = (const uint8_t*)src;
	FrameInfo fi;
	size_t e = parse_frame(in, size, bytesoftype, dst_size, fi);
	if (is_err(e))
		return e;
	if (fi.total == 0)
		return 0;
	// walk the superblock chain on the host (stenos.cpp:1124-1143): cheap here, serial on a GPU
	std::vector<uint64_t> index(fi.nsb + 1);
	uint64_t p = fi.header;
	bool gpu_codes = false, host_codes = false;
	for (uint64_t s = 0; s < fi.nsb; ++s) {
		if (p + 4 > size)
			return STENOS_ERROR_SRC_OVERFLOW;
		index[s] = p;
		const unsigned code = in[p];
		const size_t csize = (size_t)get_le(in + p + 1, 3);
		if (p + 4 + csize > size)
			return STENOS_ERROR_INVALID_INPUT;
		if (code == 1)
			gpu_codes = true;
		else if (code >= 2 && code <= 5)
			host_codes = true;
		else if (code != 6)
			return STENOS_ERROR_INVALID_INPUT;
		p += 4 + csize;
	}
	index[fi.nsb] = p;
	uint8_t* out = (uint8_t*)dst;
	bool device_codes = gpu_codes;
	for (uint64_t s = 0; s < fi.nsb && !device_codes; ++s)
		device_codes = in[index[s]] >= 3 && in[index[s]] <= 5;
	if (!device_codes) { // copies and zstd-only superblocks: nothing for the GPU to do
		std::atomic<size_t> err(0);
		parallel_for(fi.nsb, [&](uint64_t s) {
			const uint64_t begin = s * (uint64_t)fi.sb;
			const size_t dsize = (size_t)((fi.total - begin) < fi.sb ? (fi.total - begin) : fi.sb);
			const unsigned code = in[index[s]];
			const size_t csize = (size_t)get_le(in + index[s] + 1, 3);
			if (code == 6) {
				if (csize != dsize)
					err = STENOS_ERROR_INVALID_INPUT;
				else
					memcpy(out + begin, in + index[s] + 4, csize);
			}
			else if (code == 2) {
				if (!zstd().ok)
					err = STENOS_ERROR_ZSTD_INTERNAL;
				else if (zstd().is_error(zstd().decompress(out + begin, dsize, in + index[s] + 4, csize)))
					err = STENOS_ERROR_INVALID_INPUT;
			}
			else
				err = STENOS_ERROR_INVALID_INPUT;
		});
		if (err)
			return err;
		return (size_t)fi.total;
	}
	if (!ctx->device_ready())
		return STENOS_ERROR_INVALID_INSTRUCTION_SET;
	if (!host_codes) {
		const int lanes = lane_count(ctx, (size_t)fi.total);
		ctx->last_devices = 1;
		if (lanes > 1 && fi.nsb >= (uint64_t)(2 * lanes)) {
			ctx->last_devices = lanes;
			return decompress_lanes(ctx, in, bytesoftype, fi, index, out, lanes);
		}
	}
	if (!host_codes) {
		void* a_src = device_alias(src, size);
		void* a_dst = device_alias(dst, (size_t)fi.total);
		if ((a_src && a_dst) || ((a_src || a_dst) && fi.total < kHostChunkedFrom)) { // (one side only and large: the chunked path overlaps)
			if ((!a_src && !ctx->in.ensure(size + 64)) || (!a_dst && !ctx->out.ensure((size_t)fi.total + 64)) || !ctx->sboff.ensure((fi.nsb + 2) * 8))
				return STENOS_ERROR_ALLOC;
			if ((!a_src && hipMemcpy(ctx->in.p, src, size, hipMemcpyHostToDevice) != hipSuccess) ||
			    hipMemcpy(ctx->sboff.p, index.data(), (fi.nsb + 1) * 8, hipMemcpyHostToDevice) != hipSuccess)
				return STENOS_ERROR_UNDEFINED;
			const size_t r = decompress_device(ctx, a_src ? a_src : ctx->in.p, bytesoftype, size, a_dst ? a_dst : ctx->out.p, (size_t)fi.total, ctx->sboff.as<uint64_t>(),
							   index.data(), in, nullptr, true);
			if (is_err(r))
				return r;
			if (!a_dst && hipMemcpy(dst, ctx->out.p, (size_t)fi.total, hipMemcpyDeviceToHost) != hipSuccess)
				return STENOS_ERROR_UNDEFINED;
			return (size_t)fi.total;
		}
	}
	if (fi.total >= kHostChunkedFrom && !host_codes) {
		bool no_thread = false;
		const size_t r = decompress_chunked(ctx, in, bytesoftype, fi, index, out, 0, fi.nsb, &no_thread);
		if (!no_thread)
			return r;
	}
	if (!ctx->in.ensure(size + 64) || !ctx->out.ensure((size_t)fi.total + 64) || !ctx->sboff.ensure((fi.nsb + 2) * 8))
		return STENOS_ERROR_ALLOC;
	if (hipMemcpy(ctx->in.p, src, size, hipMemcpyHostToDevice) != hipSuccess ||
	    hipMemcpy(ctx->sboff.p, index.data(), (fi.nsb + 1) * 8, hipMemcpyHostToDevice) != hipSuccess)
		return STENOS_ERROR_UNDEFINED;
	(void)host_codes;
	size_t r = decompress_device(ctx, ctx->in.p, bytesoftype, size, ctx->out.p, (size_t)fi.total, ctx->sboff.as<uint64_t>(), index.data(), in, nullptr, true);
	if (is_err(r))
		return r;
	if (hipMemcpy(dst, ctx->out.p, (size_t)fi.total, hipMemcpyDeviceToHost) != hipSuccess)
		return STENOS_ERROR_UNDEFINED;
	return (size_t)fi.total;
}

// The reference builds a temporary context per call (stenos.cpp:1210-1226).  Here a context owns device buffers of about
// twice the input, so the one-shot calls of a thread share one context that lives as long as the thread: its buffers
// are kept between calls (and freed by the thread's exit).
static stenos_context_s& one_shot_context()
{
	static thread_local stenos_context_s ctx;
	ctx.threads = 1; // every parameter as a fresh context has it (the level is set by the caller)
	ctx.max_nanoseconds = 0;
	ctx.custom_shift = STENOS_NO_BLOCK_SHIFT;
	return ctx;
}
size_t stenos_compress(const void* src, size_t bytesoftype, size_t bytes, void* dst, size_t dst_size, int level)
{
	stenos_context_s& ctx = one_shot_context();
	ctx.level = level > 9 ? 9 : (level < 0 ? 0 : level);
	return stenos_compress_generic(&ctx, src, bytesoftype, bytes, dst, dst_size);
}
size_t stenos_decompress(const void* src, size_t bytesoftype, size_t bytes, void* dst, size_t dst_size)
{
	return stenos_decompress_generic(&one_shot_context(), src, bytesoftype, bytes, dst, dst_size);
}

size_t stenos_get_info(const void* src, size_t bytesoftype, size_t bytes, stenos_info* info) // stenos.cpp:1019-1050
{
	const uint8_t* in = (const uint8_t*)src;
	if (bytes < 8)
		return STENOS_ERROR_SRC_OVERFLOW;
	const unsigned shift = in[0];
	if (shift > 4 && shift != 255)
		return STENOS_ERROR_INVALID_INPUT;
	info->decompressed_size = (size_t)get_le(in + 1, 7);
	if (shift == 255) {
		if (bytes < 12)
			return STENOS_ERROR_SRC_OVERFLOW;
		info->superblock_size = (size_t)get_le(in + 8, 4);
		return 12;
	}
	info->superblock_size = base_superblock(bytesoftype * 256) << shift;
	return 8;
}

// ---- timer (stenos.cpp:1232-1257, timer.hpp:104-133) ----
struct stenos_timer_s {
	struct timespec t0;
};
stenos_timer* stenos_make_timer(void)
{
	stenos_timer* t = (stenos_timer*)malloc(sizeof(stenos_timer));
	if (t)
		clock_gettime(CLOCK_MONOTONIC, &t->t0);
	return t;
}
void stenos_destroy_timer(stenos_timer* timer) { free(timer); }
void stenos_tick(stenos_timer* timer) { clock_gettime(CLOCK_MONOTONIC, &timer->t0); }
uint64_t stenos_tock(stenos_timer* timer)
{
	struct timespec t1;
	clock_gettime(CLOCK_MONOTONIC, &t1);
	return (uint64_t)(t1.tv_sec - timer->t0.tv_sec) * 1000000000ull + (uint64_t)t1.tv_nsec - (uint64_t)timer->t0.tv_nsec;
}

// ---- private single-superblock API used by stenos::cvector (stenos.cpp:768-842) ----
size_t stenos_private_compress_block(stenos_context* ctx, const void* src, size_t bytesoftype, size_t super_block_size, size_t bytes, void* dst,
				     size_t dst_size)
{
	if (dst_size < 4) // stenos.cpp:427-429
		return STENOS_ERROR_DST_OVERFLOW;
	if (bytesoftype == 0 || bytesoftype >= STENOS_MAX_BYTESOFTYPE)
		return STENOS_ERROR_INVALID_BYTESOFTYPE;
	uint8_t* out = (uint8_t*)dst;
	if (bytes == 0 || ctx->level == 0) { // MEMCPY (stenos.cpp:431-433)
		if (dst_size < bytes + 4)
			return STENOS_ERROR_DST_OVERFLOW;
		out[0] = 6;
		put_le(out + 1, bytes, 3);
		memcpy(out + 4, src, bytes);
		return bytes + 4;
	}
	size_t e = check_supported(ctx, bytesoftype, ctx->level);
	if (is_err(e))
		return e;
	if (!ctx->device_ready())
		return STENOS_ERROR_INVALID_INSTRUCTION_SET;
	if (bytes > super_block_size || super_block_size >= STENOS_MAX_BLOCK_BYTES)
		return STENOS_ERROR_INVALID_PARAMETER;
	FramePlan f;
	f.sb = super_block_size;
	f.nsb = 1;
	f.nfull = bytes / (bytesoftype * 256);
	f.tail = (uint32_t)(bytes % (bytesoftype * 256));
	f.bps = (uint32_t)(super_block_size / (bytesoftype * 256));
	if (f.bps == 0)
		return STENOS_ERROR_INVALID_PARAMETER;
	if (!ctx->in.ensure(bytes + 64) || !ctx->out.ensure(bytes + 64))
		return STENOS_ERROR_ALLOC;
	if (hipMemcpy(ctx->in.p, src, bytes, hipMemcpyHostToDevice) != hipSuccess)
		return STENOS_ERROR_UNDEFINED;
	if (needs_strategy(bytesoftype, ctx->level)) {
		// levels >= 2 and bytesoftype 1 (stenos::cvector<char>, cvector at higher levels): the strategy layer on this one
		// superblock, as a frame of one superblock whose 8-byte header is dropped; the superblock sees the caller's capacity
		f.header = 8;
		f.shift = 0;
		HostBuf& tmp = ctx->h_out;
		if (!tmp.ensure(dst_size + 8 + 64))
			return STENOS_ERROR_ALLOC;
		const size_t r = compress_strategy(ctx, (const uint8_t*)src, ctx->in.as<uint8_t>(), bytesoftype, bytes, tmp.data(), dst_size + 8, ctx->level, f, nullptr);
		if (is_err(r))
			return r;
		memcpy(dst, tmp.data() + 8, r - 8);
		return r - 8;
	}
	e = enqueue_compress(ctx, ctx->in.as<uint8_t>(), bytesoftype, bytes, ctx->out.as<uint8_t>(), dst_size, ctx->level, f, false, nullptr);
	if (is_err(e))
		return e;
	ctx->job_kind = 1;
	ctx->job_stream = nullptr;
	ctx->job_dst_size = dst_size;
	size_t r = finish_job(ctx);
	if (is_err(r))
		return r;
	if (r > dst_size)
		return STENOS_ERROR_DST_OVERFLOW;
	if (hipMemcpy(dst, ctx->out.p, r, hipMemcpyDeviceToHost) != hipSuccess)
		return STENOS_ERROR_UNDEFINED;
	return r;
}

size_t stenos_private_decompress_block(stenos_context* ctx, const void* src, size_t bytesoftype, size_t super_block_size, size_t bytes, void* dst,
				       size_t dst_size)
{
	const uint8_t* in = (const uint8_t*)src;
	if (bytes < 4) // stenos.cpp:792-793
		return STENOS_ERROR_SRC_OVERFLOW;
	if (bytesoftype == 0 || bytesoftype >= STENOS_MAX_BYTESOFTYPE)
		return STENOS_ERROR_INVALID_BYTESOFTYPE;
	const unsigned code = in[0];
	const size_t csize = (size_t)get_le(in + 1, 3);
	if (4 + csize > bytes)
		return STENOS_ERROR_INVALID_INPUT;
	if (code == 6) {
		if (csize != dst_size)
			return STENOS_ERROR_INVALID_INPUT;
		memcpy(dst, in + 4, csize);
		return dst_size;
	}
	if (code == 2) {
		if (!zstd().ok)
			return STENOS_ERROR_ZSTD_INTERNAL;
		size_t r = zstd().decompress(dst, dst_size, in + 4, csize);
		return zstd().is_error(r) ? STENOS_ERROR_INVALID_INPUT : dst_size;
	}
	if (code >= 3 && code <= 5) {
		// transposed / transposed + delta / block codec output, each under zstd (decompress_generic_superblock, stenos.cpp:700-740):
		// the same machinery as for frames, given this superblock as a frame of one superblock of a custom size
		if (dst_size == 0 || dst_size > super_block_size || super_block_size < bytesoftype * 256 || super_block_size >= STENOS_MAX_BLOCK_BYTES)
			return STENOS_ERROR_INVALID_INPUT;
		std::vector<uint8_t> frame(12 + 4 + csize);
		frame[0] = 255;
		put_le(frame.data() + 1, dst_size, 7);
		put_le(frame.data() + 8, super_block_size, 4);
		memcpy(frame.data() + 12, in, 4 + csize);
		return stenos_decompress_generic(ctx, frame.data(), bytesoftype, frame.size(), dst, dst_size);
	}
	if (code != 1)
		return STENOS_ERROR_INVALID_INPUT;
	if (dst_size == 0)
		return 0;
	if (!ctx->device_ready())
		return STENOS_ERROR_INVALID_INSTRUCTION_SET;
	if (!ctx->in.ensure(4 + csize + 64) || !ctx->out.ensure(dst_size + 64) || !ctx->sboff.ensure(32) || !ctx->misc.ensure(4096))
		return STENOS_ERROR_ALLOC;
	const uint64_t index[2] = { 0, 4 + csize };
	if (hipMemcpy(ctx->in.p, src, 4 + csize, hipMemcpyHostToDevice) != hipSuccess ||
	    hipMemcpy(ctx->sboff.p, index, sizeof(index), hipMemcpyHostToDevice) != hipSuccess)
		return STENOS_ERROR_UNDEFINED;
	uint32_t* d_status = (uint32_t*)(ctx->misc.as<uint8_t>() + 8);
	if (hipMemsetAsync(d_status, 0, 4, nullptr) != hipSuccess)
		return STENOS_ERROR_UNDEFINED;
	DecodeArgs a;
	a.frame = ctx->in.as<uint8_t>();
	a.size = 4 + csize;
	a.sb_off = ctx->sboff.as<uint64_t>();
	a.dst = ctx->out.as<uint8_t>();
	a.total_bytes = dst_size;
	a.nsb = 1;
	a.sb_bytes = (uint32_t)dst_size;
	a.T = (uint32_t)bytesoftype;
	a.status = d_status;
	if (!wide_scratch(ctx, bytesoftype, a.nsb, &a.wide_scratch, &a.wide_scratch_bytes))
		return STENOS_ERROR_ALLOC;
	if (stenos_k_launch_decode(a, nullptr) != hipSuccess)
		return STENOS_ERROR_UNDEFINED;
	uint32_t status = 0;
	if (hipMemcpy(&status, d_status, 4, hipMemcpyDeviceToHost) != hipSuccess)
		return STENOS_ERROR_UNDEFINED;
	if (status)
		return STENOS_ERROR_INVALID_INPUT;
	if (hipMemcpy(dst, ctx->out.p, dst_size, hipMemcpyDeviceToHost) != hipSuccess)
		return STENOS_ERROR_UNDEFINED;
	return dst_size;
}

size_t stenos_private_block_size(const void* src, size_t src_size)
{
	if (src_size < 4)
		return STENOS_ERROR_SRC_OVERFLOW;
	return (size_t)get_le((const uint8_t*)src + 1, 3) + 4;
}
size_t stenos_private_block_csize(const void* src)
{
	if (!src)
		return 0;
	return (size_t)get_le((const uint8_t*)src + 1, 3) + 4;
}
size_t stenos_private_create_compression_header(size_t decompressed_size, size_t super_block_size, void* dst, size_t dst_size)
{
	if (dst_size < 12)
		return STENOS_ERROR_DST_OVERFLOW;
	uint8_t* out = (uint8_t*)dst;
	out[0] = 255;
	put_le(out + 1, decompressed_size, 7);
	put_le(out + 8, super_block_size, 4);
	return 12;
}

// =====================================================================================================
// device-pointer entry points (include/stenos_hip.h)
// =====================================================================================================

int stenos_hip_last_devices(stenos_context* ctx) { return ctx ? ctx->last_devices : 0; }
void stenos_hip_set_devices(stenos_context* ctx, int devices)
{
	if (ctx)
		ctx->hip_devices = devices > 0 ? devices : 0;
}
int stenos_hip_stage_ms(stenos_context* ctx, double* out, int n, int reset)
{
	if (!ctx)
		return 0;
	const int k = n < (int)STAGE_COUNT ? n : (int)STAGE_COUNT;
	for (int i = 0; i < k && out; ++i)
		out[i] = ctx->stage_ms[i];
	if (reset)
		for (double& v : ctx->stage_ms)
			v = 0;
	return (int)STAGE_COUNT;
}
int stenos_hip_fused_fallbacks(stenos_context* ctx) { return ctx ? ctx->fused_fallbacks : 0; }
#ifdef STENOS_TEST_HOOKS // (the test suite's own build only: tests/hooks/Makefile)
int stenos_hip_test_walk(stenos_context* ctx, int serial)
{
	if (!ctx)
		return -1;
	int fell_back = -1; // the flag word the last parallel walk left in its scratch (walk_kernels.hip): 1 = the serial walk ran after all
	uint32_t flags = 0;
	if (!ctx->test_serial_walk && ctx->walk.p && ctx->device_ready() && hipDeviceSynchronize() == hipSuccess &&
	    hipMemcpy(&flags, ctx->walk.p, 4, hipMemcpyDeviceToHost) == hipSuccess)
		fell_back = flags != 0;
	ctx->test_serial_walk = serial != 0;
	return fell_back;
}
void stenos_hip_test_lanes(stenos_context* ctx, int share_current_device, int fail_lane)
{
	if (!ctx)
		return;
	ctx->test_lanes_share_device = share_current_device != 0;
	ctx->test_fail_lane = fail_lane;
}
void stenos_hip_test_fused_timeouts(stenos_context* ctx, int n)
{
	if (ctx && n > 0)
		ctx->inject_chain_timeout = n;
}
#endif
int stenos_hip_device_count(void)
{
	int n = 0;
	return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

size_t stenos_hip_workspace_bytes(size_t bytesoftype, size_t bytes)
{
	if (bytesoftype == 0 || bytesoftype > kMaxT)
		return 0;
	const size_t bs = bytesoftype * 256;
	const size_t sb = base_superblock(bs);
	const size_t nblocks = bytes / bs + 2;
	const size_t nsb = bytes / sb + 2;
	const uint32_t T = (uint32_t)bytesoftype;
	// the arena: staging buffers of the fused encoder (two per resident workgroup, whatever the input size) plus the slots of
	// the last superblocks or, where that kernel does not apply, one padded slot per block; then 12 bytes of tables per block
	// and 37 per superblock (default superblock size).  (A destination below stenos_bound() sends every block through slots.)
	const size_t arena = stenos_k_fused_supported(T) ? stenos_k_fused_stage_bytes(T, (uint32_t)(sb / bs), nsb) + 2 * (sb / bs + 1) * stenos_k_slot_stride(T)
							 : nblocks * (size_t)stenos_k_slot_stride(T);
	size_t wide = 0; // bytesoftype above 64: the scratch of kernels_wide.hip (wide_scratch())
	if (T > STENOS_K_LDS_MAX_T) {
		const size_t stride = stenos_kw_scratch_stride(T);
		size_t groups = ((size_t)1 << 30) / stride;
		groups = groups > nblocks ? nblocks : groups;
		wide = (groups > 2048 ? 2048 : (groups < 1 ? 1 : groups)) * stride;
	}
	return arena + wide + nblocks * 16 + nsb * 37 + 4096;
}

size_t stenos_hip_compress(stenos_context* ctx, const void* d_src, size_t bytesoftype, size_t bytes, void* d_dst, size_t dst_size, void* stream)
{
	return compress_device(ctx, d_src, bytesoftype, bytes, d_dst, dst_size, (hipStream_t)stream, true);
}
size_t stenos_hip_compress_async(stenos_context* ctx, const void* d_src, size_t bytesoftype, size_t bytes, void* d_dst, size_t dst_size, void* stream)
{
	return compress_device(ctx, d_src, bytesoftype, bytes, d_dst, dst_size, (hipStream_t)stream, false);
}

size_t stenos_hip_finish(stenos_context* ctx)
{
	size_t r = finish_job(ctx);
	// frames with zstd-based superblocks need the synchronous call, which finishes them on the host
	return (!is_err(r) && ctx->job_host_codes) ? STENOS_ERROR_ZSTD_INTERNAL : r;
}

} // extern "C"

namespace {
size_t finish_job(stenos_context_s* ctx)
{
	ctx->job_host_codes = false;
	if (!ctx->job_kind)
		return STENOS_ERROR_INVALID_PARAMETER;
	if (hipStreamSynchronize(ctx->job_stream) != hipSuccess)
		return STENOS_ERROR_UNDEFINED;
	ctx->warm = true;
	const int kind = ctx->job_kind;
	ctx->job_kind = 0;
	if (kind == 1) {
		const uint64_t total = ctx->h_total[0];
		uint32_t estatus = *(const uint32_t*)((const uint8_t*)ctx->h_total + 12);
		if (ctx->inject_chain_timeout > 0 && ctx->job_src && !ctx->no_fused) {
			--ctx->inject_chain_timeout;
			estatus |= codec::ENCODE_STATUS_CHAIN_TIMEOUT;
		}
		if (estatus & codec::ENCODE_STATUS_CHAIN_TIMEOUT) {
			// The fused kernel gave up waiting for a frame offset (its waits are bounded so that a scheduling accident cannot
			// hang the device; never seen in practice).  The frame is then produced once more by the kernels that need no
			// such wait (encode_blocks / plan / scan / pack); only when that fails too is the call an error.
			if (ctx->no_fused || !ctx->job_src)
				return STENOS_ERROR_UNDEFINED;
			ctx->no_fused = true;
			++ctx->fused_fallbacks;
			const void* src = ctx->job_src;
			ctx->job_src = nullptr;
			const size_t r = compress_device(ctx, src, ctx->job_T, ctx->job_bytes, ctx->job_dst, ctx->job_dst_size, ctx->job_stream, true);
			ctx->no_fused = false;
			return r;
		}
		ctx->job_src = nullptr;
		return (estatus || total > ctx->job_dst_size) ? STENOS_ERROR_DST_OVERFLOW : (size_t)total;
	}
	const uint32_t status = *(const uint32_t*)((const uint8_t*)ctx->h_total + 32);
	if (status & DECODE_STATUS_TRUNCATED)
		return STENOS_ERROR_SRC_OVERFLOW;
	if (status & DECODE_STATUS_INVALID)
		return STENOS_ERROR_INVALID_INPUT;
	ctx->job_host_codes = (status & DECODE_STATUS_HOST_CODES) != 0;
	return ctx->job_expected;
}
} // namespace

extern "C" {

static size_t byte_kernel(hipError_t e) { return e == hipSuccess ? 0 : STENOS_ERROR_UNDEFINED; }
size_t stenos_hip_shuffle(const void* d_src, size_t bytesoftype, size_t bytes, void* d_dst, void* stream)
{
	if (bytesoftype == 0 || bytesoftype >= STENOS_MAX_BYTESOFTYPE)
		return STENOS_ERROR_INVALID_BYTESOFTYPE;
	return byte_kernel(stenos_k_launch_shuffle((const uint8_t*)d_src, (uint8_t*)d_dst, (uint32_t)bytesoftype, bytes, false, (hipStream_t)stream));
}
size_t stenos_hip_unshuffle(const void* d_src, size_t bytesoftype, size_t bytes, void* d_dst, void* stream)
{
	if (bytesoftype == 0 || bytesoftype >= STENOS_MAX_BYTESOFTYPE)
		return STENOS_ERROR_INVALID_BYTESOFTYPE;
	return byte_kernel(stenos_k_launch_shuffle((const uint8_t*)d_src, (uint8_t*)d_dst, (uint32_t)bytesoftype, bytes, true, (hipStream_t)stream));
}
size_t stenos_hip_delta(const void* d_src, void* d_dst, size_t bytes, void* stream)
{
	return byte_kernel(stenos_k_launch_delta((const uint8_t*)d_src, (uint8_t*)d_dst, bytes, false, (hipStream_t)stream));
}
size_t stenos_hip_delta_inv(const void* d_src, void* d_dst, size_t bytes, void* stream)
{
	return byte_kernel(stenos_k_launch_delta((const uint8_t*)d_src, (uint8_t*)d_dst, bytes, true, (hipStream_t)stream));
}

void stenos_hip_set_profiling(stenos_context* ctx, int enabled) { ctx->profiling = enabled != 0; }
double stenos_hip_kernel_ms(stenos_context* ctx, int which)
{
	if (which < 0 || which > 1 || !ctx->ev_valid[which])
		return -1.0;
	float ms = 0.f;
	if (hipEventSynchronize(ctx->ev[2 * which + 1]) != hipSuccess || hipEventElapsedTime(&ms, ctx->ev[2 * which], ctx->ev[2 * which + 1]) != hipSuccess)
		return -1.0;
	return (double)ms;
}

const uint64_t* stenos_hip_last_index(stenos_context* ctx, size_t* nsb)
{
	if (nsb)
		*nsb = ctx->last_nsb;
	return ctx->sboff.as<uint64_t>();
}

const uint64_t* stenos_hip_frame_index(stenos_context* ctx, const void* d_src, size_t bytesoftype, size_t bytes, size_t* nsb, void* stream_)
{
	if (nsb)
		*nsb = 0;
	hipStream_t stream = (hipStream_t)stream_;
	if (!ctx || !d_src || !ctx->device_ready())
		return nullptr;
	uint8_t h[12] = { 0 };
	const size_t have = bytes < 12 ? bytes : 12;
	if (have && (hipMemcpyAsync(h, d_src, have, hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess))
		return nullptr;
	FrameInfo fi;
	if (is_err(parse_frame(h, have, bytesoftype, ~(size_t)0, fi)) || fi.total == 0)
		return nullptr;
	if (!ctx->misc.ensure(4096) || !ctx->sboff.ensure((fi.nsb + 2) * 8) || !ctx->walk.ensure(stenos_k_walk_scratch_bytes()))
		return nullptr;
	uint32_t* d_status = (uint32_t*)(ctx->misc.as<uint8_t>() + 8);
	uint32_t status = 0;
	if (hipMemsetAsync(d_status, 0, 4, stream) != hipSuccess ||
	    stenos_k_launch_walk((const uint8_t*)d_src, bytes, fi.header, fi.nsb, (uint32_t)fi.sb, ctx->sboff.as<uint64_t>(), d_status,
				 ctx->test_serial_walk ? nullptr : ctx->walk.p, stream) != hipSuccess ||
	    hipMemcpyAsync(&status, d_status, 4, hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess || status)
		return nullptr; // a header or payload runs past the end of the frame
	if (nsb)
		*nsb = (size_t)fi.nsb;
	return ctx->sboff.as<uint64_t>();
}

size_t stenos_hip_decompress(stenos_context* ctx, const void* d_src, size_t bytesoftype, size_t bytes, void* d_dst, size_t dst_size,
			     const uint64_t* d_index, void* stream)
{
	return decompress_device(ctx, d_src, bytesoftype, bytes, d_dst, dst_size, d_index, nullptr, nullptr, (hipStream_t)stream, true);
}
size_t stenos_hip_decompress_async(stenos_context* ctx, const void* d_src, size_t bytesoftype, size_t bytes, void* d_dst, size_t dst_size,
				   const uint64_t* d_index, void* stream)
{
	return decompress_device(ctx, d_src, bytesoftype, bytes, d_dst, dst_size, d_index, nullptr, nullptr, (hipStream_t)stream, false);
}

} // extern "C"
