// ubench_copy.hip -- what a device-to-device copy reaches on this pool, to settle the "copy ceiling" the decoder is held
// against: MI355X_MICROARCH.md quotes 6.29 TB/s (read + written) for a float4 copy; the streaming kernel of ubench_bw.hip and the
// runtime's own copy reach 4.9-5.1 here.  Forms tried, 16 bytes per lane and access, read + written bytes per second:
//   grid-stride   persistent grid (CUs x waves), accesses of a wave 64 x 16 B apart per step, UNROLL loads in flight (ubench_bw.hip)
//   one-shot      one thread per 16 bytes x UNROLL, the grid as large as the data (the naive float4 copy)
//   chunked       workgroups of 1024 threads, each streaming a contiguous chunk of CHUNK MiB front to back
//   split         the same chunks, but a workgroup either only reads (and folds) or only writes: read and write streams that do
//                 not wait for each other (not a copy: the ceiling of concurrent read and write traffic)
// each with plain and with non-temporal accesses, at 2 and 8 GiB; and hipMemcpyAsync device to device.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_copy.hip -o build/ubench_copy && build/ubench_copy
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <bool NT> __device__ inline u32x4 ld(const u32x4* p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT> __device__ inline void st(u32x4* p, u32x4 v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

template <bool NT, int UNROLL>
__global__ __launch_bounds__(256) void copy_gridstride(const u32x4* __restrict__ src, u32x4* __restrict__ dst, uint64_t n)
{
	const uint64_t tid = blockIdx.x * 256ull + threadIdx.x, step = gridDim.x * 256ull;
	for (uint64_t i = tid; i < n; i += step * UNROLL) {
		u32x4 v[UNROLL];
#pragma unroll
		for (int k = 0; k < UNROLL; ++k)
			if (i + k * step < n) v[k] = ld<NT>(src + i + k * step);
#pragma unroll
		for (int k = 0; k < UNROLL; ++k)
			if (i + k * step < n) st<NT>(dst + i + k * step, v[k]);
	}
}
template <bool NT, int UNROLL>
__global__ __launch_bounds__(256) void copy_oneshot(const u32x4* __restrict__ src, u32x4* __restrict__ dst, uint64_t n)
{
	const uint64_t base = (uint64_t)blockIdx.x * 256 * UNROLL + threadIdx.x;
	u32x4 v[UNROLL];
#pragma unroll
	for (int k = 0; k < UNROLL; ++k)
		if (base + k * 256 < n) v[k] = ld<NT>(src + base + k * 256);
#pragma unroll
	for (int k = 0; k < UNROLL; ++k)
		if (base + k * 256 < n) st<NT>(dst + base + k * 256, v[k]);
}
// mode 0: copy; 1: even workgroups read their chunk (and fold it), odd ones write theirs
template <bool NT, int UNROLL>
__global__ __launch_bounds__(1024) void copy_chunked(const u32x4* __restrict__ src, u32x4* __restrict__ dst, uint64_t n, uint64_t chunk, int mode, uint32_t* sink)
{
	u32x4 acc = { 0, 0, 0, 0 };
	for (uint64_t c = blockIdx.x; c * chunk < n; c += gridDim.x) {
		const uint64_t lo = c * chunk, hi = lo + chunk < n ? lo + chunk : n;
		for (uint64_t i = lo + threadIdx.x; i < hi; i += 1024ull * UNROLL) {
			u32x4 v[UNROLL];
#pragma unroll
			for (int k = 0; k < UNROLL; ++k) {
				v[k] = acc;
				if (i + k * 1024ull < hi && !(mode == 1 && (c & 1))) v[k] = ld<NT>(src + i + k * 1024ull);
			}
#pragma unroll
			for (int k = 0; k < UNROLL; ++k)
				if (i + k * 1024ull < hi) {
					if (mode == 0 || (c & 1)) st<NT>(dst + i + k * 1024ull, v[k]);
					else acc ^= v[k];
				}
		}
	}
	if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) *sink = 1;
}

int main()
{
	hipDeviceProp_t prop;
	CHECK(hipGetDeviceProperties(&prop, 0));
	const uint64_t cap = 8ull << 30;
	u32x4 *a, *b;
	uint32_t* sink;
	CHECK(hipMalloc(&a, cap));
	CHECK(hipMalloc(&b, cap));
	CHECK(hipMalloc(&sink, 64));
	CHECK(hipMemset(a, 1, cap));
	CHECK(hipMemset(b, 2, cap));
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0));
	CHECK(hipEventCreate(&e1));
	const int cus = prop.multiProcessorCount;
	printf("device %s, %d CUs\n", prop.name, cus);
	auto timed = [&](const char* name, uint64_t bytes, double moved, auto launch) -> int {
		float best = 1e9f;
		for (int rep = 0; rep < 5; ++rep) {
			CHECK(hipEventRecord(e0));
			launch();
			CHECK(hipEventRecord(e1));
			CHECK(hipEventSynchronize(e1));
			float ms;
			CHECK(hipEventElapsedTime(&ms, e0, e1));
			best = ms < best ? ms : best;
		}
		printf("%-58s %4.0f GiB: %7.3f ms  %5.2f TB/s\n", name, bytes / 1073741824.0, best, moved / best / 1e9);
		return 0;
	};
	for (uint64_t bytes : { 2ull << 30, 8ull << 30 }) {
		const uint64_t n = bytes / 16;
		const double copy = 2.0 * bytes;
		for (int waves : { 16, 32 }) {
			char name[96];
			snprintf(name, sizeof name, "grid-stride, %d waves/CU, 4 in flight, plain", waves);
			if (timed(name, bytes, copy, [&] { hipLaunchKernelGGL((copy_gridstride<false, 4>), dim3(cus * waves / 4), dim3(256), 0, 0, a, b, n); })) return 1;
			snprintf(name, sizeof name, "grid-stride, %d waves/CU, 4 in flight, non-temporal", waves);
			if (timed(name, bytes, copy, [&] { hipLaunchKernelGGL((copy_gridstride<true, 4>), dim3(cus * waves / 4), dim3(256), 0, 0, a, b, n); })) return 1;
		}
		if (timed("grid-stride, 32 waves/CU, 8 in flight, plain", bytes, copy, [&] { hipLaunchKernelGGL((copy_gridstride<false, 8>), dim3(cus * 8), dim3(256), 0, 0, a, b, n); })) return 1;
		if (timed("one-shot, 1 x 16 B per thread, plain", bytes, copy, [&] { hipLaunchKernelGGL((copy_oneshot<false, 1>), dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, 0, a, b, n); })) return 1;
		if (timed("one-shot, 4 x 16 B per thread, plain", bytes, copy, [&] { hipLaunchKernelGGL((copy_oneshot<false, 4>), dim3((uint32_t)((n + 1023) / 1024)), dim3(256), 0, 0, a, b, n); })) return 1;
		if (timed("one-shot, 4 x 16 B per thread, non-temporal", bytes, copy, [&] { hipLaunchKernelGGL((copy_oneshot<true, 4>), dim3((uint32_t)((n + 1023) / 1024)), dim3(256), 0, 0, a, b, n); })) return 1;
		if (timed("one-shot, 8 x 16 B per thread, plain", bytes, copy, [&] { hipLaunchKernelGGL((copy_oneshot<false, 8>), dim3((uint32_t)((n + 2047) / 2048)), dim3(256), 0, 0, a, b, n); })) return 1;
		for (uint64_t mib : { 2ull, 8ull }) {
			const uint64_t chunk = (mib << 20) / 16;
			char name[96];
			snprintf(name, sizeof name, "chunked, 1024 threads x %llu MiB, 2 WG/CU, plain", (unsigned long long)mib);
			if (timed(name, bytes, copy, [&] { hipLaunchKernelGGL((copy_chunked<false, 4>), dim3(cus * 2), dim3(1024), 0, 0, a, b, n, chunk, 0, sink); })) return 1;
			snprintf(name, sizeof name, "chunked, 1024 threads x %llu MiB, 2 WG/CU, non-temporal", (unsigned long long)mib);
			if (timed(name, bytes, copy, [&] { hipLaunchKernelGGL((copy_chunked<true, 4>), dim3(cus * 2), dim3(1024), 0, 0, a, b, n, chunk, 0, sink); })) return 1;
			snprintf(name, sizeof name, "split: read-only and write-only WGs, %llu MiB chunks, plain", (unsigned long long)mib);
			// (half of the chunks are read, the other half written: bytes moved = one array's worth in all)
			if (timed(name, bytes, 1.0 * bytes, [&] { hipLaunchKernelGGL((copy_chunked<false, 4>), dim3(cus * 2), dim3(1024), 0, 0, a, b, n, chunk, 1, sink); })) return 1;
			snprintf(name, sizeof name, "split: read-only and write-only WGs, %llu MiB chunks, non-temporal", (unsigned long long)mib);
			if (timed(name, bytes, 1.0 * bytes, [&] { hipLaunchKernelGGL((copy_chunked<true, 4>), dim3(cus * 2), dim3(1024), 0, 0, a, b, n, chunk, 1, sink); })) return 1;
		}
		if (timed("hipMemcpyAsync device to device", bytes, copy, [&] { (void)hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); })) return 1;
	}
	return 0;
}
