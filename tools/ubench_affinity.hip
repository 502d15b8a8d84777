// ubench_affinity.hip -- does it matter WHICH workgroup touches which addresses?  A one-shot copy (one workgroup of 256 threads
// per 4 KiB, one 16-byte access per thread: the form that reaches 6.3 TB/s, tools/ubench_copy.hip) with the workgroup -> chunk
// mapping permuted: workgroups are dealt round-robin over the 8 XCDs, so chunk = blockIdx.x keeps "chunk index mod 8 = XCD";
// a rotation by r breaks that pairing while every XCD still streams 4 KiB pieces 32 KiB apart.  Also: chunk sizes of 1, 2, 8,
// 16 KiB per workgroup (64, 128, 512, 1024 threads).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_affinity.hip -o build/ubench_affinity && build/ubench_affinity
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// mode 0: chunk = b; 1: chunk = b rotated by rot inside its group of 8; 2: chunk = b ^ rot; 3: blocks of an XCD take a
// contiguous eighth of the array (XCD x: chunks [x * n/8, (x+1) * n/8))
template <int THREADS>
__global__ __launch_bounds__(THREADS) void copy_perm(const u32x4* __restrict__ src, u32x4* __restrict__ dst, uint64_t nchunks, int mode, uint32_t rot)
{
	uint64_t b = blockIdx.x, c = b;
	if (mode == 1) c = (b & ~7ull) | ((b + rot) & 7ull);
	else if (mode == 2) c = b ^ rot;
	else if (mode == 3) c = (b & 7ull) * (nchunks / 8) + (b >> 3);
	if (c >= nchunks) return;
	const uint64_t i = c * THREADS + threadIdx.x;
	dst[i] = src[i];
}
template <int THREADS>
__global__ __launch_bounds__(THREADS) void read_perm(const u32x4* __restrict__ src, uint64_t nchunks, int mode, uint32_t rot, uint32_t* sink)
{
	uint64_t b = blockIdx.x, c = b;
	if (mode == 1) c = (b & ~7ull) | ((b + rot) & 7ull);
	if (c >= nchunks) return;
	const u32x4 v = src[c * THREADS + threadIdx.x];
	if ((v.x ^ v.y ^ v.z ^ v.w) == 0x12345u) *sink = 1;
}

int main()
{
	const uint64_t bytes = 8ull << 30;
	u32x4 *a, *b;
	uint32_t* sink;
	CHECK(hipMalloc(&a, bytes));
	CHECK(hipMalloc(&b, bytes));
	CHECK(hipMalloc(&sink, 64));
	CHECK(hipMemset(a, 1, bytes));
	CHECK(hipMemset(b, 2, bytes));
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0));
	CHECK(hipEventCreate(&e1));
	auto timed = [&](const char* name, double moved, auto launch) -> int {
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
		printf("%-64s %7.3f ms  %5.2f TB/s\n", name, best, moved / best / 1e9);
		return 0;
	};
	char name[128];
	const uint64_t n4k = bytes / 4096;
	for (uint32_t rot : { 0u, 1u, 2u, 4u, 7u }) {
		snprintf(name, sizeof name, "copy, 4 KiB per workgroup, chunk rotated by %u within 8", rot);
		if (timed(name, 2.0 * bytes, [&] { hipLaunchKernelGGL(copy_perm<256>, dim3((uint32_t)n4k), dim3(256), 0, 0, a, b, n4k, 1, rot); })) return 1;
	}
	for (uint32_t x : { 8u, 64u, 1024u }) {
		snprintf(name, sizeof name, "copy, 4 KiB per workgroup, chunk = block ^ %u", x);
		if (timed(name, 2.0 * bytes, [&] { hipLaunchKernelGGL(copy_perm<256>, dim3((uint32_t)n4k), dim3(256), 0, 0, a, b, n4k, 2, x); })) return 1;
	}
	if (timed("copy, 4 KiB per workgroup, every XCD a contiguous eighth", 2.0 * bytes, [&] { hipLaunchKernelGGL(copy_perm<256>, dim3((uint32_t)n4k), dim3(256), 0, 0, a, b, n4k, 3, 0u); })) return 1;
	if (timed("copy, 1 KiB per workgroup (64 threads)", 2.0 * bytes, [&] { hipLaunchKernelGGL(copy_perm<64>, dim3((uint32_t)(bytes / 1024)), dim3(64), 0, 0, a, b, bytes / 1024, 0, 0u); })) return 1;
	if (timed("copy, 2 KiB per workgroup (128 threads)", 2.0 * bytes, [&] { hipLaunchKernelGGL(copy_perm<128>, dim3((uint32_t)(bytes / 2048)), dim3(128), 0, 0, a, b, bytes / 2048, 0, 0u); })) return 1;
	if (timed("copy, 8 KiB per workgroup (512 threads)", 2.0 * bytes, [&] { hipLaunchKernelGGL(copy_perm<512>, dim3((uint32_t)(bytes / 8192)), dim3(512), 0, 0, a, b, bytes / 8192, 0, 0u); })) return 1;
	if (timed("copy, 16 KiB per workgroup (1024 threads)", 2.0 * bytes, [&] { hipLaunchKernelGGL(copy_perm<1024>, dim3((uint32_t)(bytes / 16384)), dim3(1024), 0, 0, a, b, bytes / 16384, 0, 0u); })) return 1;
	for (uint32_t rot : { 0u, 1u, 4u }) {
		snprintf(name, sizeof name, "read only, 4 KiB per workgroup, chunk rotated by %u within 8", rot);
		if (timed(name, 1.0 * bytes, [&] { hipLaunchKernelGGL(read_perm<256>, dim3((uint32_t)n4k), dim3(256), 0, 0, a, n4k, 1, rot, sink); })) return 1;
	}
	return 0;
}
