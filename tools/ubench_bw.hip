// ubench_bw.hip -- what a plain streaming kernel reaches on this device for the mixes of reads and writes the codec produces:
// read only, write only, copy (1:1), the fused encoder's mix (12 read : 7 written) and the decoder's (3.4 : 8.6).
// Every lane moves 16 bytes per access; UNROLL accesses are in flight per lane; the grid is persistent (CUs x waves).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_bw.hip -o build/ubench_bw && build/ubench_bw
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// reads nr 16-byte groups from src and writes nw groups to dst, interleaved in chunks of 64 KiB per workgroup iteration
template <int UNROLL>
__global__ __launch_bounds__(256) void stream(const u32x4* __restrict__ src, u32x4* __restrict__ dst, uint64_t nr, uint64_t nw, uint32_t* sink)
{
	const uint64_t tid = blockIdx.x * 256ull + threadIdx.x, step = gridDim.x * 256ull;
	u32x4 acc = { 0, 0, 0, 0 };
	const uint64_t n = nr > nw ? nr : nw;
	for (uint64_t i = tid; i < n; i += step * UNROLL) {
		u32x4 v[UNROLL];
#pragma unroll
		for (int k = 0; k < UNROLL; ++k) {
			const uint64_t j = i + (uint64_t)k * step;
			if (j < nr) v[k] = __builtin_nontemporal_load(src + j); else v[k] = acc;
		}
#pragma unroll
		for (int k = 0; k < UNROLL; ++k) {
			const uint64_t j = i + (uint64_t)k * step;
			if (j < nw) __builtin_nontemporal_store(v[k], dst + j); else acc ^= v[k];
		}
	}
	if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) *sink = 1;
}

int main()
{
	hipDeviceProp_t prop;
	CHECK(hipGetDeviceProperties(&prop, 0));
	const uint64_t bytes = 8ull << 30, groups = bytes / 16;
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
	struct Mix { const char* name; double r, w; } mixes[] = { { "read only", 1, 0 }, { "write only", 0, 1 }, { "copy 1:1", 1, 1 }, { "encoder 12:7", 1, 7.0 / 12 }, { "encoder without staging 8.6:3.4", 1, 3.4 / 8.6 }, { "decoder 3.4:8.6", 3.4 / 8.6, 1 } };
	for (const Mix& m : mixes)
		for (int waves : { 8, 16, 32 }) {
			const uint64_t nr = (uint64_t)(groups * m.r), nw = (uint64_t)(groups * m.w);
			const int grid = prop.multiProcessorCount * waves / 4;
			float best = 1e9f;
			for (int rep = 0; rep < 4; ++rep) {
				CHECK(hipEventRecord(e0));
				hipLaunchKernelGGL(stream<4>, dim3(grid), dim3(256), 0, 0, a, b, nr, nw, sink);
				CHECK(hipEventRecord(e1));
				CHECK(hipEventSynchronize(e1));
				float ms;
				CHECK(hipEventElapsedTime(&ms, e0, e1));
				best = ms < best ? ms : best;
			}
			printf("%-34s %2d waves/CU: %.3f ms, %.2f TB/s (read %.2f + written %.2f GB)\n", m.name, waves, best, (nr + nw) * 16.0 / best / 1e9, nr * 16 / 1e9, nw * 16 / 1e9);
		}
	return 0;
}
