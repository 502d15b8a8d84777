// ubench_pattern.hip -- the fused encoder's memory access PATTERN without its arithmetic: resident workgroups of four waves
// take superblocks of 128 KiB; every wave streams its own contiguous 32 KiB of the superblock in steps of 2 KiB (two
// blocks), appends 0.8 KiB per step to its staging stream, and when the superblock is done copies the 13 KiB it staged for the
// previous one to the output stream.  Same bytes as tools/ubench_stage.hip; what differs is who touches what when.
//   variant 0: as the encoder does it          variant 1: the four waves of a workgroup interleave their steps (wave w takes
//   steps w, w + 4, ... of the superblock: the workgroup reads 8 KiB of consecutive addresses at a time)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// VARIANT 2: as 0, with the loads of the next step requested before the current step's bytes are used (two steps in flight)
// VARIANT 3: as 0 without staging: the bytes go straight to the output stream (what a known offset would allow)
// VARIANT 4: as 2 and 3 together
template <int VARIANT>
__global__ __launch_bounds__(256) void kern(const u32x4* __restrict__ src, u32x4* __restrict__ dst, u32x4* __restrict__ stage, uint32_t nsb, uint32_t* ticket, uint32_t* sink)
{
	const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	__shared__ uint32_t sb_s;
	u32x4 acc = { 0, 0, 0, 0 };
	u32x4* my_stage = stage + ((uint64_t)blockIdx.x * 4 + w) * 2 * 1024; // two buffers of 16 KiB per wave
	uint32_t parity = 0;
	int64_t prev = -1;
	for (;;) {
		if (threadIdx.x == 0) sb_s = atomicAdd(ticket, 1u);
		__syncthreads();
		const uint32_t sb = sb_s;
		__syncthreads();
		if (sb < nsb) {
			const u32x4* in = src + (uint64_t)sb * 8192; // 128 KiB = 8192 groups
			u32x4* st = my_stage + parity * 1024;
			if (VARIANT == 3 || VARIANT == 4)
				st = dst + (uint64_t)sb * 3328 + w * 832;
			if (VARIANT == 2 || VARIANT == 4) {
				u32x4 a = in[(w * 16) * 128 + lane], b = in[(w * 16) * 128 + 64 + lane];
				for (uint32_t p = 0; p < 16; ++p) {
					const uint32_t nx = w * 16 + (p < 15 ? p + 1 : p);
					const u32x4 na = in[nx * 128 + lane], nb = in[nx * 128 + 64 + lane];
					acc ^= b;
					if (lane < 52) st[p * 52 + lane] = a ^ acc;
					__builtin_amdgcn_sched_barrier(0);
					a = na;
					b = nb;
				}
			}
			else
			for (uint32_t p = 0; p < 16; ++p) { // 16 steps of two blocks per wave
				const uint32_t step = VARIANT != 1 ? w * 16 + p : p * 4 + w;
				const u32x4 a = in[step * 128 + lane], b = in[step * 128 + 64 + lane];
				acc ^= b;
				if (lane < 52) st[p * 52 + lane] = a ^ acc; // 0.8 KiB staged per step
				__builtin_amdgcn_sched_barrier(0);
			}
		}
		if (prev >= 0 && VARIANT != 3 && VARIANT != 4) { // copy the previous superblock's run: 832 groups = 13 KiB per wave
			const u32x4* st = my_stage + (parity ^ 1) * 1024;
			u32x4* out = dst + (uint64_t)prev * 3328 + w * 832;
			for (uint32_t o = 0; o < 832; o += 256) {
				u32x4 v[4];
				for (int k = 0; k < 4; ++k) v[k] = o + k * 64 + lane < 832 ? st[o + k * 64 + lane] : acc;
				for (int k = 0; k < 4; ++k) if (o + k * 64 + lane < 832) out[o + k * 64 + lane] = v[k];
			}
		}
		if (sb >= nsb) break;
		prev = sb;
		parity ^= 1;
	}
	if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) *sink = 1;
}

int main()
{
	hipDeviceProp_t prop;
	CHECK(hipGetDeviceProperties(&prop, 0));
	const uint64_t bytes = 8ull << 30;
	const uint32_t nsb = (uint32_t)(bytes / 131072);
	u32x4 *a, *b, *st;
	uint32_t *sink, *ticket;
	const int grid = prop.multiProcessorCount * 8;
	const uint64_t stage_bytes = (uint64_t)grid * 4 * 2 * 16384;
	CHECK(hipMalloc(&a, bytes));
	CHECK(hipMalloc(&b, bytes));
	CHECK(hipMalloc(&st, stage_bytes));
	CHECK(hipMalloc(&sink, 64));
	CHECK(hipMalloc(&ticket, 64));
	CHECK(hipMemset(a, 1, bytes));
	CHECK(hipMemset(b, 2, bytes));
	CHECK(hipMemset(st, 3, stage_bytes));
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0));
	CHECK(hipEventCreate(&e1));
	for (int variant = 0; variant < 5; ++variant) {
		float best = 1e9f;
		for (int rep = 0; rep < 4; ++rep) {
			CHECK(hipMemset(ticket, 0, 4));
			CHECK(hipEventRecord(e0));
			switch (variant) {
				case 0: hipLaunchKernelGGL(kern<0>, dim3(grid), dim3(256), 0, 0, a, b, st, nsb, ticket, sink); break;
				case 1: hipLaunchKernelGGL(kern<1>, dim3(grid), dim3(256), 0, 0, a, b, st, nsb, ticket, sink); break;
				case 2: hipLaunchKernelGGL(kern<2>, dim3(grid), dim3(256), 0, 0, a, b, st, nsb, ticket, sink); break;
				case 3: hipLaunchKernelGGL(kern<3>, dim3(grid), dim3(256), 0, 0, a, b, st, nsb, ticket, sink); break;
				default: hipLaunchKernelGGL(kern<4>, dim3(grid), dim3(256), 0, 0, a, b, st, nsb, ticket, sink); break;
			}
			CHECK(hipEventRecord(e1));
			CHECK(hipEventSynchronize(e1));
			float ms;
			CHECK(hipEventElapsedTime(&ms, e0, e1));
			best = ms < best ? ms : best;
		}
		printf("variant %d: %.3f ms (8.6 GB read, 3.5 GB staged and read back, 3.5 GB written; staging footprint %.0f MB)\n", variant, best, stage_bytes / 1e6);
	}
	return 0;
}
