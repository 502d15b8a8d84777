// ubench_stage.hip -- does a staging round trip through a small, quickly reused buffer stay in a cache (L2 / Infinity Cache)
// on this device, next to streaming traffic?  The traffic pattern of the fused encoder: every workgroup streams 16 KiB of
// input per step, writes 6.5 KiB (the encoded bytes) to its private ring, reads back what it wrote LAG steps earlier and
// writes that to the output stream.  The ring's size sets the staging footprint; the time per launch tells whether the
// round trip costs HBM bandwidth.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_stage.hip -o build/ubench_stage && build/ubench_stage
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// per step and workgroup (256 threads): 4 x 4 KiB input groups read, 416 groups (6.5 KiB) staged / read back / written
constexpr int IN_GROUPS = 1024, OUT_GROUPS = 416;
template <bool STAGED>
__global__ __launch_bounds__(256) void kern(const u32x4* __restrict__ src, u32x4* __restrict__ dst, u32x4* __restrict__ stage, uint64_t steps_total, uint32_t ring_groups,
					     uint32_t lag_groups, uint32_t* sink)
{
	const uint32_t t = threadIdx.x;
	u32x4* ring = stage + (uint64_t)blockIdx.x * ring_groups;
	uint32_t wpos = 0;
	u32x4 acc = { 0, 0, 0, 0 };
	for (uint64_t s = blockIdx.x; s < steps_total; s += gridDim.x) {
		const u32x4* in = src + s * IN_GROUPS;
		u32x4 v[4];
#pragma unroll
		for (int k = 0; k < 4; ++k)
			v[k] = __builtin_nontemporal_load(in + k * 256 + t);
		acc ^= v[2] ^ v[3];
		u32x4* out = dst + s * OUT_GROUPS;
		if (STAGED) {
			// stage 416 groups (two rounds, the second partial), read back the groups written lag_groups earlier, store those
			const uint32_t a0 = (wpos + t) % ring_groups, a1 = (wpos + 256 + t) % ring_groups;
			ring[a0] = v[0];
			if (t < OUT_GROUPS - 256) ring[a1] = v[1];
			const uint32_t r0 = (wpos + ring_groups - lag_groups + t) % ring_groups, r1 = (wpos + ring_groups - lag_groups + 256 + t) % ring_groups;
			const u32x4 b0 = ring[r0];
			u32x4 b1 = acc;
			if (t < OUT_GROUPS - 256) b1 = ring[r1];
			__builtin_nontemporal_store(b0, out + t);
			if (t < OUT_GROUPS - 256) __builtin_nontemporal_store(b1, out + 256 + t);
			wpos = (wpos + OUT_GROUPS) % ring_groups;
		}
		else {
			__builtin_nontemporal_store(v[0], out + t);
			if (t < OUT_GROUPS - 256) __builtin_nontemporal_store(v[1], out + 256 + t);
		}
	}
	if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) *sink = 1;
}

int main()
{
	hipDeviceProp_t prop;
	CHECK(hipGetDeviceProperties(&prop, 0));
	const uint64_t bytes = 8ull << 30, steps = bytes / (IN_GROUPS * 16);
	u32x4 *a, *b, *st;
	uint32_t* sink;
	const uint64_t stage_bytes = 1ull << 30;
	CHECK(hipMalloc(&a, bytes));
	CHECK(hipMalloc(&b, bytes));
	CHECK(hipMalloc(&st, stage_bytes));
	CHECK(hipMalloc(&sink, 64));
	CHECK(hipMemset(a, 1, bytes));
	CHECK(hipMemset(b, 2, bytes));
	CHECK(hipMemset(st, 3, stage_bytes));
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0));
	CHECK(hipEventCreate(&e1));
	const int grid = prop.multiProcessorCount * 8;
	auto run = [&](bool staged, uint32_t ring_groups, uint32_t lag_groups, const char* what) -> int {
		float best = 1e9f;
		for (int rep = 0; rep < 4; ++rep) {
			CHECK(hipEventRecord(e0));
			if (staged)
				hipLaunchKernelGGL(kern<true>, dim3(grid), dim3(256), 0, 0, a, b, st, steps, ring_groups, lag_groups, sink);
			else
				hipLaunchKernelGGL(kern<false>, dim3(grid), dim3(256), 0, 0, a, b, st, steps, ring_groups, lag_groups, sink);
			CHECK(hipEventRecord(e1));
			CHECK(hipEventSynchronize(e1));
			float ms;
			CHECK(hipEventElapsedTime(&ms, e0, e1));
			best = ms < best ? ms : best;
		}
		printf("%-60s %.3f ms\n", what, best);
		return 0;
	};
	run(false, 0, 0, "no staging: 8.6 GB read, 3.5 GB written");
	char buf[200];
	for (uint32_t ring_kb : { 13, 26, 52, 104, 208, 416 }) { // ring per workgroup; x 2048 workgroups = footprint
		const uint32_t ring_groups = ring_kb * 1024 / 16 / OUT_GROUPS * OUT_GROUPS; // a whole number of steps
		for (uint32_t lag_steps : { 1u, ring_groups / OUT_GROUPS - 1 }) {
			if (lag_steps == 0) continue;
			snprintf(buf, sizeof buf, "staged: footprint %4.0f MB, read back %3u steps (%5.1f KB) later", (double)grid * ring_groups * 16 / 1e6, lag_steps, lag_steps * OUT_GROUPS * 16 / 1024.0);
			if (run(true, ring_groups, lag_steps * OUT_GROUPS, buf)) return 1;
		}
	}
	return 0;
}
