// ubench_ticket.hip -- price of work tickets on gfx950: every workgroup of a resident grid takes `rounds` tickets with a
// returning device-scope atomicAdd, (a) all from one counter, (b) from 8 / 32 counters chosen by blockIdx % shards
// (round-robin dispatch puts blockIdx % 8 on one XCD), each counter on a cache line of its own.  Also prints which XCC
// the first workgroups ran on (HW_REG_XCC_ID).  Decides the chunk size of the streaming encoder.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_ticket.hip -o /tmp/ubench_ticket && /tmp/ubench_ticket
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define CHECK(x)                                                                                                       \
	do {                                                                                                               \
		hipError_t e = (x);                                                                                            \
		if (e != hipSuccess) {                                                                                         \
			printf("%s: %s\n", #x, hipGetErrorString(e));                                                              \
			return 1;                                                                                                  \
		}                                                                                                              \
	} while (0)

__global__ void __launch_bounds__(256) take(uint32_t* counters, uint32_t shards, uint32_t rounds, uint32_t work, uint32_t* sink, uint32_t* xcc)
{
	__shared__ uint32_t t;
	uint32_t acc = threadIdx.x;
	uint32_t* c = counters + (blockIdx.x % shards) * 32; // 128 bytes apart
	for (uint32_t r = 0; r < rounds; ++r) {
		if (threadIdx.x == 0)
			t = atomicAdd(c, 1u);
		__syncthreads();
		acc += t;
		for (uint32_t k = 0; k < work; ++k) // stand-in for the work of one ticket
			acc = acc * 1664525u + 1013904223u;
		__syncthreads();
	}
	if (acc == 0x12345u)
		sink[0] = acc;
	if (threadIdx.x == 0 && blockIdx.x < 64) {
		uint32_t id;
		asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
		xcc[blockIdx.x] = id;
	}
}

int main()
{
	uint32_t *counters, *sink, *xcc;
	CHECK(hipMalloc(&counters, 32 * 128));
	CHECK(hipMalloc(&sink, 64));
	CHECK(hipMalloc(&xcc, 64 * 4));
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0));
	CHECK(hipEventCreate(&e1));
	const uint32_t grid = 2048, rounds = 512;
	for (uint32_t work : { 0u, 200u, 1000u })
		for (uint32_t shards : { 1u, 8u, 32u }) {
			float best = 1e9f;
			for (int rep = 0; rep < 3; ++rep) {
				CHECK(hipMemset(counters, 0, 32 * 128));
				CHECK(hipEventRecord(e0));
				hipLaunchKernelGGL(take, dim3(grid), dim3(256), 0, 0, counters, shards, rounds, work, sink, xcc);
				CHECK(hipEventRecord(e1));
				CHECK(hipEventSynchronize(e1));
				float ms;
				CHECK(hipEventElapsedTime(&ms, e0, e1));
				best = ms < best ? ms : best;
			}
			printf("work %4u shards %2u: %8.3f ms for %u tickets = %7.1f tickets/us\n", work, shards, best, grid * rounds, grid * rounds / (best * 1000.0f));
		}
	uint32_t h[64];
	CHECK(hipMemcpy(h, xcc, sizeof(h), hipMemcpyDeviceToHost));
	printf("xcc of blocks 0..31:");
	for (int i = 0; i < 32; ++i)
		printf(" %u", h[i] & 15u);
	printf("\n");
	return 0;
}
