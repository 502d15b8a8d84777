// Latency of dependent LDS reads of different shapes, one wave and eight waves per SIMD: a byte at a wave-uniform address, a byte
// per lane at consecutive addresses, a 16-bit value across a dword boundary, an aligned dword per lane.
// usage: hipcc --offload-arch=gfx950 -O3 tools/ubench_lds_bytes.hip -o /tmp/ulb && /tmp/ulb
#include <hip/hip_runtime.h>
#include <cstdio>
typedef uint16_t __attribute__((aligned(1))) u16u;
template <int MODE>
__global__ __launch_bounds__(64) void k(uint32_t* out, uint64_t* cycles, uint32_t start)
{
	__shared__ uint8_t lds[4096];
	const uint32_t l = threadIdx.x;
	for (uint32_t i = l; i < 4096; i += 64)
		lds[i] = (uint8_t)((i * 37 + 11) & 3); // small steps
	__syncthreads();
	uint32_t at = start, acc = 0;
	const uint64_t t0 = clock64();
	for (int i = 0; i < 512; ++i) {
		uint32_t v;
		if (MODE == 0)
			v = lds[at & 4095]; // uniform address
		else if (MODE == 1)
			v = lds[(at + l) & 4095]; // a byte per lane, consecutive
		else if (MODE == 2)
			v = *(const u16u*)(lds + ((at & 4092) | 3)) & 3; // 16 bits across a dword boundary, uniform
		else
			v = *(const uint32_t*)(lds + (((at + l) * 4) & 4092)) & 3; // aligned dword per lane
		const uint32_t u = __builtin_amdgcn_readfirstlane(v);
		acc += v;
		at += 1 + u;
	}
	const uint64_t t1 = clock64();
	out[blockIdx.x * 64 + l] = acc + at;
	if (l == 0)
		cycles[blockIdx.x] = t1 - t0;
}
template <int MODE>
void run(const char* name)
{
	uint32_t* d;
	uint64_t* c;
	hipMalloc(&d, 8192 * 64 * 4);
	hipMalloc(&c, 8192 * 8);
	for (int blocks : { 1, 8192 }) {
		hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, c, 5u);
		hipDeviceSynchronize();
		uint64_t h[8192];
		hipMemcpy(h, c, blocks * 8, hipMemcpyDeviceToHost);
		double s = 0;
		for (int i = 0; i < blocks; ++i)
			s += (double)h[i];
		printf("%-40s %5d waves: %7.1f clock64 ticks per dependent read\n", name, blocks, s / blocks / 512);
	}
}
int main()
{
	run<0>("byte, wave-uniform address");
	run<1>("byte per lane, consecutive");
	run<2>("16 bits across a dword boundary, uniform");
	run<3>("aligned dword per lane");
	return 0;
}
