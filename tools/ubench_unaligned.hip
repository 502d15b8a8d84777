// Does the LDS do unaligned 16/32/64-bit reads and writes on this device?  (The compiler emits them for align-1 types on
// gfx950; this checks the hardware mode the runtime set up.)  usage: hipcc --offload-arch=gfx950 tools/ubench_unaligned.hip -o /tmp/ua && /tmp/ua
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef uint16_t __attribute__((aligned(1))) u16u;
typedef uint32_t __attribute__((aligned(1))) u32u;
typedef uint64_t __attribute__((aligned(1))) u64u;
__global__ void k(uint64_t* out)
{
	__shared__ uint8_t lds[2048];
	const uint32_t l = threadIdx.x;
	for (uint32_t i = l; i < 2048; i += 64)
		lds[i] = (uint8_t)(i * 7 + 3);
	__syncthreads();
	const uint32_t off = l * 13 + (l & 7); // every misalignment
	out[l] = *(const u16u*)(lds + off);
	out[64 + l] = *(const u32u*)(lds + off + 1);
	out[128 + l] = *(const u64u*)(lds + off + 3);
	__syncthreads();
	// writes: lane l writes 3 bytes-ish pieces at odd places, then everything is read back bytewise
	*(u16u*)(lds + 1024 + l * 9 + 1) = (uint16_t)(0xA000u + l);
	*(u32u*)(lds + 1024 + l * 9 + 3) = 0xB0000000u + l * 0x010101u;
	__syncthreads();
	uint64_t v = 0;
	for (int b = 0; b < 8; ++b)
		v |= (uint64_t)lds[1024 + l * 9 + b] << (8 * b);
	out[192 + l] = v;
}
int main()
{
	uint64_t* d;
	hipMalloc(&d, 256 * 8);
	hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
	uint64_t h[256];
	hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
	uint8_t ref[2048];
	for (uint32_t i = 0; i < 2048; ++i)
		ref[i] = (uint8_t)(i * 7 + 3);
	int bad = 0;
	for (uint32_t l = 0; l < 64; ++l) {
		const uint32_t off = l * 13 + (l & 7);
		uint16_t a;
		uint32_t b;
		uint64_t c;
		memcpy(&a, ref + off, 2);
		memcpy(&b, ref + off + 1, 4);
		memcpy(&c, ref + off + 3, 8);
		bad += h[l] != a;
		bad += h[64 + l] != b;
		bad += h[128 + l] != c;
	}
	for (uint32_t l = 0; l < 64; ++l) {
		uint8_t w[16];
		for (int b = 0; b < 16; ++b)
			w[b] = ref[1024 + l * 9 + b];
		// what lane l-1 wrote into the first byte(s) is not modelled: compare bytes 1..6 only
		uint16_t a = (uint16_t)(0xA000u + l);
		uint32_t b32 = 0xB0000000u + l * 0x010101u;
		memcpy(w + 1, &a, 2);
		memcpy(w + 3, &b32, 4);
		for (int b = 1; b < 7; ++b)
			bad += ((h[192 + l] >> (8 * b)) & 0xFF) != w[b];
	}
	printf("unaligned LDS access: %s (%d mismatches)\n", bad ? "WRONG" : "ok", bad);
	return bad != 0;
}
