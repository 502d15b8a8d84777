// tests/emul/emul.cpp -- TEST INFRASTRUCTURE.  Compiles the wave-level codec sources of the product
// (stenos_amd/csrc/*.h) for the host with WV_HOST_EMULATION: 64 lanes executed in lockstep, LDS as a
// plain buffer.  Lets the CPU test-suite diff the exact kernel logic against the oracle without a GPU.
// The shipped library never contains or calls this.
#define WV_HOST_EMULATION 1
#include "../../stenos_amd/csrc/superblock_codec.h"

#include <stdlib.h>
#include <string.h>

using namespace codec;

static uint8_t* alloc_lds(uint32_t bytes)
{
	void* p = nullptr;
	if (posix_memalign(&p, 64, bytes + 256))
		return nullptr;
	memset(p, 0xCD, bytes + 256); // LDS is not zero-initialised on the device either
	return (uint8_t*)p;
}

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
		uint32_t n = encode_block_job(lds, L, (uint32_t)T, stage, slot, allow_lz != 0);
		memcpy(dst + off, slot, n);
		off += n;
	}
	size_t rem = bytes - nb * bs;
	if (rem) {
		memcpy(stage, src + nb * bs, rem);
		uint32_t n = encode_tail_job(lds, L, (uint32_t)T, stage, (uint32_t)rem, slot);
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
	uint32_t r = decode_superblock(lds, L, (uint32_t)T, in + 16 + misalign, (uint32_t)csize, out, (uint32_t)dsize);
	if (r != DEC_ERROR)
		memcpy(dst, out, dsize);
	free(in);
	free(out);
	free(lds);
	return r == DEC_ERROR ? (size_t)-4 : r;
}

void emul_copy_g2g(uint8_t* dst, const uint8_t* src, size_t n) { copy_g2g(dst, src, (uint32_t)n); }

size_t emul_lds_bytes_encode(size_t T) { return make_layout((uint32_t)T, true).total; }
size_t emul_lds_bytes_decode(size_t T) { return make_dec_layout((uint32_t)T).total; }
}
