/* tests/link/abi_link.c -- TEST INFRASTRUCTURE.  A plain C translation unit compiled against include/stenos.h and
 * linked with -lstenos: proves that the header is valid C, that every declared symbol resolves, and that the
 * calling convention agrees (ctypes bindings cannot show that).  Mirrors the README usage of the reference
 * (README.md "Usage"; stenos/stenos.h:115-301).  Prints ABI_LINK_OK and returns 0 on success. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "stenos.h"
#include "stenos_hip.h"

int main(void)
{
	const size_t n = 100000;
	int* v = (int*)malloc(n * sizeof(int));
	size_t i, bound, r, d;
	void *dst, *back;
	stenos_context* ctx;
	stenos_info info;
	stenos_timer* t;
	if (!v)
		return 2;
	for (i = 0; i < n; ++i)
		v[i] = (int)(i * 7 % 1000);
	bound = stenos_bound(n * sizeof(int));
	dst = malloc(bound);
	back = malloc(n * sizeof(int));
	if (!dst || !back)
		return 2;
	if (stenos_hip_device_count() < 1) {
		printf("no HIP device\n");
		return 3;
	}
	/* one-shot calls */
	r = stenos_compress(v, sizeof(int), n * sizeof(int), dst, bound, 1);
	if (stenos_has_error(r))
		return 4;
	if (stenos_get_info(dst, sizeof(int), r, &info) != 8 /* the frame header size */ || info.decompressed_size != n * sizeof(int))
		return 5;
	d = stenos_decompress(dst, sizeof(int), r, back, n * sizeof(int));
	if (d != n * sizeof(int) || memcmp(v, back, d) != 0)
		return 6;
	/* context calls, every setter */
	ctx = stenos_make_context();
	if (!ctx)
		return 7;
	if (stenos_set_level(ctx, 2) != 0 || stenos_set_threads(ctx, 4) != 0 || stenos_set_max_nanoseconds(ctx, 0) != 0 || stenos_set_block_size(ctx, 3) != 0)
		return 8;
	r = stenos_compress_generic(ctx, v, sizeof(int), n * sizeof(int), dst, bound);
	if (stenos_has_error(r))
		return 9;
	memset(back, 0, n * sizeof(int));
	d = stenos_decompress_generic(ctx, dst, sizeof(int), r, back, n * sizeof(int));
	if (d != n * sizeof(int) || memcmp(v, back, d) != 0)
		return 10;
	(void)stenos_memory_footprint(ctx);
	stenos_reset_context(ctx);
	/* the private single-superblock calls stenos::cvector uses */
	r = stenos_private_compress_block(ctx, v, sizeof(int), 4096, 4096, dst, bound);
	if (stenos_has_error(r) || stenos_private_block_size(dst, r) != r)
		return 11;
	d = stenos_private_decompress_block(ctx, dst, sizeof(int), 4096, r, back, 4096);
	if (d != 4096 || memcmp(v, back, 4096) != 0)
		return 12;
	stenos_destroy_context(ctx);
	t = stenos_make_timer();
	stenos_tick(t);
	(void)stenos_tock(t);
	stenos_destroy_timer(t);
	free(v);
	free(dst);
	free(back);
	printf("ABI_LINK_OK\n");
	return 0;
}
