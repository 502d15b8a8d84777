/*
 * stenos.h -- C ABI of the MI355X-native Stenos block codec (libstenos.so built from stenos_amd/csrc).
 *
 * This is the drop-in boundary: the same exported names, signatures, constants and error codes as
 * the reference library's public header (reference: stenos/stenos.h:57-84 constants, :115-301
 * functions), so a program linked against the reference's `stenos` library links against this one
 * unchanged.  The block codec behind stenos_compress / stenos_compress_generic / stenos_decompress*
 * runs on the GPU (hand-written HIP for gfx950); there is no CPU codec in this library.
 *
 * Declarations are restated here (not copied) with the reference location each one replaces.
 */
#ifndef STENOS_H
#define STENOS_H

#include <stddef.h>
#include <stdint.h>

#if defined(__GNUC__)
#define STENOS_EXPORT __attribute__((visibility("default")))
#else
#define STENOS_EXPORT
#endif

/* reference stenos/stenos.h:57, 61, 65, 70 */
#define STENOS_BLOCK_SIZE (131072)
#define STENOS_MAX_BLOCK_BYTES ((1u << 24u) - 1u)
#define STENOS_MAX_BYTESOFTYPE (STENOS_MAX_BLOCK_BYTES / 256)
#define STENOS_NO_BLOCK_SHIFT ((size_t)-1)

/* reference stenos/stenos.h:75-84 */
#define STENOS_ERROR_UNDEFINED ((size_t)(-1))
#define STENOS_ERROR_SRC_OVERFLOW ((size_t)(-2))
#define STENOS_ERROR_ALLOC ((size_t)(-3))
#define STENOS_ERROR_INVALID_INPUT ((size_t)(-4))
#define STENOS_ERROR_INVALID_INSTRUCTION_SET ((size_t)(-5)) /* here: no usable gfx950 device */
#define STENOS_ERROR_DST_OVERFLOW ((size_t)(-6))
#define STENOS_ERROR_INVALID_BYTESOFTYPE ((size_t)(-7))
#define STENOS_ERROR_ZSTD_INTERNAL ((size_t)(-8))
#define STENOS_ERROR_INVALID_PARAMETER ((size_t)(-9))
#define STENOS_LAST_ERROR_CODE ((size_t)(-100))

#ifdef __cplusplus
namespace stenos
{
	/* reference stenos/stenos.h:37-42 */
	inline constexpr size_t compress_bound(size_t bytes)
	{
		return 12 + ((bytes / 65792 + (bytes % 65792 ? 1 : 0)) == 0 ? 1 : (bytes / 65792 + (bytes % 65792 ? 1 : 0))) * 4 + bytes;
	}
}
extern "C" {
#endif

typedef struct stenos_context_s stenos_context; /* reference stenos/stenos.h:103 */

STENOS_EXPORT stenos_context* stenos_make_context(void);                                       /* :115 */
STENOS_EXPORT void stenos_destroy_context(stenos_context* ctx);                                /* :122 */
STENOS_EXPORT void stenos_reset_context(stenos_context* ctx);                                  /* :127 */
STENOS_EXPORT size_t stenos_set_level(stenos_context* ctx, int level);                         /* :135 */
STENOS_EXPORT size_t stenos_set_threads(stenos_context* ctx, int threads);                     /* :140 */
STENOS_EXPORT size_t stenos_set_max_nanoseconds(stenos_context* ctx, uint64_t nanoseconds);    /* :156 */
STENOS_EXPORT size_t stenos_set_block_size(stenos_context* ctx, size_t blocksize_shift);       /* :168 */
STENOS_EXPORT size_t stenos_memory_footprint(stenos_context* ctx);                             /* :173 */
STENOS_EXPORT int stenos_has_error(size_t r);                                                  /* :180 */
STENOS_EXPORT size_t stenos_bound(size_t bytes);                                               /* :185 */
/* :198 */
STENOS_EXPORT size_t stenos_compress_generic(stenos_context* ctx, const void* src, size_t bytesoftype, size_t bytes, void* dst, size_t dst_size);
/* :211 */
STENOS_EXPORT size_t stenos_decompress_generic(stenos_context* ctx, const void* src, size_t bytesoftype, size_t bytes, void* dst, size_t dst_size);
/* :224 */
STENOS_EXPORT size_t stenos_compress(const void* src, size_t bytesoftype, size_t bytes, void* dst, size_t dst_size, int level);
/* :237 */
STENOS_EXPORT size_t stenos_decompress(const void* src, size_t bytesoftype, size_t bytes, void* dst, size_t dst_size);

typedef struct stenos_info_s /* :242-246 */
{
	size_t decompressed_size;
	size_t superblock_size;
} stenos_info;
STENOS_EXPORT size_t stenos_get_info(const void* src, size_t bytesoftype, size_t bytes, stenos_info* info); /* :256 */

typedef struct stenos_timer_s stenos_timer;              /* :266 */
STENOS_EXPORT stenos_timer* stenos_make_timer(void);     /* :272 */
STENOS_EXPORT void stenos_destroy_timer(stenos_timer*);  /* :278 */
STENOS_EXPORT void stenos_tick(stenos_timer*);           /* :283 */
STENOS_EXPORT uint64_t stenos_tock(stenos_timer*);       /* :288 */

/* private API used by stenos::cvector, reference stenos/stenos.h:294-301 */
STENOS_EXPORT size_t stenos_private_compress_block(stenos_context* ctx, const void* src, size_t bytesoftype, size_t super_block_size, size_t bytes, void* dst, size_t dst_size);
STENOS_EXPORT size_t stenos_private_decompress_block(stenos_context* ctx, const void* src, size_t bytesoftype, size_t super_block_size, size_t bytes, void* dst, size_t dst_size);
STENOS_EXPORT size_t stenos_private_block_size(const void* src, size_t src_size);
STENOS_EXPORT size_t stenos_private_block_csize(const void* src);
STENOS_EXPORT size_t stenos_private_create_compression_header(size_t decompressed_size, size_t super_block_size, void* dst, size_t dst_size);

#ifdef __cplusplus
}
#endif
#endif
