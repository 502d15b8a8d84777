/*
 * stenos_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C, scalar CPU restatement of the Stenos level-1 block codec (reference:
 * Thermadiag/stenos v0.2, stenos/internal/block_compress.h, lz_compress.h, stenos.cpp).
 * It is written from the bit-stream specification, not
 * translated from the reference's SIMD code.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it; the product library (libstenos_hip.so) never does.
 *
 * Parity pin: byte-identical to the unmodified reference built by oracle/Makefile into
 * oracle/_ref/libstenos_ref_det.so (checked by tests/test_oracle_vs_ref.py in the build
 * container) and to the committed golden fixtures under tests/golden/ that were generated
 * from that build (tests/golden/make_golden.py).
 *
 * Scope: levels 0 and 1 for bytesoftype > 1 (superblock codes 1 BLOCK, 6 COPY, and code 2 ZSTD
 * for superblocks shorter than 128 bytes through a dlopen'ed libzstd), the whole block decoder, and
 * DECODING of every superblock code (2-5 through libzstd + unshuffle / delta_inv / block decoder), so that
 * frames the reference produced at levels >= 2 can be checked.
 * Mini-LZ hash table starts empty for every block (see DESIGN.md "LZ table determinism").
 */
#ifndef STENOS_ORACLE_H
#define STENOS_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* error codes: identical values to stenos/stenos.h:75-84 */
#define SO_ERROR_UNDEFINED ((size_t)(-1))
#define SO_ERROR_SRC_OVERFLOW ((size_t)(-2))
#define SO_ERROR_ALLOC ((size_t)(-3))
#define SO_ERROR_INVALID_INPUT ((size_t)(-4))
#define SO_ERROR_INVALID_INSTRUCTION_SET ((size_t)(-5))
#define SO_ERROR_DST_OVERFLOW ((size_t)(-6))
#define SO_ERROR_INVALID_BYTESOFTYPE ((size_t)(-7))
#define SO_ERROR_ZSTD_INTERNAL ((size_t)(-8))
#define SO_ERROR_INVALID_PARAMETER ((size_t)(-9))
#define SO_LAST_ERROR_CODE ((size_t)(-100))

/* plane types (block_compress.h:52-55) and block markers (:58-60) */
enum { SO_PLANE_SAME = 0, SO_PLANE_RAW = 1, SO_PLANE_NORMAL = 2, SO_PLANE_NORMAL_RLE = 3 };
enum { SO_BLOCK_COPY = 252, SO_BLOCK_LZ = 253, SO_BLOCK_PARTIAL = 254 };

/* Result of analysing one 256-byte plane (find_pack_bits_params, block_compress.h:385-535). */
typedef struct so_plane_info {
	uint8_t type;        /* SO_PLANE_* before the size>256 -> RAW override */
	uint16_t size;       /* encoded size in bytes (1 for SAME) */
	uint8_t hdr[16];     /* row header nibbles */
	uint8_t mins[16];    /* per-row min (FOR) or min of deltas (delta rows); defined for all rows */
	uint8_t cost[16];    /* per-row payload cost incl. its min byte when one is emitted */
	uint16_t rle_mask[16];
	uint16_t drle_mask[16];
	uint16_t mins_mask;
} so_plane_info;

int so_has_error(size_t r);
size_t so_bound(size_t bytes);                                   /* stenos.h:37-42 */
size_t so_superblock_size(size_t bytesoftype, size_t bytes, int level); /* stenos.cpp:71-76,157-164 */

/* analyse one plane of 256 bytes; rle = 1 for full blocks, 0 for partial blocks */
void so_analyse_plane(const uint8_t plane[256], int rle, so_plane_info* info);

/* Encode one full block (256 elements of bytesoftype bytes).  `out` needs 256*T + T/2 + 32 bytes.
 * allow_lz: whether the mini-LZ attempt is permitted (capacity condition of block_compress.h:1214).
 * Returns the encoded size.  If info_out != NULL it receives bytesoftype plane infos. */
size_t so_encode_block(const uint8_t* block, size_t bytesoftype, uint8_t* out, int allow_lz);

/* block_compress (block_compress.h:1099-1302) at block_level 2 with the reference's capacity rules */
size_t so_block_compress(const uint8_t* src, size_t bytesoftype, size_t bytes, uint8_t* dst, size_t dst_size);
/* block_decompress (block_compress.h:1797-1879) */
size_t so_block_decompress(const uint8_t* src, size_t size, size_t bytesoftype, size_t bytes, uint8_t* dst);

/* stenos_compress / stenos_decompress (stenos.cpp:844-1017, 1052-1208), serial path.
 * so_decompress fixes the reference's exact-multiple bug (stenos.cpp:1115-1116, 1131): when
 * fix_exact_multiple == 0 it reproduces the reference and returns SO_ERROR_INVALID_INPUT. */
size_t so_compress(const void* src, size_t bytesoftype, size_t bytes, void* dst, size_t dst_size, int level);
size_t so_decompress(const void* src, size_t bytesoftype, size_t bytes, void* dst, size_t dst_size, int fix_exact_multiple);

/* Coverage counters over a compressed frame (decodes it): plane types, row headers, LZ blocks,
 * partial blocks, superblock codes.  Not part of the reference; used by tests to prove that the
 * parity cases exercise every branch of the bit stream. */
enum {
	SO_STAT_PLANE_TYPE = 0, /* +0..3 */
	SO_STAT_ROW_HDR = 4,    /* +0..15 */
	SO_STAT_LZ_BLOCKS = 20,
	SO_STAT_PARTIAL_BLOCKS = 21,
	SO_STAT_SB_CODE = 22,   /* +0..7 */
	SO_STAT_COPY_BLOCKS = 30, /* [252][raw] blocks (time-limited mode only, block_compress.h:1158-1176) */
	SO_STAT_COUNT = 32
};
size_t so_frame_stats(const void* src, size_t bytesoftype, size_t size, uint64_t counts[SO_STAT_COUNT]);

/* byte shuffle / unshuffle (shuffle-generic.h:33-125) and byte delta (delta.cpp:30-71, 230-268) */
void so_shuffle(size_t bytesoftype, size_t bytes, const uint8_t* src, uint8_t* dst);
void so_unshuffle(size_t bytesoftype, size_t bytes, const uint8_t* src, uint8_t* dst);
void so_delta(const uint8_t* src, uint8_t* dst, size_t bytes);
void so_delta_inv(const uint8_t* src, uint8_t* dst, size_t bytes);

#ifdef __cplusplus
}
#endif
#endif
