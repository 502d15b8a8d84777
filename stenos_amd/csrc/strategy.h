// strategy.h -- host side of the per-superblock strategy selection for levels >= 2 and bytesoftype 1
// (reference stenos/internal/stenos.cpp:451-604, 617-678): the LZ4 "dry" size estimator
// (lz4dry.cpp:658-848) that predicts what zstd would achieve, and the level mapping of zstd_wrapper.h:49-56.
// The block codec, the byte shuffle and the byte delta themselves run on the GPU.
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace strategy {

// bytes an LZ4-fast stream of src[0, n) would take (256-entry position table, acceleration as in LZ4)
size_t lz4_dry_size(const uint8_t* src, size_t n, int acceleration);

// guess_transposed_lz_ratio (stenos.cpp:376-401): `planes` holds, for each of the T byte planes of a shuffled
// superblock, the `step` bytes around its middle (raw or byte-delta'd), back to back
double transposed_ratio(const uint8_t* planes, size_t T, size_t step, int level);
// length of those middle pieces for a superblock of `bytes` bytes
size_t middle_step(size_t T, size_t bytes, int level);

} // namespace strategy
