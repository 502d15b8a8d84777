// wavevec.h -- one-wavefront (64 lanes) vocabulary used by the block codec.
//
// The codec (block_codec.h) is written once against this vocabulary:
//   * compiled by hipcc for gfx950, U32 is the lane's own 32-bit register and every operation is
//     a plain VALU instruction, a DPP/ds_bpermute cross-lane move, a ballot or an LDS access;
//   * compiled by g++ with -DWV_HOST_EMULATION, U32 is an array of 64 lanes executed in lockstep, so
//     the exact same codec source can be diffed against the CPU oracle without a GPU
//     (tests/emul/ -- test infrastructure only, never part of the shipped library).
//
// Rules the codec follows so that both builds mean the same thing:
//   * control flow only on wave-uniform scalars (plain uint32_t / bool obtained from ballot, readlane...);
//   * lane-varying choices through sel() and predicated LDS operations;
//   * LDS is a byte-addressed scratch private to the wave; accesses of one wave are ordered by wave_sync().
#pragma once
#include <stdint.h>

#ifdef WV_HOST_EMULATION
// (tests/emul/wavevec_host.h: the lockstep emulation of this vocabulary; tests/emul/Makefile puts it on the include path)
#include "wavevec_host.h"
#else
// ------------------------------------------------------------------------------------------------
// gfx950 device build
// ------------------------------------------------------------------------------------------------
#include <hip/hip_runtime.h>
#define WV_FN static __device__ __forceinline__
#define WV_MFN __device__ __forceinline__
#define WV_TABLE static __device__ const
#define WV_HD static __host__ __device__ __forceinline__
// a comment line in the generated ISA (tools/isa_regions.py counts the instructions between marks); no code
#define WV_MARK(name) asm volatile("; MARK " name)
// At the start of a block guarded by a wave-uniform condition: keeps the compiler from folding that condition into the
// conditions tested inside the block (it evaluates "a && b && c" of cheap scalar terms without branches: three scalar
// instructions per term on every pass, where the first test alone usually decides).  No code.
#define WV_NESTED() asm volatile("")
namespace wv {
constexpr int WAVE = 64;
typedef uint32_t U32;
typedef bool Pred;
typedef uint8_t* Lds; // points into __shared__ memory

WV_FN Pred pred_all(bool v) { return v; }
WV_FN U32 sel(Pred p, U32 a, U32 b) { return p ? a : b; }
WV_FN U32 lane_id()
{
	U32 l = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
	// Opaque to the optimiser on purpose: otherwise every value derived from the lane number (addresses, masks,
	// predicates) is computed once and kept alive across the whole block loop -- 97 instead of 61 vector registers
	// for the int32 encoder, i.e. 4 instead of 8 resident waves per SIMD.  Recomputing them costs next to nothing.
	asm volatile("" : "+v"(l));
	__builtin_assume(l < 64u); // lets the compiler drop predicates that are always true for a full wave
	return l;
}
// the lane number the optimiser may reason about (common subexpressions, hoisting): for code that is not short of registers
WV_FN U32 lane_id_plain()
{
	U32 l = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
	__builtin_assume(l < 64u);
	return l;
}
// number of set bits of mask below the lane's own: the lane's rank among the lanes of the mask
WV_FN U32 lane_rank(uint64_t mask) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u)); }
WV_FN U32 umin(U32 a, U32 b) { return a < b ? a : b; }
WV_FN U32 umax(U32 a, U32 b) { return a > b ? a : b; }
WV_FN U32 popc(U32 a) { return (U32)__builtin_popcount(a); }
typedef unsigned short wv_us2 __attribute__((ext_vector_type(2)));
WV_FN U32 pk_min_u16(U32 a, U32 b) { return __builtin_bit_cast(U32, __builtin_elementwise_min(__builtin_bit_cast(wv_us2, a), __builtin_bit_cast(wv_us2, b))); }
WV_FN U32 pk_max_u16(U32 a, U32 b) { return __builtin_bit_cast(U32, __builtin_elementwise_max(__builtin_bit_cast(wv_us2, a), __builtin_bit_cast(wv_us2, b))); }
WV_FN U32 bitlen(U32 a) { return a ? 32u - (U32)__builtin_clz(a) : 0u; }
WV_FN U32 mulhi(U32 a, U32 b) { return __umulhi(a, b); }
WV_FN U32 mul24(U32 a, U32 b) { return __umul24(a, b); } // low 24 bits of both operands, full rate
WV_FN U32 mad24(U32 a, U32 b, U32 c) { return __umul24(a, b) + c; } // (one v_mad_u32_u24)
WV_FN U32 dot4_u8(U32 a, uint32_t b, U32 c) { return __builtin_amdgcn_udot4(a, b, c, false); } // c + sum of the four byte products
WV_FN U32 msad_u8(U32 a, U32 ref, U32 c) { return __builtin_amdgcn_msad_u8(a, ref, c); } // c + sum of |a.byte - ref.byte| over the bytes with ref.byte != 0
WV_FN U32 bfe(U32 x, U32 off, U32 width) { return __builtin_amdgcn_ubfe(x, off, width); }
WV_FN U32 funnel_shr(U32 hi, U32 lo, U32 sh) { return __builtin_amdgcn_alignbit(hi, lo, sh); }
WV_FN U32 perm_bytes(U32 hi, U32 lo, uint32_t selw) { return __builtin_amdgcn_perm(hi, lo, selw); }
WV_FN U32 perm_bytes_v(U32 hi, U32 lo, U32 selw) { return __builtin_amdgcn_perm(hi, lo, selw); }
WV_FN uint64_t ballot(Pred p) { return __builtin_amdgcn_ballot_w64(p); }
// {hi, lo} = (64-bit) x << sh, sh in 0..32: one v_lshlrev_b64
WV_FN void shl64(U32 x, U32 sh, U32& lo, U32& hi)
{
	const uint64_t v = (uint64_t)x << sh;
	lo = (U32)v;
	hi = (U32)(v >> 32);
}
// bit r of lane 16k + q: p in lane 16k + r (the ballot of the lane's own group of 16 lanes): one 64-bit shift by a per-lane amount
WV_FN U32 row_ballot16(Pred p) { return (U32)(__builtin_amdgcn_ballot_w64(p) >> (lane_id_plain() & 48u)) & 0xFFFFu; }
WV_FN uint32_t readlane(U32 a, uint32_t lane) { return (uint32_t)__builtin_amdgcn_readlane((int)a, (int)lane); }
// 1 when the lane mask is not empty, else 0 -- as an integer in a scalar register (a C++ bool would be kept as a
// lane mask and turned into a number through vector registers)
WV_FN uint32_t mask_nonzero(uint64_t m)
{
	uint32_t r;
	asm("s_bcnt1_i32_b64 %0, %1\n\ts_min_u32 %0, %0, 1" : "=s"(r) : "s"(m) : "scc");
	return r;
}
// `bit` when the lane mask is not empty, else 0 (bit: a constant): a compare and a select in the scalar unit
template <uint32_t BIT>
WV_FN uint32_t mask_bit(uint64_t m)
{
	uint32_t r;
	asm("s_cmp_lg_u64 %1, 0\n\ts_cselect_b32 %0, %2, 0" : "=s"(r) : "s"(m), "i"(BIT) : "scc");
	return r;
}
WV_FN U32 shfl(U32 a, U32 src) { return (U32)__builtin_amdgcn_ds_bpermute((int)(src << 2), (int)a); }
// generic forms (LDS crossbar); the shapes the codec uses most have DPP forms below
WV_FN U32 shfl_up_any(U32 a, uint32_t n, uint32_t fill)
{
	U32 l = lane_id();
	U32 v = (U32)__builtin_amdgcn_ds_bpermute((int)((l - n) << 2), (int)a);
	return l >= n ? v : fill;
}
WV_FN U32 shfl_xor_any(U32 a, uint32_t m) { return (U32)__builtin_amdgcn_ds_bpermute((int)((lane_id() ^ m) << 2), (int)a); }
// lane i reads a[i - n]; lanes < n read `fill`.  n == 1 is one DPP wave_shr:1 (gfx9: ctrl 0x138).
WV_FN U32 shfl_up(U32 a, uint32_t n, uint32_t fill)
{
	if (__builtin_constant_p(n) && n == 1)
		return (U32)__builtin_amdgcn_update_dpp((int)fill, (int)a, 0x138, 0xf, 0xf, false);
	return shfl_up_any(a, n, fill);
}
// lane i reads a[i ^ m]; m = 1, 2 are DPP quad_perm [1,0,3,2] (0xB1) and [2,3,0,1] (0x4E)
WV_FN U32 shfl_xor(U32 a, uint32_t m)
{
	if (__builtin_constant_p(m) && m == 1)
		return (U32)__builtin_amdgcn_update_dpp(0, (int)a, 0xB1, 0xf, 0xf, false);
	if (__builtin_constant_p(m) && m == 2)
		return (U32)__builtin_amdgcn_update_dpp(0, (int)a, 0x4E, 0xf, 0xf, false);
	if (__builtin_constant_p(m) && m == 8)
		return (U32)__builtin_amdgcn_update_dpp(0, (int)a, 0x128, 0xf, 0xf, false); // row_ror:8
	return shfl_xor_any(a, m);
}

// DPP controls (gfx9): row_shr:n = 0x110+n, row_ror:n = 0x120+n
template <int CTRL>
WV_FN U32 dpp_zero(U32 a) // lanes without a source read 0
{
	return (U32)__builtin_amdgcn_update_dpp(0, (int)a, CTRL, 0xf, 0xf, true);
}
WV_FN U32 row_ror(U32 a, uint32_t n)
{
	switch (n) {
		case 1: return (U32)__builtin_amdgcn_update_dpp(0, (int)a, 0x121, 0xf, 0xf, false);
		case 2: return (U32)__builtin_amdgcn_update_dpp(0, (int)a, 0x122, 0xf, 0xf, false);
		case 4: return (U32)__builtin_amdgcn_update_dpp(0, (int)a, 0x124, 0xf, 0xf, false);
		default: return (U32)__builtin_amdgcn_update_dpp(0, (int)a, 0x128, 0xf, 0xf, false);
	}
}
WV_FN U32 row_shr(U32 a, uint32_t n, uint32_t fill)
{
	U32 v;
	switch (n) {
		case 1: v = dpp_zero<0x111>(a); break;
		case 2: v = dpp_zero<0x112>(a); break;
		case 4: v = dpp_zero<0x114>(a); break;
		default: v = dpp_zero<0x118>(a); break;
	}
	if (__builtin_constant_p(fill) && fill == 0)
		return v; // lanes without a source already read 0
	switch (n) { // lanes without a source keep the old value of the destination: the fill
		case 1: return (U32)__builtin_amdgcn_update_dpp((int)fill, (int)a, 0x111, 0xf, 0xf, false);
		case 2: return (U32)__builtin_amdgcn_update_dpp((int)fill, (int)a, 0x112, 0xf, 0xf, false);
		case 4: return (U32)__builtin_amdgcn_update_dpp((int)fill, (int)a, 0x114, 0xf, 0xf, false);
		default: return (U32)__builtin_amdgcn_update_dpp((int)fill, (int)a, 0x118, 0xf, 0xf, false);
	}
}
WV_FN U32 quad_last(U32 a) { return (U32)__builtin_amdgcn_update_dpp(0, (int)a, 0xFF, 0xf, 0xf, false); } // quad_perm:[3,3,3,3]
// lanes 16k .. 16k+15 receive a_k (four uniform values): four moves under narrowing execution masks, no compare
WV_FN U32 row_select4(uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3)
{
	U32 v;
	uint64_t save;
	asm volatile("s_mov_b64 %1, exec\n"
		     "v_mov_b32 %0, %2\n"
		     "s_and_b64 exec, %1, %6\n"
		     "v_mov_b32 %0, %3\n"
		     "s_and_b64 exec, %1, %7\n"
		     "v_mov_b32 %0, %4\n"
		     "s_and_b64 exec, %1, %8\n"
		     "v_mov_b32 %0, %5\n"
		     "s_mov_b64 exec, %1"
		     : "=&v"(v), "=&s"(save)
		     : "s"(a0), "s"(a1), "s"(a2), "s"(a3), "s"(0xFFFFFFFFFFFF0000ull), "s"(0xFFFFFFFF00000000ull), "s"(0xFFFF000000000000ull)
		     : "scc");
	return v;
}
// lanes 16k .. 16k+15 keep their own value of a_k (four vectors): the same four moves
WV_FN U32 row_select4v(U32 a0, U32 a1, U32 a2, U32 a3)
{
	U32 v;
	uint64_t save;
	asm volatile("s_mov_b64 %1, exec\n"
		     "v_mov_b32 %0, %2\n"
		     "s_and_b64 exec, %1, %6\n"
		     "v_mov_b32 %0, %3\n"
		     "s_and_b64 exec, %1, %7\n"
		     "v_mov_b32 %0, %4\n"
		     "s_and_b64 exec, %1, %8\n"
		     "v_mov_b32 %0, %5\n"
		     "s_mov_b64 exec, %1"
		     : "=&v"(v), "=&s"(save)
		     : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "s"(0xFFFFFFFFFFFF0000ull), "s"(0xFFFFFFFF00000000ull), "s"(0xFFFF000000000000ull)
		     : "scc");
	return v;
}
// orders this wave's LDS accesses (program order is enough for one wave on the hardware; this
// stops the compiler from moving accesses across the point)
WV_FN void wave_sync()
{
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

#ifdef STENOS_WIDE
// kernels_wide.hip: the wave's scratch lies in HBM.  Its OR-ed images are built by atomics, which execute in the L2, so
// the reads go there too instead of trusting a line of the vector L1.
WV_FN uint32_t wide_ld32(const uint8_t* p) { return __hip_atomic_load((const uint32_t*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
WV_FN U32 lds_ld8(Lds m, U32 a) { return __hip_atomic_load(m + a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
WV_FN U32 lds_ld32(Lds m, U32 a) { return wide_ld32(m + (a & ~3u)); }
WV_FN void lds_ld64(Lds m, U32 a, U32& lo, U32& hi)
{
	lo = wide_ld32(m + (a & ~3u));
	hi = wide_ld32(m + (a & ~3u) + 4);
}
#else
WV_FN U32 lds_ld8(Lds m, U32 a) { return m[a]; }
// (the address is aligned down after the base is added: the scratch starts at a multiple of 16, and that way the addition
// joins the scalar arithmetic of a wave-uniform part of `a` instead of costing a vector instruction behind the v_and)
WV_FN U32 lds_ld32(Lds m, U32 a) { return *(const uint32_t*)__builtin_align_down(m + a, 4); }
WV_FN void lds_ld64(Lds m, U32 a, U32& lo, U32& hi)
{
	const uint32_t* p = (const uint32_t*)__builtin_align_down(m + a, 4);
	lo = p[0];
	hi = p[1];
}
#endif
// Predicated memory accesses without a branch.  `if (p) store` is a divergent branch, and with one divergent branch
// anywhere inside a loop the compiler structurizes every wave-uniform `if` of that loop as well -- whatever
// -structurizecfg-skip-uniform-regions says, which spares regions made of uniform branches only (decode_kernels.hip is
// compiled with it) -- : conditions kept as lane masks, s_cselect_b64 / s_and_b64 / s_cbranch_vccnz instead of s_cmp /
// s_cbranch_scc, flags carried between flow blocks; a third of the decoder's scalar instructions were of that kind
// (tools/divergent_branches.sh lists a kernel's divergent branches).  So the predicate goes into the execution mask for the
// one instruction, in a single asm statement the compiler cannot interleave with anything.  LDS executes a wave's
// accesses in order, so stores the compiler does not count need no wait; global accesses issued here are waited for
// inside the statement (they are the rare ones: the hot loops clamp their addresses instead, see superblock_codec.h).
// The low 32 bits of a flat address of LDS are the LDS offset.
WV_FN uint32_t lds_offset(Lds m, U32 a) { return (uint32_t)(uintptr_t)m + a; }
typedef uint32_t wv_u4 __attribute__((ext_vector_type(4)));
typedef uint32_t wv_u2 __attribute__((ext_vector_type(2)));
#define WV_MASKED(instr, mask, ...)                                                                                    \
	do {                                                                                                              \
		uint64_t wv_save;                                                                                               \
		asm volatile("s_mov_b64 %0, exec\n\ts_and_b64 exec, exec, %1\n\t" instr "\n\ts_mov_b64 exec, %0"                 \
			     : "=&s"(wv_save)                                                                                       \
			     : "s"(mask), __VA_ARGS__                                                                               \
			     : "memory", "scc");                                                                                    \
	} while (0)
#if defined(STENOS_WIDE) || defined(WV_PREDICATE_BRANCHES)
// (STENOS_WIDE: the scratch is global memory there; WV_PREDICATE_BRANCHES: the encoders, which are faster with branches,
// csrc/Makefile)
WV_FN void lds_st32(Lds m, U32 a, U32 v, Pred p)
{
	if (p) *(uint32_t*)(m + (a & ~3u)) = v;
}
WV_FN void lds_st8(Lds m, U32 a, U32 v, Pred p)
{
	if (p) m[a] = (uint8_t)v;
}
WV_FN U32 lds_cas32(Lds m, U32 a, U32 expect, U32 v, Pred p)
{
	uint32_t e = expect;
	if (p) __hip_atomic_compare_exchange_strong((uint32_t*)(m + (a & ~3u)), &e, v, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
	return e; // the value found (== expect when the swap happened)
}
WV_FN U32 lds_add_rtn32(Lds m, U32 a, U32 v, Pred p)
{
	return p ? __hip_atomic_fetch_add((uint32_t*)(m + (a & ~3u)), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) : 0u;
}
#else
WV_FN void lds_st32(Lds m, U32 a, U32 v, Pred p)
{
	if (__builtin_constant_p(p) && p)
		*(uint32_t*)(m + (a & ~3u)) = v;
	else
		WV_MASKED("ds_write_b32 %2, %3", ballot(p), "v"(lds_offset(m, a & ~3u)), "v"(v));
}
WV_FN void lds_st8(Lds m, U32 a, U32 v, Pred p)
{
	if (__builtin_constant_p(p) && p)
		m[a] = (uint8_t)v;
	else
		WV_MASKED("ds_write_b8 %2, %3", ballot(p), "v"(lds_offset(m, a)), "v"(v));
}
WV_FN U32 lds_cas32(Lds m, U32 a, U32 expect, U32 v, Pred p)
{
	// (ds_cmpst_rtn_b32 vdst, addr, compare, new value; lanes that stay out get `expect` back, as if the swap had happened)
	uint32_t e = expect;
	uint64_t save;
	asm volatile("s_mov_b64 %1, exec\n\ts_and_b64 exec, exec, %2\n\tds_cmpst_rtn_b32 %0, %3, %4, %5\n\ts_waitcnt lgkmcnt(0)\n\ts_mov_b64 exec, %1"
		     : "+v"(e), "=&s"(save)
		     : "s"(ballot(p)), "v"(lds_offset(m, a & ~3u)), "v"(expect), "v"(v)
		     : "memory", "scc");
	return e; // the value found (== expect when the swap happened)
}
WV_FN U32 lds_add_rtn32(Lds m, U32 a, U32 v, Pred p)
{
	uint32_t r = 0;
	uint64_t save;
	asm volatile("s_mov_b64 %1, exec\n\ts_and_b64 exec, exec, %2\n\tds_add_rtn_u32 %0, %3, %4\n\ts_waitcnt lgkmcnt(0)\n\ts_mov_b64 exec, %1"
		     : "+v"(r), "=&s"(save)
		     : "s"(ballot(p)), "v"(lds_offset(m, a & ~3u)), "v"(v)
		     : "memory", "scc");
	return r;
}
#endif
// The walk over sixteen run-length rows that follow each other (a row: mask16, then one byte per clear bit of the mask --
// 18 - popcount(mask) bytes).  In: `at` = the offset of the first row's mask, the same in all lanes.  Out: in the lanes of row
// r (lane >> 2) the offset of that row's mask and the mask.  The chain of offsets is serial by nature; what it need not cost is
// a compare and two selects per row to hand each row its step: every step drops the lowest row's four lanes from the exec mask
// (one scalar shift), so the lanes that stay behind keep what their row's step found.  Four vector instructions a row.
// The wave's exec mask must be full on entry.
// (two byte reads: the LDS does read 16 bits at an odd address, tools/ubench_unaligned.hip, but a read that straddles a dword
// takes a slow path -- 170 cycles alone, 1200 with the device busy, tools/ubench_lds_bytes.hip: the whole decode of run-length
// data was 1.8 x slower with ds_read_u16 here)
#define WV_RLE_WALK_LOAD "ds_read_u8 %1, %0\n\tds_read_u8 %3, %0 offset:1\n\ts_waitcnt lgkmcnt(0)\n\tv_lshl_or_b32 %1, %3, 8, %1\n\t"
// (~mask has its upper half set: 16 + the clear bits of the mask, so the next row is at + popcount(~mask) - 14)
#define WV_RLE_WALK_STEP "s_lshl_b64 exec, exec, 4\n\tv_not_b32 %3, %1\n\tv_bcnt_u32_b32 %3, %3, %0\n\tv_add_u32 %0, -14, %3\n\t" WV_RLE_WALK_LOAD
#define WV_RLE_WALK_STEP5 WV_RLE_WALK_STEP WV_RLE_WALK_STEP WV_RLE_WALK_STEP WV_RLE_WALK_STEP WV_RLE_WALK_STEP
#ifdef STENOS_WIDE
WV_FN void lds_rle_walk16(Lds m, U32& at, U32& mask) // (the scratch is global memory there: the same walk in plain terms)
{
	const U32 row = lane_id_plain() >> 2;
	U32 a = at;
	for (uint32_t r = 0; r < 16; ++r) {
		const U32 mk = lds_ld8(m, a) | (lds_ld8(m, a + 1u) << 8);
		const Pred here = row == U32(r);
		mask = sel(here, mk, mask);
		at = sel(here, a, at);
		a = a + (U32(18u) - popc(mk));
	}
}
#else
WV_FN void lds_rle_walk16(Lds m, U32& at, U32& mask)
{
	uint32_t a = lds_offset(m, at), mk, t;
	uint64_t save;
	asm volatile("s_mov_b64 %2, exec\n\t" WV_RLE_WALK_LOAD WV_RLE_WALK_STEP5 WV_RLE_WALK_STEP5 WV_RLE_WALK_STEP5 "s_mov_b64 exec, %2"
		     : "+v"(a), "=&v"(mk), "=&s"(save), "=&v"(t)
		     :
		     : "memory", "scc");
	at = a - lds_offset(m, U32(0u));
	mask = mk;
}
#endif
// One run-length row among rows of other kinds (their sizes are known without reading them): `base` is where the row starts
// if the run-length rows in front of it had no bytes, `e` what those rows do take -- the same in all lanes from this row on --,
// `first` the row's first lane.  The row's lanes leave with its mask in `mask` and `e` as it is; the lanes of the rows behind
// it with `e` grown by the row's 18 - popcount(mask) bytes.  Six vector instructions a row, no select, no compare.
#ifdef STENOS_WIDE
WV_FN void lds_rle_walk_row(Lds m, uint32_t base, uint32_t first, U32& e, U32& mask) // (global scratch there: plain terms)
{
	const U32 lane = lane_id_plain();
	const U32 at = U32(base) + U32(readlane(e, first));
	const U32 mk = lds_ld8(m, at) | (lds_ld8(m, at + 1u) << 8);
	mask = sel(lane >= U32(first), mk, mask);
	e = sel(lane >= U32(first + 4u), e + (U32(18u) - popc(mk)), e);
}
#else
WV_FN void lds_rle_walk_row(Lds m, uint32_t base, uint32_t first, U32& e, U32& mask)
{
	uint32_t t, t2;
	uint64_t save;
	asm volatile("s_mov_b64 %4, exec\n\ts_lshl_b64 exec, -1, %6\n\tv_add_u32 %2, %5, %0\n\tds_read_u8 %1, %2\n\tds_read_u8 %3, %2 offset:1\n\t"
		     "s_waitcnt lgkmcnt(0)\n\tv_lshl_or_b32 %1, %3, 8, %1\n\ts_lshl_b64 exec, exec, 4\n\tv_not_b32 %2, %1\n\t"
		     "v_bcnt_u32_b32 %2, %2, %0\n\tv_add_u32 %0, -14, %2\n\ts_mov_b64 exec, %4"
		     : "+v"(e), "+v"(mask), "=&v"(t), "=&v"(t2), "=&s"(save)
		     : "s"(base + lds_offset(m, U32(0u))), "s"(first)
		     : "memory", "scc");
}
#endif
// The same kind of walk over the 32 groups of a mini-LZ block with every match taken for one byte: a group is its flags byte
// and eight items, a match (flag set) one byte, a literal B bytes -- 1 + 8 B - (B - 1) * popcount(flags) bytes.  In: `at` = the
// offset of the first group's flags, the same in all lanes.  Out: in lane g < 32 the offset of group g and its flags (lanes
// 32-63: what lane 31 holds, one step on).  Three vector instructions a group and nothing on the scalar unit but the shift of
// the exec mask; offsets past the data just read what lies there (the caller checks them).  Full exec mask on entry.
#ifdef STENOS_WIDE
WV_FN void lds_lz_walk32(Lds m, U32& at, U32& flags, uint32_t B) // (the scratch is global memory there; not used, kept compilable)
{
	const U32 lane = lane_id_plain();
	U32 a = at;
	for (uint32_t g = 0; g < 32; ++g) {
		const U32 fl = lds_ld8(m, a);
		const Pred here = lane >= U32(g);
		flags = sel(here, fl, flags);
		at = sel(here, a, at);
		a = a + (U32(1u + 8u * B) - popc(fl) * (B - 1u));
	}
}
#else
#define WV_LZ_WALK_STEP "s_lshl_b64 exec, exec, 1\n\tv_bcnt_u32_b32 %3, %1, 0\n\tv_mad_i32_i24 %0, %3, %4, %0\n\tv_add_u32 %0, %5, %0\n\tds_read_u8 %1, %0\n\ts_waitcnt lgkmcnt(0)\n\t"
#define WV_LZ_WALK_STEP4 WV_LZ_WALK_STEP WV_LZ_WALK_STEP WV_LZ_WALK_STEP WV_LZ_WALK_STEP
#define WV_LZ_WALK_STEP16 WV_LZ_WALK_STEP4 WV_LZ_WALK_STEP4 WV_LZ_WALK_STEP4 WV_LZ_WALK_STEP4
WV_FN void lds_lz_walk32(Lds m, U32& at, U32& flags, uint32_t B)
{
	uint32_t a = lds_offset(m, at), fl, t;
	uint64_t save;
	asm volatile("s_mov_b64 %2, exec\n\tds_read_u8 %1, %0\n\ts_waitcnt lgkmcnt(0)\n\t" WV_LZ_WALK_STEP16 WV_LZ_WALK_STEP4 WV_LZ_WALK_STEP4 WV_LZ_WALK_STEP4
		     WV_LZ_WALK_STEP WV_LZ_WALK_STEP WV_LZ_WALK_STEP "s_mov_b64 exec, %2"
		     : "+v"(a), "=&v"(fl), "=&s"(save), "=&v"(t)
		     : "s"(0u - (B - 1u)), "s"(1u + 8u * B)
		     : "memory", "scc");
	at = a - lds_offset(m, U32(0u));
	flags = fl;
}
#endif
// OR-ing 0 is a no-op, so a predicated OR needs no branch: inactive lanes OR 0 into a dword of their own
// at the start of the buffer (one shared address would serialise the whole wave in the LDS atomic unit)
WV_FN U32 lds_or_rtn32(Lds m, U32 a, U32 v) { return __hip_atomic_fetch_or((uint32_t*)(m + a), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }
WV_FN void lds_or32_all(Lds m, U32 a, U32 v) { __hip_atomic_fetch_or((uint32_t*)(m + a), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }
WV_FN void lds_or32(Lds m, U32 a, U32 v, Pred p)
{
#ifdef WV_PREDICATE_BRANCHES
	__hip_atomic_fetch_or((uint32_t*)(m + (p ? (a & ~3u) : lane_id() * 4u)), p ? v : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
#else
	const U32 own = lane_id() * 4u; // (outside the selection: lane_id() hides an asm statement, which would turn it into a branch)
	__hip_atomic_fetch_or((uint32_t*)(m + (p ? (a & ~3u) : own)), p ? v : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
#endif
}

// ---- global memory ----
struct U128 {
	U32 x, y, z, w;
};
// predicated global loads: the lanes that stay out read nothing and get 0
#define WV_MASKED_LOAD(instr, dst, mask, ptr)                                                                          \
	do {                                                                                                              \
		uint64_t wv_save;                                                                                               \
		asm volatile("s_mov_b64 %1, exec\n\ts_and_b64 exec, exec, %2\n\t" instr " %0, %3, off\n\ts_waitcnt vmcnt(0)\n\ts_mov_b64 exec, %1" \
			     : "+v"(dst), "=&s"(wv_save)                                                                            \
			     : "s"(mask), "v"(ptr)                                                                                  \
			     : "memory", "scc");                                                                                    \
	} while (0)
#define WV_MASKED_STORE(instr, mask, ptr, data)                                                                        \
	do {                                                                                                              \
		uint64_t wv_save;                                                                                               \
		asm volatile("s_mov_b64 %0, exec\n\ts_and_b64 exec, exec, %1\n\t" instr " %2, %3, off\n\ts_waitcnt vmcnt(0)\n\ts_mov_b64 exec, %0" \
			     : "=&s"(wv_save)                                                                                       \
			     : "s"(mask), "v"(ptr), "v"(data)                                                                       \
			     : "memory", "scc");                                                                                    \
	} while (0)
#ifdef WV_PREDICATE_BRANCHES
WV_FN U32 gld8(const uint8_t* g, U32 off, Pred p) { return p ? (U32)g[off] : 0u; }
WV_FN U32 gld32(const uint8_t* g, U32 off, Pred p) { return p ? *(const uint32_t*)(g + off) : 0u; }
#else
WV_FN U32 gld8(const uint8_t* g, U32 off, Pred p)
{
	if (__builtin_constant_p(p) && p)
		return (U32)g[off];
	uint32_t r = 0;
	WV_MASKED_LOAD("global_load_ubyte", r, ballot(p), g + off);
	return r;
}
WV_FN U32 gld32(const uint8_t* g, U32 off, Pred p)
{
	if (__builtin_constant_p(p) && p)
		return *(const uint32_t*)(g + off);
	uint32_t r = 0;
	WV_MASKED_LOAD("global_load_dword", r, ballot(p), g + off);
	return r;
}
#endif
WV_FN void gld64(const uint8_t* g, U32 off, U32& lo, U32& hi)
{
	uint2 v = *(const uint2*)(g + off);
	lo = v.x;
	hi = v.y;
}
#ifdef WV_PREDICATE_BRANCHES
WV_FN U128 gld128(const uint8_t* g, U32 off, Pred p)
{
	U128 r = { 0u, 0u, 0u, 0u };
	if (p) {
		uint4 v = *(const uint4*)(g + off);
		r.x = v.x; r.y = v.y; r.z = v.z; r.w = v.w;
	}
	return r;
}
// 16 bytes from any byte address (gfx9 and later serve unaligned global accesses in hardware)
WV_FN U128 gld128_unaligned(const uint8_t* g, U32 off, Pred p)
{
	typedef uint4 __attribute__((aligned(1))) uint4_u;
	U128 r = { 0u, 0u, 0u, 0u };
	if (p) {
		uint4 v = *(const uint4_u*)(g + off);
		r.x = v.x; r.y = v.y; r.z = v.z; r.w = v.w;
	}
	return r;
}
#else
// 16 bytes from any byte address (gfx9 and later serve unaligned global accesses in hardware)
WV_FN U128 gld128_unaligned(const uint8_t* g, U32 off, Pred p)
{
	typedef uint4 __attribute__((aligned(1))) uint4_u;
	if (__builtin_constant_p(p) && p) {
		uint4 v = *(const uint4_u*)(g + off);
		U128 r = { v.x, v.y, v.z, v.w };
		return r;
	}
	wv_u4 d = { 0u, 0u, 0u, 0u };
	WV_MASKED_LOAD("global_load_dwordx4", d, ballot(p), g + off);
	U128 r = { d.x, d.y, d.z, d.w };
	return r;
}
WV_FN U128 gld128(const uint8_t* g, U32 off, Pred p)
{
	if (__builtin_constant_p(p) && p) {
		uint4 v = *(const uint4*)(g + off);
		U128 r = { v.x, v.y, v.z, v.w };
		return r;
	}
	return gld128_unaligned(g, off, p);
}
#endif
WV_FN void gld64_unaligned(const uint8_t* g, U32 off, U32& lo, U32& hi)
{
	typedef uint2 __attribute__((aligned(1))) uint2_u;
	uint2 v = *(const uint2_u*)(g + off);
	lo = v.x;
	hi = v.y;
}
#ifdef WV_PREDICATE_BRANCHES
WV_FN void gst8(uint8_t* g, U32 off, U32 v, Pred p)
{
	if (p) g[off] = (uint8_t)v;
}
WV_FN void gst32(uint8_t* g, U32 off, U32 v, Pred p)
{
	if (p) *(uint32_t*)(g + off) = v;
}
WV_FN void gst64(uint8_t* g, U32 off, U32 lo, U32 hi, Pred p)
{
	if (p) *(uint2*)(g + off) = make_uint2(lo, hi);
}
WV_FN void gst128(uint8_t* g, U32 off, const U128& v, Pred p)
{
	if (p) *(uint4*)(g + off) = make_uint4(v.x, v.y, v.z, v.w);
}
WV_FN void gst128_unaligned(uint8_t* g, U32 off, const U128& v, Pred p)
{
	typedef uint4 __attribute__((aligned(1))) uint4_u;
	if (p) *(uint4_u*)(g + off) = make_uint4(v.x, v.y, v.z, v.w);
}
WV_FN void gst64_unaligned(uint8_t* g, U32 off, U32 lo, U32 hi, Pred p)
{
	typedef uint2 __attribute__((aligned(1))) uint2_u;
	if (p) *(uint2_u*)(g + off) = make_uint2(lo, hi);
}
WV_FN void gst128_streamed(uint8_t* g, U32 off, const U128& v, Pred p) { gst128(g, off, v, p); }
WV_FN void gst8_streamed(uint8_t* g, U32 off, U32 v, Pred p) { gst8(g, off, v, p); }
#else
WV_FN void gst8(uint8_t* g, U32 off, U32 v, Pred p)
{
	if (__builtin_constant_p(p) && p)
		g[off] = (uint8_t)v;
	else
		WV_MASKED_STORE("global_store_byte", ballot(p), g + off, v);
}
WV_FN void gst32(uint8_t* g, U32 off, U32 v, Pred p)
{
	if (__builtin_constant_p(p) && p)
		*(uint32_t*)(g + off) = v;
	else
		WV_MASKED_STORE("global_store_dword", ballot(p), g + off, v);
}
WV_FN void gst64_unaligned(uint8_t* g, U32 off, U32 lo, U32 hi, Pred p)
{
	typedef uint2 __attribute__((aligned(1))) uint2_u;
	if (__builtin_constant_p(p) && p)
		*(uint2_u*)(g + off) = make_uint2(lo, hi);
	else {
		wv_u2 d = { lo, hi };
		WV_MASKED_STORE("global_store_dwordx2", ballot(p), g + off, d);
	}
}
WV_FN void gst64(uint8_t* g, U32 off, U32 lo, U32 hi, Pred p)
{
	if (__builtin_constant_p(p) && p)
		*(uint2*)(g + off) = make_uint2(lo, hi);
	else
		gst64_unaligned(g, off, lo, hi, p);
}
WV_FN void gst128_unaligned(uint8_t* g, U32 off, const U128& v, Pred p)
{
	typedef uint4 __attribute__((aligned(1))) uint4_u;
	if (__builtin_constant_p(p) && p)
		*(uint4_u*)(g + off) = make_uint4(v.x, v.y, v.z, v.w);
	else {
		wv_u4 d = { v.x, v.y, v.z, v.w };
		WV_MASKED_STORE("global_store_dwordx4", ballot(p), g + off, d);
	}
}
// The predicated 16-byte store of the hot loops: as above, but nothing waits for it.  The compiler does not count it; its
// own waits only become more cautious by that (memory operations of a wave complete in order), and the bytes are read
// again by the same wave or after the kernel only (superblock_codec.h: stream_append, copy_g2g_wide).
WV_FN uint8_t* uniform_pointer(uint8_t* g) // (a no-op where the compiler knows that all lanes hold the same pointer)
{
	const uint64_t a = (uint64_t)g;
	return (uint8_t*)((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a) | ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32)) << 32));
}
WV_FN void gst128_streamed(uint8_t* g, U32 off, const U128& v, Pred p)
{
	// (g is the same for all lanes: a scalar base and a 32-bit lane offset, as the compiler addresses its own stores)
	wv_u4 d = { v.x, v.y, v.z, v.w };
	uint64_t save;
	// (two wait states behind the store: on gfx940 and later a store of more than 8 bytes still reads its data registers that
	// long, and the compiler's hazard recognizer does not look into an asm statement -- block_codec.h found this the hard way)
	// (s_nop 2: a vector-memory instruction may read a scalar register five wait states after a VALU instruction -- the
	// v_readfirstlane of uniform_pointer -- wrote it, at the earliest; the two scalar instructions in front count as two)
	asm volatile("s_mov_b64 %0, exec\n\ts_and_b64 exec, exec, %1\n\ts_nop 2\n\tglobal_store_dwordx4 %2, %3, %4\n\ts_nop 0\n\ts_mov_b64 exec, %0"
		     : "=&s"(save)
		     : "s"(ballot(p)), "v"(off), "v"(d), "s"(uniform_pointer(g))
		     : "memory", "scc");
}
WV_FN void gst8_streamed(uint8_t* g, U32 off, U32 v, Pred p)
{
	uint64_t save;
	asm volatile("s_mov_b64 %0, exec\n\ts_and_b64 exec, exec, %1\n\ts_nop 2\n\tglobal_store_byte %2, %3, %4\n\ts_mov_b64 exec, %0"
		     : "=&s"(save)
		     : "s"(ballot(p)), "v"(off), "v"(v), "s"(uniform_pointer(g))
		     : "memory", "scc");
}
WV_FN void gst128(uint8_t* g, U32 off, const U128& v, Pred p)
{
	if (__builtin_constant_p(p) && p)
		*(uint4*)(g + off) = make_uint4(v.x, v.y, v.z, v.w);
	else
		gst128_unaligned(g, off, v, p);
}
#endif
// Write-through stores (all lanes): the bytes go to memory, past this XCD's L2 -- the per-XCD L2s are not coherent with
// each other -- so that a workgroup on another XCD that overwrites them later wins whatever the order of the write-backs
// (kernels.hip, speculative copy).  The compiler does not count them: gst_through_wait() waits for them.
WV_FN uint8_t* wave_base(uint8_t* g) // (all lanes hold the same pointer: say so)
{
	const uint64_t a = (uint64_t)g;
	return (uint8_t*)((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a) | ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32)) << 32));
}
WV_FN void gst128_through(uint8_t* g, U32 off, const U128& v)
{
	wv_u4 d = { v.x, v.y, v.z, v.w };
	// (s_nop 1: a store of more than 8 bytes still reads its data registers for two more cycles on gfx940 and later, and the compiler's hazard
	// recognizer does not look into an asm statement: without it the instruction behind may overwrite the first of them)
	// (s_nop 4 in front: the base is read from scalar registers that a VALU instruction -- the v_readfirstlane of wave_base --
	// may have written just before; a vector-memory instruction needs five wait states behind that, and the compiler's hazard
	// recognizer does not look into an asm statement either way.  Found as a memory fault at address 0 on the box.)
	asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2 sc0 sc1\n\ts_nop 1" : : "v"(off), "v"(d), "s"(wave_base(g)) : "memory");
}
WV_FN void gst64_through(uint8_t* g, U32 off, U32 lo, U32 hi)
{
	typedef uint32_t wv_pair __attribute__((ext_vector_type(2)));
	wv_pair d = { lo, hi };
	asm volatile("s_nop 4\n\tglobal_store_dwordx2 %0, %1, %2 sc0 sc1\n\ts_nop 1" : : "v"(off), "v"(d), "s"(wave_base(g)) : "memory");
}
WV_FN void gst_through_wait() { asm volatile("s_waitcnt vmcnt(0)" : : : "memory"); }
// wave-uniform scalar accesses to global memory (stores by one lane)
WV_FN uint32_t gload_uniform(const uint32_t* p) { return *(const volatile uint32_t*)p; }
WV_FN uint32_t gload_uniform8(const uint8_t* p) { return *(const volatile uint8_t*)p; }
WV_FN uint64_t gload_uniform64(const uint64_t* p) { return *(const volatile uint64_t*)p; }
#ifdef WV_PREDICATE_BRANCHES
WV_FN void status_or(uint32_t* status, uint32_t bits)
{
	if (lane_id() == 0) atomicOr(status, bits);
}
// wave-uniform scalar accesses to global memory (stores by one lane)
WV_FN void gstore_uniform(uint32_t* p, uint32_t v)
{
	if (lane_id() == 0) *p = v;
}
WV_FN void gstore_uniform8(uint8_t* p, uint32_t v)
{
	if (lane_id() == 0) *p = (uint8_t)v;
}
WV_FN void gstore_uniform64(uint64_t* p, uint64_t v)
{
	if (lane_id() == 0) *p = v;
}
WV_FN void gmin32(uint32_t* p, uint32_t v)
{
	if (lane_id() == 0) atomicMin(p, v);
}
#else
// What one lane of a wavefront does for all, without a branch: the execution mask is 1 for the one instruction.
// status |= bits, by the first lane
WV_FN void status_or(uint32_t* status, uint32_t bits) { WV_MASKED_STORE("global_atomic_or", 1ull, (uint8_t*)status, bits); }
WV_FN void gstore_uniform(uint32_t* p, uint32_t v) { WV_MASKED_STORE("global_store_dword", 1ull, (uint8_t*)p, v); }
WV_FN void gstore_uniform8(uint8_t* p, uint32_t v) { WV_MASKED_STORE("global_store_byte", 1ull, p, v); }
WV_FN void gstore_uniform64(uint64_t* p, uint64_t v)
{
	wv_u2 d = { (uint32_t)v, (uint32_t)(v >> 32) };
	WV_MASKED_STORE("global_store_dwordx2", 1ull, (uint8_t*)p, d);
}
WV_FN void gmin32(uint32_t* p, uint32_t v) { WV_MASKED_STORE("global_atomic_umin", 1ull, (uint8_t*)p, v); }
#endif
WV_FN U128 lds_ld128(Lds m, U32 a)
{
#ifdef STENOS_WIDE
	U128 r = { wide_ld32(m + a), wide_ld32(m + a + 4), wide_ld32(m + a + 8), wide_ld32(m + a + 12) };
#else
	uint4 v = *(const uint4*)(m + a);
	U128 r = { v.x, v.y, v.z, v.w };
#endif
	return r;
}
WV_FN void lds_st128(Lds m, U32 a, const U128& v, Pred p)
{
#if defined(STENOS_WIDE) || defined(WV_PREDICATE_BRANCHES)
	if (p) *(uint4*)(m + a) = make_uint4(v.x, v.y, v.z, v.w);
	return;
#endif
	if (__builtin_constant_p(p) && p)
		*(uint4*)(m + a) = make_uint4(v.x, v.y, v.z, v.w);
	else {
		wv_u4 d = { v.x, v.y, v.z, v.w };
		WV_MASKED("ds_write_b128 %2, %3", ballot(p), "v"(lds_offset(m, a)), "v"(d));
	}
}
} // namespace wv
#endif

// ------------------------------------------------------------------------------------------------
// helpers written in the vocabulary above (identical for both builds)
// ------------------------------------------------------------------------------------------------
namespace wv {

WV_FN bool any(const Pred& p) { return ballot(p) != 0; }
// A value the first lane holds, as a wave-uniform scalar: needed for scalars that were updated inside lanes_below(), which
// on the device only the participating lanes have seen.
WV_FN uint32_t first_lane_value(uint32_t x)
{
#ifdef WV_HOST_EMULATION
	return x;
#else
	return (uint32_t)__builtin_amdgcn_readfirstlane((int)x);
#endif
}
// Run f(p) for the first n lanes only: on the device one divergent region (p is all-true inside it), on the host a predicate.
template <class F>
WV_FN void lanes_below(uint32_t n, F f)
{
#ifdef WV_HOST_EMULATION
	f(lane_id() < U32(n));
#elif defined(WV_PREDICATE_BRANCHES)
	if (lane_id_plain() < n)
		f(true);
#else
	f(lane_id_plain() < n); // (no divergent region: see the predicated accesses above)
#endif
}
// byte 0 of x in all four bytes (one v_perm_b32)
WV_FN U32 splat_byte0(const U32& x) { return perm_bytes(x, x, 0u); }

// unaligned little-endian 32-bit read from LDS (two aligned reads + funnel).  gfx950 does serve a ds_read_b32 of any byte
// address and the compiler emits it for a load of alignment 1, but it is far slower than this: the int16 decoder went
// from 3.57 to 5.8 ms per 8 GiB with it (measured, round 4).
WV_FN U32 lds_ld32_unaligned(Lds m, const U32& a)
{
	U32 lo, hi;
	lds_ld64(m, a, lo, hi); // reads the aligned dword containing a and the next one
	return funnel_shr(hi, lo, (a & 3u) << 3); // (hi:lo) >> 0, 8, 16 or 24: one v_alignbit_b32, no case for the aligned address
}

// the lane's dword of a run of dwords that starts at the wave-uniform byte address a: lane l reads a + 4 * l.  The
// misalignment is the same in all lanes, so it stays in scalar registers: one vector addition and the funnel shift.
WV_FN U32 lds_ld32_run(Lds m, uint32_t a, const U32& lane4)
{
	U32 lo, hi;
	lds_ld64(m, U32(a & ~3u) + lane4, lo, hi);
	return funnel_shr(hi, lo, U32((a & 3u) << 3));
}
// 32 bits of the LDS bit stream (LSB first) from bit position bitpos on: the two aligned dwords around it, one funnel shift
WV_FN U32 lds_ld32_bits(Lds m, const U32& bitpos)
{
	U32 lo, hi;
	lds_ld64(m, bitpos >> 3, lo, hi);
	return funnel_shr(hi, lo, bitpos); // (v_alignbit_b32 takes the low five bits: 8 * (byte address & 3) + bitpos & 7)
}

// OR `nbits` (<= 32) bits of value into the LDS bit stream at bit position bitpos (LSB first);
// the buffer must have been zeroed and have 4 bytes of slack after the last piece
WV_FN void lds_put_bits(Lds m, const U32& bitpos, const U32& value, const Pred& p)
{
	U32 addr = (bitpos >> 5) << 2;
	U32 sh = bitpos & 31u;
	lds_or32(m, addr, value << sh, p);
	U32 hi = sel(sh == U32(0u), U32(0u), value >> (U32(32u) - sh));
	lds_or32(m, addr + 4u, hi, p); // hi == 0 ORs nothing; the image has 4 bytes of slack
}

// OR the 8 bytes lo, hi (little endian) into the zeroed LDS image at byte position pos, by every lane; lanes with nothing
// to write pass zeros and a position of their own.  Three aligned dwords, shifted with one byte permute each.
WV_FN void lds_put_bytes8(Lds m, const U32& pos, const U32& lo, const U32& hi)
{
	const U32 a = pos & ~3u;
	const U32 selw = U32(0x07060504u) - splat_byte0(pos & 3u); // bytes 4-k .. 7-k of the pair {upper, lower}
	const U32 zero(0u);
	lds_or32_all(m, a, perm_bytes_v(lo, zero, selw));
	lds_or32_all(m, a + 4u, perm_bytes_v(hi, lo, selw));
	lds_or32_all(m, a + 8u, perm_bytes_v(zero, hi, selw));
}

// same for a value that cannot straddle a dword: a nibble at a nibble-aligned position, a byte at a byte-aligned one
WV_FN void lds_put_small(Lds m, const U32& bitpos, const U32& value, const Pred& p)
{
	lds_or32(m, (bitpos >> 5) << 2, value << (bitpos & 31u), p);
}

// all-reduce inside each aligned group of 4 lanes
WV_FN U32 quad_add(U32 x)
{
	x = x + shfl_xor(x, 1);
	return x + shfl_xor(x, 2);
}
WV_FN U32 quad_min(U32 x)
{
	x = umin(x, shfl_xor(x, 1));
	return umin(x, shfl_xor(x, 2));
}
WV_FN U32 quad_max(U32 x)
{
	x = umax(x, shfl_xor(x, 1));
	return umax(x, shfl_xor(x, 2));
}
// all-reduce (sum) inside each aligned group of 16 lanes
WV_FN U32 row_add(U32 x)
{
	x = x + row_ror(x, 8);
	x = x + row_ror(x, 4);
	x = x + row_ror(x, 2);
	return x + row_ror(x, 1);
}
// exclusive prefix sum inside each aligned group of 16 lanes
WV_FN U32 row_excl_scan(const U32& x)
{
	U32 s = x;
	s = s + row_shr(s, 1, 0);
	s = s + row_shr(s, 2, 0);
	s = s + row_shr(s, 4, 0);
	s = s + row_shr(s, 8, 0);
	return s - x;
}
#ifndef WV_HOST_EMULATION // (the lockstep forms of these scans and reductions: tests/emul/wavevec_host.h)
// gfx9 DPP forms: reductions inside the 16-lane rows, then four readlanes
WV_FN uint32_t wave_max(U32 x)
{
	x = umax(x, row_ror(x, 8));
	x = umax(x, row_ror(x, 4));
	x = umax(x, row_ror(x, 2));
	x = umax(x, row_ror(x, 1));
	uint32_t a = readlane(x, 0), b = readlane(x, 16), c = readlane(x, 32), d = readlane(x, 48);
	a = a > b ? a : b;
	c = c > d ? c : d;
	return a > c ? a : c;
}
// OR over the 64 lanes: inside the rows with row_ror, then the four row results through scalar registers
WV_FN uint32_t wave_or(U32 x)
{
	x |= row_ror(x, 8);
	x |= row_ror(x, 4);
	x |= row_ror(x, 2);
	x |= row_ror(x, 1);
	return readlane(x, 0) | readlane(x, 16) | readlane(x, 32) | readlane(x, 48);
}
// row_shr 1,2,4,8 with zero fill, then row_bcast:15 (0x142, rows 1 and 3) and row_bcast:31 (0x143, rows 2 and 3)
// one data movement of a wave-wide inclusive scan (see the host version): row_shr:1/2/4/8, row_bcast:15, row_bcast:31
WV_FN U32 scan_source(U32 x, int step, uint32_t ident)
{
	switch (step) {
		case 0: return (U32)__builtin_amdgcn_update_dpp((int)ident, (int)x, 0x111, 0xf, 0xf, false);
		case 1: return (U32)__builtin_amdgcn_update_dpp((int)ident, (int)x, 0x112, 0xf, 0xf, false);
		case 2: return (U32)__builtin_amdgcn_update_dpp((int)ident, (int)x, 0x114, 0xf, 0xf, false);
		case 3: return (U32)__builtin_amdgcn_update_dpp((int)ident, (int)x, 0x118, 0xf, 0xf, false);
		case 4: return (U32)__builtin_amdgcn_update_dpp((int)ident, (int)x, 0x142, 0xa, 0xf, false);
		default: return (U32)__builtin_amdgcn_update_dpp((int)ident, (int)x, 0x143, 0xc, 0xf, false);
	}
}
// the same DPP pattern with a maximum (lanes without a source read 0, the neutral element for unsigned values)
WV_FN U32 wave_incl_scan_max(U32 s)
{
	s = umax(s, dpp_zero<0x111>(s));
	s = umax(s, dpp_zero<0x112>(s));
	s = umax(s, dpp_zero<0x114>(s));
	s = umax(s, dpp_zero<0x118>(s));
	s = umax(s, (U32)__builtin_amdgcn_update_dpp(0, (int)s, 0x142, 0xa, 0xf, false));
	s = umax(s, (U32)__builtin_amdgcn_update_dpp(0, (int)s, 0x143, 0xc, 0xf, false));
	return s;
}
WV_FN U32 wave_incl_scan(U32 s)
{
	s += dpp_zero<0x111>(s);
	s += dpp_zero<0x112>(s);
	s += dpp_zero<0x114>(s);
	s += dpp_zero<0x118>(s);
	s += (U32)__builtin_amdgcn_update_dpp(0, (int)s, 0x142, 0xa, 0xf, false);
	s += (U32)__builtin_amdgcn_update_dpp(0, (int)s, 0x143, 0xc, 0xf, false);
	return s;
}
#endif

// Sixteen values, one per quad of lanes (the four lanes of a quad hold the same value): exclusive prefix sum over the
// quads, again the same in the four lanes of a quad.  The wave scan without its two in-quad steps.
WV_FN U32 quads_excl_scan(const U32& x)
{
	U32 s = x;
	s = s + scan_source(s, 2, 0);
	s = s + scan_source(s, 3, 0);
	s = s + scan_source(s, 4, 0);
	s = s + scan_source(s, 5, 0);
	return s - x;
}

// the inclusive forms, sum and maximum (values of a quad equal in its four lanes)
WV_FN U32 quads_incl_scan(const U32& x)
{
	U32 s = x;
	s = s + scan_source(s, 2, 0);
	s = s + scan_source(s, 3, 0);
	s = s + scan_source(s, 4, 0);
	return s + scan_source(s, 5, 0);
}
WV_FN U32 quads_incl_scan_max(const U32& x)
{
	U32 s = x;
	s = umax(s, scan_source(s, 2, 0));
	s = umax(s, scan_source(s, 3, 0));
	s = umax(s, scan_source(s, 4, 0));
	return umax(s, scan_source(s, 5, 0));
}
// ---- SWAR on four packed bytes ----
WV_FN U32 bytes_sub(const U32& a, const U32& b) // per-byte a - b (mod 256)
{
	const U32 H(0x80808080u);
	return ((a | H) - (b & ~H)) ^ ((a ^ ~b) & H);
}
WV_FN U32 bytes_add(const U32& a, const U32& b) // per-byte a + b (mod 256)
{
	const U32 H(0x80808080u);
	return ((a & ~H) + (b & ~H)) ^ ((a ^ b) & H);
}
// 0x80 in every byte of x that is zero, 0 elsewhere
WV_FN U32 bytes_zero_mask(const U32& x)
{
	const U32 L(0x7f7f7f7fu);
	return ~(((x & L) + L) | x | L);
}
// compress the four 0x80 flags of a zero mask into bits 0..3
// (z holds nothing but those flags: the four products land on bits 24..27, every other term below bit 24 or beyond bit 31)
WV_FN U32 zero_mask_to_bits(const U32& z) { return ((z >> 7) * 0x01020408u) >> 24; }
WV_FN U32 byte_of(const U32& x, int k) { return (x >> U32(8u * (uint32_t)k)) & 0xFFu; }
WV_FN U32 bytes_splat(const U32& b) { return (b & 0xFFu) * 0x01010101u; }
} // namespace wv
