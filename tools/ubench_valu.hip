// ubench_valu.hip -- how many cycles one SIMD of gfx950 needs per wave64 vector instruction, for the instruction
// kinds the block codec is made of, as a function of resident waves per SIMD and of dependence between consecutive
// instructions.  Decides whether "VALU bound" means 4 or 2 cycles per instruction.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o /tmp/ubench_valu && /tmp/ubench_valu
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#define CHECK(x)                                                                                                       \
	do {                                                                                                               \
		hipError_t e = (x);                                                                                            \
		if (e != hipSuccess) {                                                                                         \
			printf("%s: %s\n", #x, hipGetErrorString(e));                                                              \
			return 1;                                                                                                  \
		}                                                                                                              \
	} while (0)

constexpr int UNROLL = 64; // instructions per loop trip (8 chains x 8)

// 8 independent accumulators, each instruction depends on the one 8 places earlier (IND) or on the previous one (DEP)
#define BODY8(INS)                                                                                                     \
	INS(a0) INS(a1) INS(a2) INS(a3) INS(a4) INS(a5) INS(a6) INS(a7)
#define BODY_IND(INS) BODY8(INS) BODY8(INS) BODY8(INS) BODY8(INS) BODY8(INS) BODY8(INS) BODY8(INS) BODY8(INS)
#define BODY1(INS) INS(a0) INS(a0) INS(a0) INS(a0) INS(a0) INS(a0) INS(a0) INS(a0)
#define BODY_DEP(INS) BODY1(INS) BODY1(INS) BODY1(INS) BODY1(INS) BODY1(INS) BODY1(INS) BODY1(INS) BODY1(INS)

#define I_ADD(r) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r) : "v"(k));
#define I_AND(r) asm volatile("v_and_b32 %0, %0, %1" : "+v"(r) : "v"(k));
#define I_XOR(r) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(r) : "v"(k));
#define I_LSHL(r) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(r));
#define I_ANDOR(r) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(r) : "v"(k), "v"(k2));
#define I_LSHLADD(r) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(r) : "v"(k));
#define I_ADD3(r) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(r) : "v"(k), "v"(k2));
#define I_BFE(r) asm volatile("v_bfe_u32 %0, %0, 3, 9" : "+v"(r));
#define I_PERM(r) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(r) : "v"(k), "v"(k2));
#define I_PKMIN(r) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(r) : "v"(k));
#define I_MIN(r) asm volatile("v_min_u32 %0, %0, %1" : "+v"(r) : "v"(k));
#define I_MIN3(r) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(r) : "v"(k), "v"(k2));
#define I_CNDMASK(r) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r) : "v"(k));
#define I_BCNT(r) asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(r) : "v"(k));
#define I_FFBH(r) asm volatile("v_ffbh_u32 %0, %0" : "+v"(r));
#define I_MULLO(r) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r) : "v"(k));
#define I_MUL24(r) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(r) : "v"(k));
#define I_MAD24(r) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(r) : "v"(k), "v"(k2));
#define I_DPPQ(r) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(r));
#define I_DPPADD(r) asm volatile("v_add_u32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r) : "v"(k));
#define I_DPPROR(r) asm volatile("v_add_u32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf" : "+v"(r));
#define I_SDWA(r) asm volatile("v_min_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2" : "+v"(r) : "v"(k));
#define I_SADU8(r) asm volatile("v_sad_u8 %0, %0, %1, %2" : "+v"(r) : "v"(k), "v"(k2));
#define I_CMP(r) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(r), "v"(k) : "vcc");
#define I_CMPS(r) asm volatile("v_cmp_lt_u32 s[20:21], %0, %1" : : "v"(r), "v"(k) : "s20", "s21");
#define I_READLANE(r) asm volatile("v_readlane_b32 s20, %0, 3" : : "v"(r) : "s20");
#define I_FMA(r) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r) : "v"(k), "v"(k2));
#define I_PKFMA(r) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(r##w) : "v"(kw));
#define I_SALU(r) asm volatile("s_add_u32 s20, s20, 1" : : : "s20", "scc");
#define I_BPERM(r) asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(r) : "v"(k));
#define I_BPERM_NOWAIT(r) asm volatile("ds_bpermute_b32 %0, %1, %0" : "+v"(r) : "v"(k));


#define I_OR(r) asm volatile("v_or_b32 %0, %0, %1" : "+v"(r) : "v"(k));
#define I_SUB(r) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(r) : "v"(k));
#define I_LSHR(r) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(r));
#define I_LSHLV(r) asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(r) : "v"(k));
#define I_MAX(r) asm volatile("v_max_u32 %0, %0, %1" : "+v"(r) : "v"(k));
#define I_MOV(r) asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(k));
#define I_NOT(r) asm volatile("v_not_b32 %0, %0" : "+v"(r));
#define I_BFI(r) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(r) : "v"(k), "v"(k2));
#define I_ALIGNBIT(r) asm volatile("v_alignbit_b32 %0, %0, %1, 8" : "+v"(r) : "v"(k));
#define I_ADDLIT(r) asm volatile("v_add_u32 %0, 0x12345678, %0" : "+v"(r));
#define I_ANDLIT(r) asm volatile("v_and_b32 %0, 0x7f7f7f7f, %0" : "+v"(r));
#define I_ADDS(r) asm volatile("v_add_u32 %0, s20, %0" : "+v"(r));
#define I_ADDCO(r) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(r) : "v"(k) : "vcc");
#define I_ADDE64(r) asm volatile("v_add_u32_e64 %0, %0, %1" : "+v"(r) : "v"(k));
#define I_XORE64(r) asm volatile("v_xor_b32_e64 %0, %0, %1" : "+v"(r) : "v"(k));
#define I_CNDVCC(r) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r) : "v"(k));
#define I_CNDS(r) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[22:23]" : "+v"(r) : "v"(k));
#define I_CMPCND(r) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r) : "v"(k) : "vcc");
#define I_ADDCND(r) asm volatile("v_add_u32 %0, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r) : "v"(k));
#define I_MBCNT(r) asm volatile("v_mbcnt_lo_u32_b32 %0, -1, %0" : "+v"(r));
#define I_PKADD(r) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(r) : "v"(k));
#define I_PKSUB(r) asm volatile("v_pk_sub_u16 %0, %0, %1" : "+v"(r) : "v"(k));
#define I_PKLSHL(r) asm volatile("v_pk_lshlrev_b16 %0, 1, %0" : "+v"(r));
#define I_ADDU16(r) asm volatile("v_add_u16 %0, %0, %1" : "+v"(r) : "v"(k));
#define I_SDWAADD(r) asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "+v"(r) : "v"(k));
#define I_SDWADST(r) asm volatile("v_sub_u16_sdwa %0, %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_1 src1_sel:BYTE_0" : "+v"(r) : "v"(k));
#define I_DSRD32(r) asm volatile("ds_read_b32 %0, %1" : "=v"(r) : "v"(la));
#define I_DSRD64(r) asm volatile("ds_read_b64 %0, %1" : "=v"(r##w) : "v"(la4));
#define I_DSRD128(r) asm volatile("ds_read_b128 %0, %1" : "=v"(q##r) : "v"(la4));
#define I_DSWR32(r) asm volatile("ds_write_b32 %1, %0" : : "v"(r), "v"(la));
#define I_DSWR128(r) asm volatile("ds_write_b128 %1, %0" : : "v"(q##r), "v"(la4));
#define I_DSOR(r) asm volatile("ds_or_b32 %1, %0" : : "v"(r), "v"(la));
#define I_DSORU(r) asm volatile("ds_or_b32 %1, %0" : : "v"(r), "v"(lau));
#define I_DSRDU8(r) asm volatile("ds_read_u8 %0, %1" : "=v"(r) : "v"(lab));
#define I_DSWRB8(r) asm volatile("ds_write_b8 %1, %0" : : "v"(r), "v"(lab));

// ---- round 4: operands from scalar registers, 16-bit forms, lane swaps, LDS widths -------------------------------------
#define I_XORS(r) asm volatile("v_xor_b32 %0, s20, %0" : "+v"(r));
#define I_ANDS(r) asm volatile("v_and_b32 %0, s20, %0" : "+v"(r));
#define I_MOVS(r) asm volatile("v_mov_b32 %0, s20" : "=v"(r));
#define I_MOVLIT(r) asm volatile("v_mov_b32 %0, 0x12345678" : "=v"(r));
#define I_PERMS(r) asm volatile("v_perm_b32 %0, %0, %1, s20" : "+v"(r) : "v"(k));
#define I_BITOP3(r) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xc8" : "+v"(r) : "v"(k), "v"(k2));
#define I_PKMINSEL(r) asm volatile("v_pk_min_u16 %0, %0, %0 op_sel:[0,1] op_sel_hi:[1,0]" : "+v"(r));
#define I_PLSWAP(r) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(r), "+v"(k));
#define I_PL16SWAP(r) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(r), "+v"(k));
#define I_MINU16(r) asm volatile("v_min_u16 %0, %0, %1" : "+v"(r) : "v"(k));
#define I_MAXU16(r) asm volatile("v_max_u16 %0, %0, %1" : "+v"(r) : "v"(k));
#define I_SUBU16(r) asm volatile("v_sub_u16 %0, %0, %1" : "+v"(r) : "v"(k));
#define I_LSHRB16(r) asm volatile("v_lshrrev_b16 %0, 1, %0" : "+v"(r));
#define I_LSHLB16(r) asm volatile("v_lshlrev_b16 %0, 1, %0" : "+v"(r));
#define I_LSHRV(r) asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(r) : "v"(k));
#define I_ASHR(r) asm volatile("v_ashrrev_i32 %0, 1, %0" : "+v"(r));
#define I_MINI32(r) asm volatile("v_min_i32 %0, %0, %1" : "+v"(r) : "v"(k));
#define I_XAD(r) asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(r) : "v"(k), "v"(k2));
#define I_LSHLOR(r) asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(r) : "v"(k));
#define I_OR3(r) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(r) : "v"(k), "v"(k2));
#define I_ADDF(r) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r) : "v"(k));
#define I_MULF(r) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r) : "v"(k));
#define I_CVTUB(r) asm volatile("v_cvt_f32_ubyte1 %0, %0" : "+v"(r));
#define I_SUBREV(r) asm volatile("v_subrev_u32 %0, %0, %1" : "+v"(r) : "v"(k));
#define I_ANDINL(r) asm volatile("v_and_b32 %0, 15, %0" : "+v"(r));
#define I_DOT4(r) asm volatile("v_dot4_u32_u8 %0, %0, %1, %2" : "+v"(r) : "v"(k), "v"(k2));
#define I_MSAD(r) asm volatile("v_msad_u8 %0, %0, %1, %2" : "+v"(r) : "v"(k), "v"(k2));
#define I_SLOAD(r) asm volatile("s_load_dword s20, %0, 0x0\n s_waitcnt lgkmcnt(0)" : : "s"(cyc) : "s20");
#define I_SLOADNW(r) asm volatile("s_load_dword s20, %0, 0x0" : : "s"(cyc) : "s20");
#define I_DSWR64(r) asm volatile("ds_write_b64 %1, %0" : : "v"(r##w), "v"(la4));
#define I_DSWR64U(r) asm volatile("ds_write_b64 %1, %0" : : "v"(r##w), "v"(lab));
#define I_DSWR32U(r) asm volatile("ds_write_b32 %1, %0" : : "v"(r), "v"(lab));
#define I_DSWR128U(r) asm volatile("ds_write_b128 %1, %0" : : "v"(q##r), "v"(lab));
#define I_DSOR64(r) asm volatile("ds_or_b64 %1, %0" : : "v"(r##w), "v"(la4));
#define I_DSWR16(r) asm volatile("ds_write_b16 %1, %0" : : "v"(r), "v"(lab));
#define I_DSRD128U(r) asm volatile("ds_read_b128 %0, %1" : "=v"(q##r) : "v"(lab));
#define I_DSWR2(r) asm volatile("ds_write2_b32 %1, %0, %0 offset0:1 offset1:65" : : "v"(r), "v"(la));
#define I_DSWR2ST64(r) asm volatile("ds_write2st64_b32 %1, %0, %0 offset0:1 offset1:3" : : "v"(r), "v"(la));

typedef uint32_t u32;
typedef uint64_t u64;

#define KERNEL(NAME, BODY, INS)                                                                                        \
	__global__ void __launch_bounds__(256) NAME(u32* out, int trips, u64* cyc)                                         \
	{                                                                                                                  \
		u32 a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
		u64 a0w = a0, a1w = a1, a2w = a2, a3w = a3, a4w = a4, a5w = a5, a6w = a6, a7w = a7, kw = 0x3f8000003f800000ull; \
		u32 k = 0x01020305u + (threadIdx.x & 1), k2 = 0x07060504u;                                                     \
		typedef u32 u32x4 __attribute__((ext_vector_type(4)));                                                         \
		u32x4 qa0 = {a0,a1,a2,a3}, qa1 = qa0, qa2 = qa0, qa3 = qa0, qa4 = qa0, qa5 = qa0, qa6 = qa0, qa7 = qa0;     \
		__shared__ u32 ldsbuf[4096];                                                                                   \
		ldsbuf[threadIdx.x] = a0; ldsbuf[threadIdx.x + 256] = a1;                                                      \
		__syncthreads();                                                                                               \
		u32 la = (threadIdx.x & 63) * 4 + (threadIdx.x >> 6) * 4096, la4 = (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 4096; \
		u32 lau = ((threadIdx.x * 37) & 255) * 4 + (threadIdx.x >> 6) * 4096, lab = ((threadIdx.x * 37) & 255) * 3 + (threadIdx.x >> 6) * 4096;                                            \
		asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cmp_lt_u32 s[22:23], %0, %1\n s_mov_b32 s20, 77" : : "v"(a0 & 3), "v"(k & 3) : "vcc", "s20", "s22", "s23"); \
		u64 t0 = __builtin_readcyclecounter();                                                                         \
		for (int i = 0; i < trips; ++i) {                                                                              \
			BODY(INS)                                                                                                  \
		}                                                                                                              \
		asm volatile("s_waitcnt lgkmcnt(0)");                                                                          \
		u64 t1 = __builtin_readcyclecounter();                                                                         \
		u32 s = qa0.x + qa1.y + qa2.z + qa3.w + qa4.x + qa5.x + qa6.x + qa7.x + a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (u32)(a0w + a1w + a2w + a3w + a4w + a5w + a6w + a7w);          \
		if (s == 0x12345u) out[0] = s;                                                                                 \
		if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;                                                     \
	}

#define BOTH(N, INS) KERNEL(N##_ind, BODY_IND, INS) KERNEL(N##_dep, BODY_DEP, INS)
BOTH(k_add, I_ADD)
BOTH(k_and, I_AND)
BOTH(k_lshl, I_LSHL)
BOTH(k_andor, I_ANDOR)
BOTH(k_lshladd, I_LSHLADD)
BOTH(k_add3, I_ADD3)
BOTH(k_bfe, I_BFE)
BOTH(k_perm, I_PERM)
BOTH(k_pkmin, I_PKMIN)
BOTH(k_min, I_MIN)
BOTH(k_min3, I_MIN3)
BOTH(k_cndmask, I_CNDMASK)
BOTH(k_bcnt, I_BCNT)
BOTH(k_ffbh, I_FFBH)
BOTH(k_mullo, I_MULLO)
BOTH(k_mul24, I_MUL24)
BOTH(k_mad24, I_MAD24)
BOTH(k_dppq, I_DPPQ)
BOTH(k_dppadd, I_DPPADD)
BOTH(k_dppror, I_DPPROR)
BOTH(k_sdwa, I_SDWA)
BOTH(k_sadu8, I_SADU8)
BOTH(k_cmp, I_CMP)
BOTH(k_cmps, I_CMPS)
BOTH(k_readlane, I_READLANE)
BOTH(k_fma, I_FMA)
BOTH(k_pkfma, I_PKFMA)
BOTH(k_salu, I_SALU)
BOTH(k_bperm, I_BPERM)
BOTH(k_bpermnw, I_BPERM_NOWAIT)

BOTH(k_or, I_OR)
BOTH(k_sub, I_SUB)
BOTH(k_lshr, I_LSHR)
BOTH(k_lshlv, I_LSHLV)
BOTH(k_max, I_MAX)
BOTH(k_mov, I_MOV)
BOTH(k_not, I_NOT)
BOTH(k_bfi, I_BFI)
BOTH(k_alignbit, I_ALIGNBIT)
BOTH(k_addlit, I_ADDLIT)
BOTH(k_andlit, I_ANDLIT)
BOTH(k_adds, I_ADDS)
BOTH(k_addco, I_ADDCO)
BOTH(k_adde64, I_ADDE64)
BOTH(k_xore64, I_XORE64)
BOTH(k_cndvcc, I_CNDVCC)
BOTH(k_cnds, I_CNDS)
BOTH(k_cmpcnd, I_CMPCND)
BOTH(k_addcnd, I_ADDCND)
BOTH(k_mbcnt, I_MBCNT)
BOTH(k_pkadd, I_PKADD)
BOTH(k_pksub, I_PKSUB)
BOTH(k_pklshl, I_PKLSHL)
BOTH(k_addu16, I_ADDU16)
BOTH(k_sdwaadd, I_SDWAADD)
BOTH(k_sdwadst, I_SDWADST)
BOTH(k_dsrd32, I_DSRD32)
BOTH(k_dsrd64, I_DSRD64)
BOTH(k_dsrd128, I_DSRD128)
BOTH(k_dswr32, I_DSWR32)
BOTH(k_dswr128, I_DSWR128)
BOTH(k_dsor, I_DSOR)
BOTH(k_dsoru, I_DSORU)
BOTH(k_dsrdu8, I_DSRDU8)
BOTH(k_dswrb8, I_DSWRB8)

BOTH(k_xors, I_XORS)
BOTH(k_ands, I_ANDS)
BOTH(k_movs, I_MOVS)
BOTH(k_movlit, I_MOVLIT)
BOTH(k_perms, I_PERMS)
BOTH(k_bitop3, I_BITOP3)
BOTH(k_pkminsel, I_PKMINSEL)
BOTH(k_plswap, I_PLSWAP)
BOTH(k_pl16swap, I_PL16SWAP)
BOTH(k_minu16, I_MINU16)
BOTH(k_maxu16, I_MAXU16)
BOTH(k_subu16, I_SUBU16)
BOTH(k_lshrb16, I_LSHRB16)
BOTH(k_lshlb16, I_LSHLB16)
BOTH(k_lshrv, I_LSHRV)
BOTH(k_ashr, I_ASHR)
BOTH(k_mini32, I_MINI32)
BOTH(k_xad, I_XAD)
BOTH(k_lshlor, I_LSHLOR)
BOTH(k_or3, I_OR3)
BOTH(k_addf, I_ADDF)
BOTH(k_mulf, I_MULF)
BOTH(k_cvtub, I_CVTUB)
BOTH(k_subrev, I_SUBREV)
BOTH(k_andinl, I_ANDINL)
BOTH(k_dot4, I_DOT4)
BOTH(k_msad, I_MSAD)
BOTH(k_sload, I_SLOAD)
BOTH(k_sloadnw, I_SLOADNW)
BOTH(k_dswr64, I_DSWR64)
BOTH(k_dswr64u, I_DSWR64U)
BOTH(k_dswr32u, I_DSWR32U)
BOTH(k_dswr128u, I_DSWR128U)
BOTH(k_dsor64, I_DSOR64)
BOTH(k_dswr16, I_DSWR16)
BOTH(k_dsrd128u, I_DSRD128U)
BOTH(k_dswr2, I_DSWR2)
BOTH(k_dswr2st64, I_DSWR2ST64)

struct Entry {
	const char* name;
	void (*ind)(u32*, int, u64*);
	void (*dep)(u32*, int, u64*);
};
#define E(N) { #N, N##_ind, N##_dep }

int main(int argc, char**)
{
	hipDeviceProp_t prop;
	CHECK(hipGetDeviceProperties(&prop, 0));
	const int cus = prop.multiProcessorCount;
	printf("device %s, %d CUs, clock %d kHz\n", prop.name, cus, prop.clockRate);
	u32* out;
	u64* cyc;
	CHECK(hipMalloc(&out, 4096));
	CHECK(hipMalloc(&cyc, 64));
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0));
	CHECK(hipEventCreate(&e1));
	std::vector<Entry> es = { E(k_add), E(k_and), E(k_lshl), E(k_andor), E(k_lshladd), E(k_add3), E(k_bfe), E(k_perm), E(k_pkmin), E(k_min),
		E(k_min3), E(k_cndmask), E(k_bcnt), E(k_ffbh), E(k_mullo), E(k_mul24), E(k_mad24), E(k_dppq), E(k_dppadd), E(k_dppror), E(k_sdwa),
		E(k_sadu8), E(k_cmp), E(k_cmps), E(k_readlane), E(k_fma), E(k_pkfma), E(k_salu), E(k_bperm), E(k_bpermnw), E(k_or), E(k_sub), E(k_lshr), E(k_lshlv), E(k_max), E(k_mov), E(k_not), E(k_bfi), E(k_alignbit), E(k_addlit), E(k_andlit), E(k_adds), E(k_addco), E(k_adde64), E(k_xore64), E(k_cndvcc), E(k_cnds), E(k_cmpcnd), E(k_addcnd), E(k_mbcnt), E(k_pkadd), E(k_pksub), E(k_pklshl), E(k_addu16), E(k_sdwaadd), E(k_sdwadst), E(k_dsrd32), E(k_dsrd64), E(k_dsrd128), E(k_dswr32), E(k_dswr128), E(k_dsor), E(k_dsoru), E(k_dsrdu8), E(k_dswrb8) };
	if (argc > 1) es = { E(k_xors), E(k_ands), E(k_movs), E(k_movlit), E(k_perms), E(k_bitop3), E(k_pkminsel), E(k_plswap), E(k_pl16swap), E(k_minu16), E(k_maxu16), E(k_subu16), E(k_lshrb16), E(k_lshlb16), E(k_lshrv), E(k_ashr), E(k_mini32), E(k_xad), E(k_lshlor), E(k_or3), E(k_addf), E(k_mulf), E(k_cvtub), E(k_subrev), E(k_andinl), E(k_dot4), E(k_msad), E(k_sload), E(k_sloadnw), E(k_dswr64), E(k_dswr64u), E(k_dswr32u), E(k_dswr128u), E(k_dsor64), E(k_dswr16), E(k_dsrd128u), E(k_dswr2), E(k_dswr2st64) };
	const int trips = 2000;
	printf("cycles per wave-instruction per SIMD (shader clock from s_memtime of one wave / wall clock of the grid)\n");
	printf("%-12s %5s | %9s %9s | %9s %9s\n", "instr", "w/SIMD", "ind own", "ind simd", "dep own", "dep simd");
	for (auto& e : es) {
		for (int wps : { 1, 8 }) {
			// blocks of 256 threads = 4 waves = one per SIMD; wps blocks per CU
			double r[4];
			for (int d = 0; d < 2; ++d) {
				auto kern = d ? e.dep : e.ind;
				const int blocks = cus * wps;
				hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 10, cyc);
				CHECK(hipDeviceSynchronize());
				CHECK(hipEventRecord(e0));
				hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, trips, cyc);
				CHECK(hipEventRecord(e1));
				CHECK(hipDeviceSynchronize());
				float ms;
				CHECK(hipEventElapsedTime(&ms, e0, e1));
				u64 c;
				CHECK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
				const double n = (double)trips * UNROLL;
				r[2 * d] = (double)c / n;                        // cycles the measured wave needed per own instruction (s_memtime ticks at 100 MHz? see header)
				r[2 * d + 1] = ms * 1e-3 * 2.4e9 / (n * wps);   // SIMD cycles at 2.4 GHz per instruction over all its waves
			}
			printf("%-12s %5d | %9.2f %9.2f | %9.2f %9.2f\n", e.name, wps, r[0], r[1], r[2], r[3]);
		}
	}
	return 0;
}
