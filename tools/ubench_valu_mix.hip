// ubench_valu_mix.hip -- how gfx950's two VALU issue classes mix (round 3).
//
// tools/ubench_valu.hip (wall clock, forced residency, PMC cross-check: profiles/r3_ubench_*) shows two classes:
//   fast  plain VOP1/VOP2 forms on VGPRs / inline constants (add, sub, and, or, xor, lshr, ashr, mov, not, the 16-bit
//         VOP2 forms, f32 add/mul/fmac) and a few VOP3 forms (v_fma_f32, v_bitop3_b32): 2.1 cycles per
//         wave-instruction and SIMD from two waves per SIMD on (one wave alone: 4.5; 64-bit encodings 5.3 / 2.7)
//   slow  everything else (packed, DPP, SDWA, most VOP3, min/max_u32, lshl_b32, mul_u24, an SGPR operand): 4.06 at
//         any occupancy
// and that a stream with ONE slow form in eight runs at 3.9 -- far from the additive 2.3.  This file measures
//   (1) runs of A slow then B fast instructions for many (A, B): what a class switch costs,
//   (2) two waves of one SIMD in DIFFERENT classes (roles by hardware wave slot): is the penalty per wave or per SIMD,
//   (3) which other instruction kinds (LDS, SALU, s_nop, VOP3 fast forms, further candidates) disturb a fast stream,
//   (4) the class of further candidate opcodes.
// Same method as ubench_valu.hip: workgroups of 256 threads (one wave per SIMD), W workgroups per CU forced through the
// dynamic LDS size, grid = rounds x CUs x W, hipEvent wall time, clock measured in the waves.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_valu_mix.hip -o tools/bin/ubench_valu_mix
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#define X8(A, B, C, D, E, F, G, H) A "\n\t" B "\n\t" C "\n\t" D "\n\t" E "\n\t" F "\n\t" G "\n\t" H "\n\t"
#define IND8(OP) X8(OP(10), OP(11), OP(12), OP(13), OP(14), OP(15), OP(16), OP(17))
#define R2(B) B B
#define R4(B) B B B B
#define R8(B) R4(B) R4(B)
#define R16(B) R8(B) R8(B)
#define R32(B) R16(B) R16(B)
#define R64(B) R32(B) R32(B)
#define CLOB "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "vcc", "scc", "s10", "s11", "memory"
#define STR2(x) #x
#define STR(x) STR2(x)

#define PROLOGUE                                                                                                  \
    asm volatile("v_mov_b32 v10, %0\n\tv_add_u32 v11, 1, %0\n\tv_add_u32 v12, 2, %0\n\tv_add_u32 v13, 3, %0\n\t"     \
                 "v_add_u32 v14, 4, %0\n\tv_add_u32 v15, 5, %0\n\tv_add_u32 v16, 6, %0\n\tv_add_u32 v17, 7, %0\n\t" \
                 "v_mul_u32_u24 v18, 3, %0\n\tv_mul_u32_u24 v19, 5, %0\n\tv_lshlrev_b32 v20, 2, %0\n\t"             \
                 "v_mov_b32 v21, 0\n\ts_mov_b64 s[10:11], 0x5555"                                                 \
                 :: "v"(threadIdx.x) : CLOB);
#define EPILOGUE(ROLE)                                                                                            \
    unsigned r;                                                                                                   \
    asm volatile("v_xor_b32 %0, v10, v11\n\tv_xor_b32 %0, %0, v12\n\tv_xor_b32 %0, %0, v13\n\tv_xor_b32 %0, %0, v14\n\t" \
                 "v_xor_b32 %0, %0, v15\n\tv_xor_b32 %0, %0, v16\n\tv_xor_b32 %0, %0, v17\n\tv_xor_b32 %0, %0, v21"  \
                 : "=v"(r) :: CLOB);                                                                              \
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;                                                        \
    if ((threadIdx.x & 63) == 0) {                                                                                \
        out[2 * (gid >> 6)] = ((t1 - t0) << 8) | ((unsigned long long)(ROLE) << 7) | (r & 0x7f);                  \
        out[2 * (gid >> 6) + 1] = r1 - r0;                                                                        \
    }

// one body for every wave
#define DEFK(NAME, BODY)                                                                                          \
    __global__ void __launch_bounds__(256) k_##NAME(unsigned long long* out, int iters)                           \
    {                                                                                                             \
        extern __shared__ unsigned lds_[];                                                                        \
        PROLOGUE                                                                                                  \
        const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();                                           \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                               \
        for (int i = 0; i < iters; ++i) asm volatile(BODY ::: CLOB);                                              \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                               \
        const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();                                           \
        EPILOGUE(0)                                                                                               \
    }
// waves in even hardware slots run BODY_A, waves in odd slots BODY_B (both bodies must have the same length)
#define DEFK2(NAME, BODY_A, BODY_B)                                                                               \
    __global__ void __launch_bounds__(256) k_##NAME(unsigned long long* out, int iters)                           \
    {                                                                                                             \
        PROLOGUE                                                                                                  \
        const unsigned role = __builtin_amdgcn_s_getreg((3 << 11) | 4) & 1u;                                      \
        const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();                                           \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                               \
        if (role == 0) {                                                                                          \
            for (int i = 0; i < iters; ++i) asm volatile(BODY_A ::: CLOB);                                        \
        } else {                                                                                                  \
            for (int i = 0; i < iters; ++i) asm volatile(BODY_B ::: CLOB);                                        \
        }                                                                                                         \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                               \
        const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();                                           \
        EPILOGUE(role)                                                                                            \
    }

// ---- instruction forms; n = accumulator register -----------------------------------------------------------------
#define O_ADD(n) "v_add_u32 v" STR(n) ", v" STR(n) ", v18"
#define O_SUB(n) "v_sub_u32 v" STR(n) ", v" STR(n) ", v18"
#define O_AND(n) "v_and_b32 v" STR(n) ", v" STR(n) ", v18"
#define O_OR(n) "v_or_b32 v" STR(n) ", v" STR(n) ", v18"
#define O_XOR(n) "v_xor_b32 v" STR(n) ", v" STR(n) ", v18"
#define O_LSHR(n) "v_lshrrev_b32 v" STR(n) ", 3, v" STR(n)
#define O_LSHL(n) "v_lshlrev_b32 v" STR(n) ", 3, v" STR(n)
#define O_PKMIN(n) "v_pk_min_u16 v" STR(n) ", v" STR(n) ", v18"
#define O_PKMAX(n) "v_pk_max_u16 v" STR(n) ", v" STR(n) ", v18"
#define O_BFI(n) "v_bfi_b32 v" STR(n) ", v" STR(n) ", v18, v19"
#define O_BITOP3(n) "v_bitop3_b32 v" STR(n) ", v" STR(n) ", v18, v19 bitop3:0xca"
#define O_FMA(n) "v_fma_f32 v" STR(n) ", v" STR(n) ", v18, v19"
#define O_ADDDPP(n) "v_add_u32_dpp v" STR(n) ", v18, v" STR(n) " wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
#define O_MOVDPP(n) "v_mov_b32_dpp v" STR(n) ", v18 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
#define O_ROWDPP(n) "v_mov_b32_dpp v" STR(n) ", v18 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
#define O_QUADDPP(n) "v_mov_b32_dpp v" STR(n) ", v18 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
#define O_ADDSG(n) "v_add_u32 v" STR(n) ", s10, v" STR(n)
#define O_ANDSG(n) "v_and_b32 v" STR(n) ", s10, v" STR(n)
#define O_ADDLIT(n) "v_add_u32 v" STR(n) ", 0x12345, v" STR(n)
#define O_ANDINL(n) "v_and_b32 v" STR(n) ", 15, v" STR(n)
#define O_MINU16(n) "v_min_u16 v" STR(n) ", v" STR(n) ", v18"
#define O_MAXU16(n) "v_max_u16 v" STR(n) ", v" STR(n) ", v18"
#define O_SNOP(n) "s_nop 0"
#define O_VNOP(n) "v_nop"
#define O_SADD(n) "s_add_u32 s10, s10, 1"
#define O_DSREAD(n) "ds_read_b32 v21, v20"
#define O_DSWRITE(n) "ds_write_b32 v20, v18"
#define O_DSBPERM(n) "ds_bpermute_b32 v21, v20, v18"
#define O_DSSWZ(n) "ds_swizzle_b32 v21, v18 offset:swizzle(SWAP,1)"
#define O_WAITL(n) "s_waitcnt lgkmcnt(0)"
// candidates whose class is not in ubench_valu's table
#define O_LSHLADD(n) "v_lshl_add_u32 v" STR(n) ", v" STR(n) ", 3, v18"
#define O_LSHLOR(n) "v_lshl_or_b32 v" STR(n) ", v" STR(n) ", 3, v18"
#define O_ADDLSHL(n) "v_add_lshl_u32 v" STR(n) ", v" STR(n) ", v18, 3"
#define O_OR3(n) "v_or3_b32 v" STR(n) ", v" STR(n) ", v18, v19"
#define O_XAD(n) "v_xad_u32 v" STR(n) ", v" STR(n) ", v18, v19"
#define O_MADU24(n) "v_mad_u32_u24 v" STR(n) ", v" STR(n) ", v18, v19"
#define O_MADU16(n) "v_mad_u16 v" STR(n) ", v" STR(n) ", v18, v19"
#define O_ADDU16(n) "v_add_u16 v" STR(n) ", v" STR(n) ", v18"
#define O_LSHRB16(n) "v_lshrrev_b16 v" STR(n) ", 3, v" STR(n)
#define O_ASHRI16(n) "v_ashrrev_i16 v" STR(n) ", 3, v" STR(n)
#define O_PKLSHR(n) "v_pk_lshrrev_b16 v" STR(n) ", 3, v" STR(n)
#define O_PKSUB(n) "v_pk_sub_u16 v" STR(n) ", v" STR(n) ", v18"
#define O_PKMULLO(n) "v_pk_mul_lo_u16 v" STR(n) ", v" STR(n) ", v18"
#define O_MIN3U16(n) "v_min3_u16 v" STR(n) ", v" STR(n) ", v18, v19"
#define O_MAX3U32(n) "v_max3_u32 v" STR(n) ", v" STR(n) ", v18, v19"
#define O_MED3(n) "v_med3_u32 v" STR(n) ", v" STR(n) ", v18, v19"
#define O_ADDCO(n) "v_add_co_u32 v" STR(n) ", vcc, v" STR(n) ", v18"
#define O_SUBCO(n) "v_sub_co_u32 v" STR(n) ", vcc, v" STR(n) ", v18"
#define O_CMPLT(n) "v_cmp_lt_u32 vcc, v" STR(n) ", v18"
#define O_CMPLTS(n) "v_cmp_lt_u32 s[10:11], v" STR(n) ", v18"
#define O_CMPLT16(n) "v_cmp_lt_u16 vcc, v" STR(n) ", v18"
#define O_SUBF(n) "v_sub_f32 v" STR(n) ", v" STR(n) ", v18"
#define O_MULF(n) "v_mul_f32 v" STR(n) ", v" STR(n) ", v18"
#define O_MINF(n) "v_min_f32 v" STR(n) ", v" STR(n) ", v18"
#define O_ADDF_ABS(n) "v_add_f32_e64 v" STR(n) ", |v" STR(n) "|, v18"
#define O_FMA_ABS(n) "v_fma_f32 v" STR(n) ", |v" STR(n) "|, v18, v19"
#define O_FMA_NEG(n) "v_fma_f32 v" STR(n) ", -v" STR(n) ", v18, v19"
#define O_FMAAK(n) "v_fmaak_f32 v" STR(n) ", v" STR(n) ", v18, 0x3d800000"
#define O_FMAMK(n) "v_fmamk_f32 v" STR(n) ", v" STR(n) ", 0x3d800000, v18"
#define O_ADDF_E64(n) "v_add_f32_e64 v" STR(n) ", v" STR(n) ", v18"
#define O_ADD_E64(n) "v_add_u32_e64 v" STR(n) ", v" STR(n) ", v18"
#define O_AND_E64(n) "v_and_b32_e64 v" STR(n) ", v" STR(n) ", v18"
#define O_FLOORF(n) "v_floor_f32 v" STR(n) ", v" STR(n)
#define O_TRUNCF(n) "v_trunc_f32 v" STR(n) ", v" STR(n)
#define O_FRACTF(n) "v_fract_f32 v" STR(n) ", v" STR(n)
#define O_CVTF32U32(n) "v_cvt_f32_u32 v" STR(n) ", v" STR(n)
#define O_CVTU32F32(n) "v_cvt_u32_f32 v" STR(n) ", v" STR(n)
#define O_PKFMA32(n) "v_pk_fma_f32 v[" STR(n) ":" "11], v[10:11], v[18:19], v[18:19]"
#define O_PKADDF16(n) "v_pk_add_f16 v" STR(n) ", v" STR(n) ", v18"
#define O_PKFMAF16(n) "v_pk_fma_f16 v" STR(n) ", v" STR(n) ", v18, v19"
#define O_ADDF16(n) "v_add_f16 v" STR(n) ", v" STR(n) ", v18"
#define O_FMAF16(n) "v_fma_f16 v" STR(n) ", v" STR(n) ", v18, v19"
#define O_FMACF16(n) "v_fmac_f16 v" STR(n) ", v18, v19"
#define O_MAXF16(n) "v_max_f16 v" STR(n) ", v" STR(n) ", v18"
#define O_DOT4(n) "v_dot4_u32_u8 v" STR(n) ", v" STR(n) ", v18, v19"
#define O_DOT2U16(n) "v_dot2_u32_u16 v" STR(n) ", v" STR(n) ", v18, v19"
#define O_MSAD(n) "v_msad_u8 v" STR(n) ", v" STR(n) ", v18, v19"
#define O_MULLOU32(n) "v_mul_lo_u32 v" STR(n) ", v" STR(n) ", v18"
#define O_SWAP(n) "v_swap_b32 v" STR(n) ", v18"
#define O_ACCMOV(n) "v_accvgpr_write_b32 a" STR(n) ", v18"
#define O_ACCRD(n) "v_accvgpr_read_b32 v" STR(n) ", a0"
#define O_MOV64(n) "v_mov_b64 v[" STR(n) ":11], v[18:19]"
#define O_LSHLADD64(n) "v_lshl_add_u64 v[10:11], v[10:11], 3, v[18:19]"
#define O_PKADDU16(n) "v_pk_add_u16 v" STR(n) ", v" STR(n) ", v18"

// eight different fast opcodes
#define FM X8(O_ADD(10), O_AND(11), O_SUB(12), O_LSHR(13), O_OR(14), O_ADD(15), O_AND(16), O_SUB(17))
#define S8 IND8(O_PKMIN)
// one "other" instruction in place of the fourth fast one
#define FM_X(OP) X8(O_ADD(10), O_AND(11), O_SUB(12), OP(13), O_OR(14), O_ADD(15), O_AND(16), O_SUB(17))
#define S1F7 FM_X(O_PKMIN)

// (1) A slow then B fast
DEFK(fast_only, R16(FM))
DEFK(slow_only, R16(S8))
DEFK(s1_f7, R16(S1F7))
DEFK(s1_f15, R8(S1F7 FM))
DEFK(s1_f31, R4(S1F7 FM FM FM))
DEFK(s1_f63, R2(S1F7 R4(FM) FM FM FM))
DEFK(s1_f127, S1F7 R8(FM) R4(FM) FM FM FM)
DEFK(s1_f255, S1F7 R16(FM) R8(FM) R4(FM) FM FM FM)
DEFK(s1_f511, S1F7 R32(FM) R16(FM) R8(FM) R4(FM) FM FM FM)
DEFK(s1_f1023, S1F7 R64(FM) R32(FM) R16(FM) R8(FM) R4(FM) FM FM FM)
DEFK(s8_f8, R8(S8 FM))
DEFK(s8_f24, R4(S8 FM FM FM))
DEFK(s8_f56, R2(S8 R4(FM) FM FM FM))
DEFK(s8_f120, S8 R8(FM) R4(FM) FM FM FM)
DEFK(s8_f248, S8 R16(FM) R8(FM) R4(FM) FM FM FM)
DEFK(s8_f504, S8 R32(FM) R16(FM) R8(FM) R4(FM) FM FM FM)
DEFK(s8_f1016, S8 R64(FM) R32(FM) R16(FM) R8(FM) R4(FM) FM FM FM)
DEFK(s32_f32, R2(R4(S8) R4(FM)))
DEFK(s32_f96, R4(S8) R8(FM) R4(FM))
DEFK(s32_f224, R4(S8) R16(FM) R8(FM) R4(FM))
DEFK(s32_f480, R4(S8) R32(FM) R16(FM) R8(FM) R4(FM))
DEFK(s32_f992, R4(S8) R64(FM) R32(FM) R16(FM) R8(FM) R4(FM))
DEFK(s128_f128, R16(S8) R16(FM))
DEFK(s128_f384, R16(S8) R32(FM) R16(FM))
DEFK(s128_f896, R16(S8) R64(FM) R32(FM) R16(FM))
DEFK(s256_f768, R32(S8) R64(FM) R32(FM))

// (2) roles by hardware wave slot
DEFK2(role_fast_fast, R16(FM), R16(FM))
DEFK2(role_slow_slow, R16(S8), R16(S8))
DEFK2(role_fast_slow, R16(FM), R16(S8))
DEFK2(role_fast_mix8, R16(FM), R16(S1F7))
DEFK2(role_fast_idle, R16(FM), R16(IND8(O_SNOP)))
DEFK2(role_slow_idle, R16(S8), R16(IND8(O_SNOP)))

// (3) one other instruction in eight
DEFK(x8_dpp_wave, R16(FM_X(O_ADDDPP)))
DEFK(x8_dpp_row, R16(FM_X(O_ROWDPP)))
DEFK(x8_dpp_quad, R16(FM_X(O_QUADDPP)))
DEFK(x8_bfi, R16(FM_X(O_BFI)))
DEFK(x8_bitop3, R16(FM_X(O_BITOP3)))
DEFK(x8_fma, R16(FM_X(O_FMA)))
DEFK(x8_lshl, R16(FM_X(O_LSHL)))
DEFK(x8_add_sgpr, R16(FM_X(O_ADDSG)))
DEFK(x8_and_sgpr, R16(FM_X(O_ANDSG)))
DEFK(x8_add_literal, R16(FM_X(O_ADDLIT)))
DEFK(x8_and_inline, R16(FM_X(O_ANDINL)))
DEFK(x8_min_u16, R16(FM_X(O_MINU16)))
DEFK(x8_s_nop, R16(FM_X(O_SNOP)))
DEFK(x8_v_nop, R16(FM_X(O_VNOP)))
DEFK(x8_salu, R16(FM_X(O_SADD)))
DEFK(x8_ds_read, R16(FM_X(O_DSREAD)) "s_waitcnt lgkmcnt(0)\n\t")
DEFK(x8_ds_write, R16(FM_X(O_DSWRITE)) "s_waitcnt lgkmcnt(0)\n\t")
DEFK(x8_ds_bpermute, R16(FM_X(O_DSBPERM)) "s_waitcnt lgkmcnt(0)\n\t")
DEFK(x8_ds_swizzle, R16(FM_X(O_DSSWZ)) "s_waitcnt lgkmcnt(0)\n\t")
DEFK(x8_waitcnt, R16(FM_X(O_WAITL)))
DEFK(x8_cmp_vcc, R16(FM_X(O_CMPLT)))
DEFK(x8_add_co, R16(FM_X(O_ADDCO)))

// (4) classes of further opcodes (independent chains, all eight accumulators)
#define CLASS(NAME, OP) DEFK(c_##NAME, R16(IND8(OP)))
CLASS(lshl_add, O_LSHLADD) CLASS(lshl_or, O_LSHLOR) CLASS(add_lshl, O_ADDLSHL) CLASS(or3, O_OR3) CLASS(xad, O_XAD)
CLASS(mad_u32_u24, O_MADU24) CLASS(mad_u16, O_MADU16) CLASS(add_u16, O_ADDU16) CLASS(lshr_b16, O_LSHRB16) CLASS(ashr_i16, O_ASHRI16)
CLASS(pk_lshr_b16, O_PKLSHR) CLASS(pk_sub_u16, O_PKSUB) CLASS(pk_mul_lo_u16, O_PKMULLO) CLASS(min3_u16, O_MIN3U16)
CLASS(max3_u32, O_MAX3U32) CLASS(med3_u32, O_MED3) CLASS(add_co, O_ADDCO) CLASS(sub_co, O_SUBCO) CLASS(cmp_lt_u32_vcc, O_CMPLT)
CLASS(cmp_lt_u32_sgpr, O_CMPLTS) CLASS(cmp_lt_u16_vcc, O_CMPLT16) CLASS(sub_f32, O_SUBF) CLASS(mul_f32, O_MULF) CLASS(min_f32, O_MINF)
CLASS(add_f32_e64_abs, O_ADDF_ABS) CLASS(fma_f32_abs, O_FMA_ABS) CLASS(fma_f32_neg, O_FMA_NEG) CLASS(fmaak_f32, O_FMAAK)
CLASS(fmamk_f32, O_FMAMK) CLASS(add_f32_e64, O_ADDF_E64) CLASS(add_u32_e64, O_ADD_E64) CLASS(and_b32_e64, O_AND_E64)
CLASS(floor_f32, O_FLOORF) CLASS(trunc_f32, O_TRUNCF) CLASS(fract_f32, O_FRACTF) CLASS(cvt_f32_u32, O_CVTF32U32)
CLASS(cvt_u32_f32, O_CVTU32F32) CLASS(pk_add_f16, O_PKADDF16) CLASS(pk_fma_f16, O_PKFMAF16) CLASS(add_f16, O_ADDF16)
CLASS(fma_f16, O_FMAF16) CLASS(max_f16, O_MAXF16) CLASS(dot4_u32_u8, O_DOT4) CLASS(dot2_u32_u16, O_DOT2U16)
CLASS(msad_u8, O_MSAD) CLASS(mul_lo_u32, O_MULLOU32) CLASS(and_sgpr, O_ANDSG) CLASS(and_inline, O_ANDINL) CLASS(min_u16, O_MINU16)
CLASS(pk_add_u16, O_PKADDU16) CLASS(bitop3, O_BITOP3) CLASS(xor, O_XOR)
DEFK(c_pk_fma_f32, R16(R8("v_pk_fma_f32 v[10:11], v[10:11], v[18:19], v[18:19]\n\t")))
DEFK(c_pk_add_f32, R16(R8("v_pk_add_f32 v[10:11], v[10:11], v[18:19]\n\t")))
DEFK(c_pk_mul_f32, R16(R8("v_pk_mul_f32 v[10:11], v[10:11], v[18:19]\n\t")))
DEFK(c_mov_b64, R16(R8("v_mov_b64 v[10:11], v[18:19]\n\t")))
DEFK(c_lshl_add_u64, R16(R8("v_lshl_add_u64 v[10:11], v[10:11], 3, v[18:19]\n\t")))
DEFK(c_swap_b32, R16(R8("v_swap_b32 v10, v11\n\t")))
DEFK(c_accvgpr_write, R16(R8("v_accvgpr_write_b32 a0, v18\n\t")))
DEFK(c_accvgpr_read, R16(R8("v_accvgpr_read_b32 v10, a0\n\t")))
DEFK(c_accvgpr_mov, R16(R8("v_accvgpr_mov_b32 a1, a0\n\t")))

typedef void (*kern_t)(unsigned long long*, int);
struct Entry {
    const char* name;
    kern_t k;
    int per_iter;
    bool roles;
};

int main(int argc, char** argv)
{
    std::vector<Entry> es = {
#define E(n, c) {#n, k_##n, c, false},
#define E2(n) {#n, k_##n, 128, true},
#define EC(n) {"class " #n, k_c_##n, 128, false},
        E(fast_only, 128) E(slow_only, 128) E(s1_f7, 128) E(s1_f15, 128) E(s1_f31, 128) E(s1_f63, 128) E(s1_f127, 128) E(s1_f255, 256)
        E(s1_f511, 512) E(s1_f1023, 1024) E(s8_f8, 128) E(s8_f24, 128) E(s8_f56, 128) E(s8_f120, 128) E(s8_f248, 256) E(s8_f504, 512)
        E(s8_f1016, 1024) E(s32_f32, 128) E(s32_f96, 128) E(s32_f224, 256) E(s32_f480, 512) E(s32_f992, 1024) E(s128_f128, 256)
        E(s128_f384, 512) E(s128_f896, 1024) E(s256_f768, 1024)
        E2(role_fast_fast) E2(role_slow_slow) E2(role_fast_slow) E2(role_fast_mix8) E2(role_fast_idle) E2(role_slow_idle)
        E(x8_dpp_wave, 128) E(x8_dpp_row, 128) E(x8_dpp_quad, 128) E(x8_bfi, 128) E(x8_bitop3, 128) E(x8_fma, 128) E(x8_lshl, 128)
        E(x8_add_sgpr, 128) E(x8_and_sgpr, 128) E(x8_add_literal, 128) E(x8_and_inline, 128) E(x8_min_u16, 128) E(x8_s_nop, 128)
        E(x8_v_nop, 128) E(x8_salu, 128) E(x8_ds_read, 128) E(x8_ds_write, 128) E(x8_ds_bpermute, 128) E(x8_ds_swizzle, 128)
        E(x8_waitcnt, 128) E(x8_cmp_vcc, 128) E(x8_add_co, 128)
        EC(lshl_add) EC(lshl_or) EC(add_lshl) EC(or3) EC(xad) EC(mad_u32_u24) EC(mad_u16) EC(add_u16) EC(lshr_b16) EC(ashr_i16)
        EC(pk_lshr_b16) EC(pk_sub_u16) EC(pk_mul_lo_u16) EC(min3_u16) EC(max3_u32) EC(med3_u32) EC(add_co) EC(sub_co)
        EC(cmp_lt_u32_vcc) EC(cmp_lt_u32_sgpr) EC(cmp_lt_u16_vcc) EC(sub_f32) EC(mul_f32) EC(min_f32) EC(add_f32_e64_abs)
        EC(fma_f32_abs) EC(fma_f32_neg) EC(fmaak_f32) EC(fmamk_f32) EC(add_f32_e64) EC(add_u32_e64) EC(and_b32_e64) EC(floor_f32)
        EC(trunc_f32) EC(fract_f32) EC(cvt_f32_u32) EC(cvt_u32_f32) EC(pk_add_f16) EC(pk_fma_f16) EC(add_f16) EC(fma_f16)
        EC(max_f16) EC(dot4_u32_u8) EC(dot2_u32_u16) EC(msad_u8) EC(mul_lo_u32) EC(and_sgpr) EC(and_inline)
        EC(min_u16) EC(pk_add_u16) EC(bitop3) EC(xor) EC(pk_fma_f32) EC(pk_add_f32) EC(pk_mul_f32) EC(mov_b64) EC(lshl_add_u64)
        EC(swap_b32) EC(accvgpr_write) EC(accvgpr_read) EC(accvgpr_mov)
    };
    std::vector<int> wlist = {1, 2, 4};
    int total = 512000, rounds = 4;
    std::vector<const char*> names;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--waves") && i + 1 < argc) {
            wlist.clear();
            for (char* t = strtok(argv[++i], ","); t; t = strtok(nullptr, ",")) wlist.push_back(atoi(t));
        } else if (!strcmp(argv[i], "--insts") && i + 1 < argc) total = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--rounds") && i + 1 < argc) rounds = atoi(argv[++i]);
        else names.push_back(argv[i]);
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) return 1;
    const int cus = prop.multiProcessorCount;
    const size_t lds_cu = 160 * 1024;
    unsigned long long* out;
    const size_t max_waves = (size_t)cus * 8 * 4 * rounds;
    if (hipMalloc(&out, max_waves * 2 * sizeof(unsigned long long)) != hipSuccess) return 1;
    std::vector<unsigned long long> host(max_waves * 2);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    printf("# %s, %d CUs; ~%d instructions per wave; grid = %d rounds x CUs x W workgroups of 256 threads (one wave per SIMD each)\n",
           prop.name, cus, total, rounds);
    printf("# per cell: cycles per wave-instruction and SIMD from hipEvent wall time x the clock measured in the waves (MHz);\n"
           "# role kernels: the wave's own s_memtime cycles per instruction for even-slot / odd-slot waves [waves of each role]\n");
    printf("%-26s", "pattern");
    for (int w : wlist) printf(" | W=%d: cyc   MHz occ", w);
    printf("\n");
    for (auto& e : es) {
        bool want = names.empty();
        for (auto n : names) want = want || strstr(e.name, n);
        if (!want) continue;
        printf("%-26s", e.name);
        const int iters = std::max(1, total / e.per_iter);
        for (int wps : wlist) {
            size_t lds = (lds_cu / wps) & ~(size_t)1023;
            if (wps == 1) lds = 96 * 1024;
            (void)hipFuncSetAttribute((const void*)e.k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            int occ = 0;
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, e.k, 256, lds);
            const int blocks = cus * wps * rounds;
            hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), lds, 0, out, 4);
            (void)hipEventRecord(e0, 0);
            hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), lds, 0, out, iters);
            (void)hipEventRecord(e1, 0);
            if (hipEventSynchronize(e1) != hipSuccess) {
                printf(" launch failed: %s\n", hipGetErrorString(hipGetLastError()));
                return 2;
            }
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e0, e1);
            const size_t waves = (size_t)blocks * 4;
            (void)hipMemcpy(host.data(), out, waves * 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
            std::vector<double> mhz, cyc[2];
            for (size_t i = 0; i < waves; ++i) {
                const double c = (double)(host[2 * i] >> 8), r = (double)host[2 * i + 1];
                cyc[(host[2 * i] >> 7) & 1].push_back(c);
                if (r > 0) mhz.push_back(c / r * 100.0);
            }
            std::nth_element(mhz.begin(), mhz.begin() + mhz.size() / 2, mhz.end());
            const double clk = mhz.empty() ? 0.0 : mhz[mhz.size() / 2];
            const double insts = (double)waves * iters * e.per_iter;
            const double ns = (double)ms * 1e6 * (cus * 4.0) / insts;
            printf(" | %9.2f %5.0f %2d", ns * clk * 1e-3, clk, occ);
            if (e.roles) {
                for (int r = 0; r < 2; ++r) {
                    if (cyc[r].empty()) { printf(" [role %d: none]", r); continue; }
                    std::nth_element(cyc[r].begin(), cyc[r].begin() + cyc[r].size() / 2, cyc[r].end());
                    printf(" [%d: %.2f x%zu]", r, cyc[r][cyc[r].size() / 2] / ((double)iters * e.per_iter), cyc[r].size());
                }
            }
        }
        printf("\n");
        fflush(stdout);
    }
    return 0;
}
