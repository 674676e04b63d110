// ubench_valu.hip -- VALU issue microbenchmark for gfx950 (round 3: wall-clock and residency checked).
//
// The whole unrolled body (128 instructions) is ONE asm block with hard-coded registers, so nothing can be
// inserted between the measured instructions (`hipcc -S --cuda-device-only` shows no s_nop inside the loops).
//
// Round 2 derived "cycles per wave-instruction and SIMD" from s_memtime inside each wave and ASSUMED that a launch
// of `cus` workgroups of 256 x W threads puts W waves on every SIMD.  Its >= 3-wave columns came out above the
// chip's vector peak, so this version trusts neither assumption:
//   * occupancy is FORCED through the dynamic LDS size (workgroups of 256 threads = one wave per SIMD; 160 KB / W
//     of LDS each, so exactly W workgroups fit a CU) and CHECKED with hipOccupancyMaxActiveBlocksPerMultiprocessor;
//   * the grid is `rounds` x (cus x W) workgroups, so placement evens out and the figure is a THROUGHPUT:
//     wave-instructions / (hipEvent wall time x SIMDs), printed in ns and in cycles of the clock the chip held,
//     which is itself measured (s_memtime ticks per s_memrealtime tick, 100 MHz, inside the loaded waves);
//   * the per-wave s_memtime figure of round 2 is printed next to it;
//   * run under `rocprofv3 --pmc SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE` the same
//     launches give a third, independent count (tools/ubench_pmc.py).
//
// For each instruction form:
//   ind   eight independent accumulator chains (a result is consumed eight instructions later)
//   dep   ONE dependent chain (every instruction consumes the previous result)
// Also: mixes of fast and slow forms shaped like the fused sweep's row body.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o /tmp/ubench_valu && /tmp/ubench_valu [--waves 1,2,4,8] [name ...]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <string>
#include <vector>

// v[10..17] accumulators, v18/v19 operands
#define X8(A, B, C, D, E, F, G, H) A "\n\t" B "\n\t" C "\n\t" D "\n\t" E "\n\t" F "\n\t" G "\n\t" H "\n\t"
#define IND8(OP) X8(OP(10), OP(11), OP(12), OP(13), OP(14), OP(15), OP(16), OP(17))
#define DEP8(OP) X8(OP(10), OP(10), OP(10), OP(10), OP(10), OP(10), OP(10), OP(10))
#define DEP2_8(OP) X8(OP(10), OP(11), OP(10), OP(11), OP(10), OP(11), OP(10), OP(11))
#define R16(B) B B B B B B B B B B B B B B B B

#define CLOB "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "vcc", "scc", "s10", "s11"

#define DEFK(NAME, BODY128)                                                                                  \
    __global__ void __launch_bounds__(256) k_##NAME(unsigned long long* out, int iters)                     \
    {                                                                                                        \
        asm volatile("v_mov_b32 v10, %0\n\tv_add_u32 v11, 1, %0\n\tv_add_u32 v12, 2, %0\n\tv_add_u32 v13, 3, %0\n\t" \
                     "v_add_u32 v14, 4, %0\n\tv_add_u32 v15, 5, %0\n\tv_add_u32 v16, 6, %0\n\tv_add_u32 v17, 7, %0\n\t" \
                     "v_mul_u32_u24 v18, 3, %0\n\tv_mul_u32_u24 v19, 5, %0\n\ts_mov_b64 s[10:11], 0x5555"       \
                     :: "v"(threadIdx.x) : CLOB);                                                            \
        const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();                                      \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                          \
        for (int i = 0; i < iters; ++i) asm volatile(BODY128 ::: CLOB);                                      \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                          \
        const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();                                      \
        unsigned r;                                                                                          \
        asm volatile("v_xor_b32 %0, v10, v11\n\tv_xor_b32 %0, %0, v12\n\tv_xor_b32 %0, %0, v13\n\tv_xor_b32 %0, %0, v14\n\t" \
                     "v_xor_b32 %0, %0, v15\n\tv_xor_b32 %0, %0, v16\n\tv_xor_b32 %0, %0, v17" : "=v"(r) :: CLOB); \
        const int gid = blockIdx.x * blockDim.x + threadIdx.x;                                               \
        if ((threadIdx.x & 63) == 0) {                                                                       \
            out[2 * (gid >> 6)] = ((t1 - t0) << 8) | (r & 0xff);                                             \
            out[2 * (gid >> 6) + 1] = r1 - r0;                                                               \
        }                                                                                                    \
    }

#define STR2(x) #x
#define STR(x) STR2(x)
// instruction forms; n = accumulator register number
#define O_ADD(n) "v_add_u32 v" STR(n) ", v" STR(n) ", v18"
#define O_SUB(n) "v_sub_u32 v" STR(n) ", v" STR(n) ", v18"
#define O_AND(n) "v_and_b32 v" STR(n) ", v" STR(n) ", v18"
#define O_OR(n) "v_or_b32 v" STR(n) ", v" STR(n) ", v18"
#define O_LSHR(n) "v_lshrrev_b32 v" STR(n) ", 3, v" STR(n)
#define O_LSHL(n) "v_lshlrev_b32 v" STR(n) ", 3, v" STR(n)
#define O_MOV(n) "v_mov_b32 v" STR(n) ", v18"
#define O_ADD3(n) "v_add3_u32 v" STR(n) ", v" STR(n) ", v18, v19"
#define O_ANDOR(n) "v_and_or_b32 v" STR(n) ", v" STR(n) ", v18, v19"
#define O_BFI(n) "v_bfi_b32 v" STR(n) ", v" STR(n) ", v18, v19"
#define O_PERM(n) "v_perm_b32 v" STR(n) ", v" STR(n) ", v18, v19"
#define O_PKMIN(n) "v_pk_min_u16 v" STR(n) ", v" STR(n) ", v18"
#define O_PKMAX(n) "v_pk_max_u16 v" STR(n) ", v" STR(n) ", v18"
#define O_PKADD(n) "v_pk_add_u16 v" STR(n) ", v" STR(n) ", v18"
#define O_PKSUBC(n) "v_pk_sub_u16 v" STR(n) ", v" STR(n) ", v18 clamp"
#define O_MINU(n) "v_min_u32 v" STR(n) ", v" STR(n) ", v18"
#define O_MUL24(n) "v_mul_u32_u24 v" STR(n) ", v" STR(n) ", v18"
#define O_SADU16(n) "v_sad_u16 v" STR(n) ", v" STR(n) ", v18, v19"
#define O_SADU8(n) "v_sad_u8 v" STR(n) ", v" STR(n) ", v18, v19"
#define O_LERP(n) "v_lerp_u8 v" STR(n) ", v" STR(n) ", v18, v19"
#define O_ALIGNBYTE(n) "v_alignbyte_b32 v" STR(n) ", v" STR(n) ", v18, 1"
#define O_CNDS(n) "v_cndmask_b32 v" STR(n) ", v" STR(n) ", v18, s[10:11]"
#define O_FMA(n) "v_fma_f32 v" STR(n) ", v" STR(n) ", v18, v19"
#define O_ADDF(n) "v_add_f32 v" STR(n) ", v" STR(n) ", v18"
#define O_MAXU16(n) "v_max_u16 v" STR(n) ", v" STR(n) ", v18"
#define O_BFE(n) "v_bfe_u32 v" STR(n) ", v" STR(n) ", 3, 8"
#define O_BITOP3(n) "v_bitop3_b32 v" STR(n) ", v" STR(n) ", v18, v19 bitop3:0xca"
// DPP forms read the neighbouring lane's accumulator; the source must not have been written by the two
// preceding VALU instructions (a hardware hazard the assembler does not pad), so they only exist in `ind` form
#define O_ADDDPP(n) "v_add_u32_dpp v" STR(n) ", v18, v" STR(n) " wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
#define O_MOVDPP(n) "v_mov_b32_dpp v" STR(n) ", v18 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
#define O_ADDSDWA(n) "v_add_u32_sdwa v" STR(n) ", v" STR(n) ", v18 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1"

#define KIND(NAME, OP) DEFK(NAME##_ind, R16(IND8(OP))) DEFK(NAME##_dep, R16(DEP8(OP))) DEFK(NAME##_dep2, R16(DEP2_8(OP)))
KIND(add, O_ADD) KIND(sub, O_SUB) KIND(and, O_AND) KIND(or, O_OR) KIND(lshr, O_LSHR) KIND(lshl, O_LSHL) KIND(mov, O_MOV)
KIND(add3, O_ADD3) KIND(and_or, O_ANDOR) KIND(bfi, O_BFI) KIND(perm, O_PERM) KIND(pk_min_u16, O_PKMIN)
KIND(pk_max_u16, O_PKMAX) KIND(pk_add_u16, O_PKADD) KIND(pk_sub_u16_clamp, O_PKSUBC) KIND(min_u32, O_MINU)
KIND(mul_u24, O_MUL24) KIND(sad_u16, O_SADU16) KIND(sad_u8, O_SADU8) KIND(lerp_u8, O_LERP) KIND(alignbyte, O_ALIGNBYTE)
KIND(cndmask_sgpr, O_CNDS) KIND(fma_f32, O_FMA) KIND(add_f32, O_ADDF) KIND(max_u16, O_MAXU16) KIND(bfe_u32, O_BFE)
KIND(bitop3, O_BITOP3) KIND(add_sdwa, O_ADDSDWA)
DEFK(add_dpp_ind, R16(IND8(O_ADDDPP))) DEFK(mov_dpp_ind, R16(IND8(O_MOVDPP)))

// s_nop 0 after every instruction: what round 1's table measured
#define O_ADDNOP(n) O_ADD(n) "\n\ts_nop 0"
#define O_PKMINNOP(n) O_PKMIN(n) "\n\ts_nop 0"
DEFK(add_nop_ind, R16(IND8(O_ADDNOP))) DEFK(pk_min_nop_ind, R16(IND8(O_PKMINNOP)))

// mixes: F = fast form, S = slow form.  `mix30` has the sweep's share of slow forms (3 in 10, spread out),
// `mix30_dep` the same with the fast instructions in ONE dependent chain (the sliding box sum), `mix50` every other.
#define MIX8_30(a, b, c, d, e, f, g, h) X8(O_ADD(a), O_AND(b), O_PKMIN(c), O_SUB(d), O_LSHR(e), O_PKMAX(f), O_OR(g), O_ADD(h))
#define MIX8_25(a, b, c, d, e, f, g, h) X8(O_ADD(a), O_AND(b), O_PKMIN(c), O_SUB(d), O_LSHR(e), O_ADD(f), O_BFI(g), O_ADD(h))
DEFK(mix_f6s2_ind, R16(MIX8_30(10, 11, 12, 13, 14, 15, 16, 17)))
DEFK(mix_f6s2_dep, R16(MIX8_30(10, 10, 11, 10, 10, 12, 10, 10)))
DEFK(mix_f6s2b_ind, R16(MIX8_25(10, 11, 12, 13, 14, 15, 16, 17)))
#define MIX8_50(a, b, c, d, e, f, g, h) X8(O_ADD(a), O_PKMIN(b), O_AND(c), O_PKMAX(d), O_SUB(e), O_BFI(f), O_LSHR(g), O_PERM(h))
DEFK(mix_f4s4_ind, R16(MIX8_50(10, 11, 12, 13, 14, 15, 16, 17)))
// slow forms in pairs / in runs of four (how the compiler emits the abs-diff block) against spread out
#define MIX8_PAIR(a, b, c, d, e, f, g, h) X8(O_PKMAX(a), O_PKMIN(b), O_SUB(c), O_ADD(d), O_ADD(e), O_SUB(f), O_AND(g), O_LSHR(h))
DEFK(mix_s2f6_ind, R16(MIX8_PAIR(10, 11, 12, 13, 14, 15, 16, 17)))
// SALU in between: scalar instructions take an issue slot of the wave, not the vector pipe
#define O_SADD(n) "s_add_u32 s10, s10, 1"
#define MIX8_SALU(a, b, c, d, e, f, g, h) X8(O_ADD(a), O_AND(b), O_SADD(c), O_SUB(d), O_LSHR(e), O_SADD(f), O_OR(g), O_ADD(h))
DEFK(mix_f6salu2_ind, R16(MIX8_SALU(10, 11, 12, 13, 14, 15, 16, 17)))

// ---- second series: how the mix matters -------------------------------------------------------------------
// different fast opcodes only (no slow form at all)
#define FMIX8(a, b, c, d, e, f, g, h) X8(O_ADD(a), O_AND(b), O_SUB(c), O_LSHR(d), O_OR(e), O_ADD(f), O_AND(g), O_SUB(h))
DEFK(fastmix_ind, R16(FMIX8(10, 11, 12, 13, 14, 15, 16, 17)))
DEFK(fastmix_dep, R16(FMIX8(10, 10, 10, 10, 10, 10, 10, 10)))
// share of slow forms: 1 in 32, 1 in 16, 1 in 8 (the rest different fast opcodes)
#define FMIX8_S1(a, b, c, d, e, f, g, h) X8(O_ADD(a), O_AND(b), O_SUB(c), O_PKMIN(d), O_OR(e), O_ADD(f), O_AND(g), O_SUB(h))
#define FMIX8_D1(a, b, c, d, e, f, g, h) X8(O_ADD(a), O_AND(b), O_SUB(c), O_ADDDPP(d), O_OR(e), O_ADD(f), O_AND(g), O_SUB(h))
#define FMIX8_B1(a, b, c, d, e, f, g, h) X8(O_ADD(a), O_AND(b), O_SUB(c), O_BFI(d), O_OR(e), O_ADD(f), O_AND(g), O_SUB(h))
#define FM FMIX8(10, 11, 12, 13, 14, 15, 16, 17)
#define FS FMIX8_S1(10, 11, 12, 13, 14, 15, 16, 17)
#define FD FMIX8_D1(10, 11, 12, 13, 14, 15, 16, 17)
#define FB FMIX8_B1(10, 11, 12, 13, 14, 15, 16, 17)
DEFK(slow1in32, FM FM FM FS FM FM FM FS FM FM FM FS FM FM FM FS)
DEFK(slow1in16, FM FS FM FS FM FS FM FS FM FS FM FS FM FS FM FS)
DEFK(slow1in8, R16(FS))
DEFK(dpp1in8, R16(FD))
DEFK(bfi1in8, R16(FB))
// slow forms bunched: 16 in a row, then 112 fast (same 1-in-8 share)
#define S8 IND8(O_PKMIN)
DEFK(slow16_fast112, S8 S8 FM FM FM FM FM FM FM FM FM FM FM FM FM FM)
// 32 slow in a row, then 96 fast (1 in 4)
DEFK(slow32_fast96, S8 S8 S8 S8 FM FM FM FM FM FM FM FM FM FM FM FM)
// the sweep's own opcode mix (k_fused_u8_v3<4,0> main loop: and 18 %, add 18 %, lshr 11 %, sub 9 %, or 7 %, pk_min 8 %,
// pk_max 4 %, add3 4 %, and_or 4 %, DPP 5 %, bfi 2 %, perm 1 %, mul_u24 2 %, lshl 1 %), 32-instruction pattern
#define SW32 X8(O_PKMAX(10), O_PKMIN(11), O_SUB(12), O_ADD(13), O_ADD(14), O_SUB(15), O_AND(16), O_LSHR(17)) \
             X8(O_ADD(10), O_OR(11), O_PKMIN(12), O_AND(13), O_LSHR(14), O_ADDDPP(15), O_ADD3(16), O_AND(17)) \
             X8(O_PKMAX(10), O_PKMIN(11), O_SUB(12), O_ADD(13), O_ANDOR(14), O_AND(15), O_LSHR(16), O_OR(17)) \
             X8(O_ADD(10), O_AND(11), O_BFI(12), O_MUL24(13), O_LSHR(14), O_ADDDPP(15), O_ADD(16), O_AND(17))
DEFK(sweepmix, SW32 SW32 SW32 SW32)
// the same without its packed min/max (as if |a-b| and the key minimum came for free): what is left of the slow forms
#define SW32B X8(O_ADD(10), O_ADD(11), O_SUB(12), O_ADD(13), O_ADD(14), O_SUB(15), O_AND(16), O_LSHR(17)) \
              X8(O_ADD(10), O_OR(11), O_AND(12), O_AND(13), O_LSHR(14), O_ADDDPP(15), O_ADD3(16), O_AND(17)) \
              X8(O_SUB(10), O_ADD(11), O_SUB(12), O_ADD(13), O_ANDOR(14), O_AND(15), O_LSHR(16), O_OR(17)) \
              X8(O_ADD(10), O_AND(11), O_BFI(12), O_MUL24(13), O_LSHR(14), O_ADDDPP(15), O_ADD(16), O_AND(17))
DEFK(sweepmix_nopk, SW32B SW32B SW32B SW32B)
// two-operand 16-bit forms on the low halves (fast class) as a replacement for packed min / max
#define O_MINU16(n) "v_min_u16 v" STR(n) ", v" STR(n) ", v18"
#define O_SUBU16(n) "v_sub_u16 v" STR(n) ", v" STR(n) ", v18"
#define O_LSHLB16(n) "v_lshlrev_b16 v" STR(n) ", 4, v" STR(n)
#define O_NOT(n) "v_not_b32 v" STR(n) ", v" STR(n)
#define O_XOR(n) "v_xor_b32 v" STR(n) ", v" STR(n) ", v18"
#define O_ASHR(n) "v_ashrrev_i32 v" STR(n) ", 3, v" STR(n)
#define O_SUBREV(n) "v_subrev_u32 v" STR(n) ", v" STR(n) ", v18"
#define O_ADDC(n) "v_addc_co_u32 v" STR(n) ", vcc, v" STR(n) ", v18, vcc"
#define O_MAXI16(n) "v_max_i16 v" STR(n) ", v" STR(n) ", v18"
#define O_CNDV(n) "v_cndmask_b32 v" STR(n) ", v" STR(n) ", v18, vcc"
#define O_MULLO16(n) "v_mul_lo_u16 v" STR(n) ", v" STR(n) ", v18"
#define O_ADDF16(n) "v_add_f16 v" STR(n) ", v" STR(n) ", v18"
#define O_MULF(n) "v_mul_f32 v" STR(n) ", v" STR(n) ", v18"
#define O_FMAC(n) "v_fmac_f32 v" STR(n) ", v18, v19"
#define O_MAXF(n) "v_max_f32 v" STR(n) ", v" STR(n) ", v18"
#define O_CVTU8(n) "v_cvt_f32_ubyte0 v" STR(n) ", v" STR(n)
#define O_ADDLIT(n) "v_add_u32 v" STR(n) ", 0x12345, v" STR(n)
#define O_ANDLIT(n) "v_and_b32 v" STR(n) ", 0xff00ff, v" STR(n)
#define O_ADDE64(n) "v_add_u32_e64 v" STR(n) ", v" STR(n) ", s10"
#define O_ADDSG(n) "v_add_u32 v" STR(n) ", s10, v" STR(n)
KIND(min_u16, O_MINU16) KIND(sub_u16, O_SUBU16) KIND(lshl_b16, O_LSHLB16) KIND(not_b32, O_NOT) KIND(xor_b32, O_XOR)
KIND(ashr, O_ASHR) KIND(subrev, O_SUBREV) KIND(max_i16, O_MAXI16) KIND(cndmask_vcc, O_CNDV) KIND(mul_lo_u16, O_MULLO16)
KIND(add_f16, O_ADDF16) KIND(mul_f32, O_MULF) KIND(fmac_f32, O_FMAC) KIND(max_f32, O_MAXF) KIND(cvt_f32_ubyte0, O_CVTU8)
KIND(add_literal, O_ADDLIT) KIND(and_literal, O_ANDLIT) KIND(add_e64_sgpr, O_ADDE64) KIND(add_sgpr, O_ADDSG)

typedef void (*kern_t)(unsigned long long*, int);
struct Entry {
    const char* name;
    kern_t k;
};

int main(int argc, char** argv)
{
    std::vector<Entry> es = {
#define E3(n) {#n " ind", k_##n##_ind}, {#n " dep", k_##n##_dep}, {#n " dep2", k_##n##_dep2},
        E3(add) E3(sub) E3(and) E3(or) E3(lshr) E3(mov) E3(fma_f32) E3(add_f32) E3(max_u16)
        E3(lshl) E3(add3) E3(and_or) E3(bfi) E3(bitop3) E3(perm) E3(pk_min_u16) E3(pk_max_u16) E3(pk_add_u16) E3(pk_sub_u16_clamp)
        E3(min_u32) E3(mul_u24) E3(sad_u16) E3(sad_u8) E3(lerp_u8) E3(alignbyte) E3(cndmask_sgpr) E3(bfe_u32) E3(add_sdwa)
        {"add_dpp ind", k_add_dpp_ind}, {"mov_dpp ind", k_mov_dpp_ind},
        {"add+s_nop ind", k_add_nop_ind}, {"pk_min+s_nop ind", k_pk_min_nop_ind},
        {"mix 6 fast 2 slow ind", k_mix_f6s2_ind}, {"mix 6 fast(dep) 2 slow", k_mix_f6s2_dep},
        {"mix 6 fast 2 slow(b) ind", k_mix_f6s2b_ind}, {"mix 4 fast 4 slow ind", k_mix_f4s4_ind},
        {"mix slow pair + 6 fast", k_mix_s2f6_ind},
        {"fast opcodes mixed ind", k_fastmix_ind}, {"fast opcodes mixed dep", k_fastmix_dep},
        {"slow 1 in 32", k_slow1in32}, {"slow 1 in 16", k_slow1in16}, {"slow 1 in 8 (pk_min)", k_slow1in8},
        {"slow 1 in 8 (add_dpp)", k_dpp1in8}, {"slow 1 in 8 (bfi)", k_bfi1in8},
        {"16 slow then 112 fast", k_slow16_fast112}, {"32 slow then 96 fast", k_slow32_fast96},
        {"sweep opcode mix", k_sweepmix}, {"sweep mix w/o pk min/max", k_sweepmix_nopk},
        E3(min_u16) E3(sub_u16) E3(lshl_b16) E3(not_b32) E3(xor_b32) E3(ashr) E3(subrev) E3(max_i16) E3(cndmask_vcc)
        E3(mul_lo_u16) E3(add_f16) E3(mul_f32) E3(fmac_f32) E3(max_f32) E3(cvt_f32_ubyte0) E3(add_literal) E3(and_literal)
        E3(add_e64_sgpr) E3(add_sgpr)
        {"mix 6 fast 2 salu", k_mix_f6salu2_ind},
    };
    // arguments: [--waves 1,2,4] [--iters N] [--rounds R] substrings of the names to run
    std::vector<int> wlist = {1, 2, 3, 4, 8};
    int iters = 4000, rounds = 4;
    std::vector<const char*> names;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--waves") && i + 1 < argc) {
            wlist.clear();
            for (char* t = strtok(argv[++i], ","); t; t = strtok(nullptr, ",")) wlist.push_back(atoi(t));
        } else if (!strcmp(argv[i], "--iters") && i + 1 < argc) iters = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--rounds") && i + 1 < argc) rounds = atoi(argv[++i]);
        else names.push_back(argv[i]);
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) return 1;
    const int cus = prop.multiProcessorCount;
    const int per_iter = 128;
    const size_t lds_cu = 160 * 1024;
    unsigned long long* out;
    const size_t max_waves = (size_t)cus * 8 * 4 * rounds;
    if (hipMalloc(&out, max_waves * 2 * sizeof(unsigned long long)) != hipSuccess) return 1;
    std::vector<unsigned long long> host(max_waves * 2);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    printf("# %s, %d CUs (%d SIMDs); %d-instruction asm body x %d iterations; grid = %d rounds x CUs x W workgroups of 256 threads\n",
           prop.name, cus, cus * 4, per_iter, iters, rounds);
    printf("# per cell: ns per wave-instruction and SIMD by hipEvent wall time | cycles = ns x measured clock | the wave's own\n"
           "# s_memtime cycles / instructions / W (round 2's figure) ; clock = s_memtime ticks per 10 ns of s_memrealtime, median wave\n");
    printf("%-26s", "instruction");
    for (int w : wlist) printf(" | W=%d: ns   cyc  wave-cyc/W  MHz occ", w);
    printf("\n");
    for (auto& e : es) {
        bool want = names.empty();
        for (auto n : names) want = want || strstr(e.name, n);
        if (!want) continue;
        printf("%-26s", e.name);
        for (int wps : wlist) {
            // one workgroup = 4 waves = one per SIMD; W workgroups per CU through the LDS each one asks for
            size_t lds = (lds_cu / wps) & ~(size_t)1023;
            if (wps == 1) lds = 96 * 1024;
            (void)hipFuncSetAttribute((const void*)e.k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            int occ = 0;
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, e.k, 256, lds);
            const int blocks = cus * wps * rounds;
            hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), lds, 0, out, 16);   // warm the code path
            (void)hipEventRecord(e0, 0);
            hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), lds, 0, out, iters);
            (void)hipEventRecord(e1, 0);
            if (hipEventSynchronize(e1) != hipSuccess) return 2;
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e0, e1);
            const size_t waves = (size_t)blocks * 4;
            (void)hipMemcpy(host.data(), out, waves * 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
            std::vector<double> cyc, mhz;
            for (size_t i = 0; i < waves; ++i) {
                const double c = (double)(host[2 * i] >> 8), r = (double)host[2 * i + 1];
                cyc.push_back(c);
                if (r > 0) mhz.push_back(c / r * 100.0);
            }
            std::nth_element(cyc.begin(), cyc.begin() + cyc.size() / 2, cyc.end());
            std::nth_element(mhz.begin(), mhz.begin() + mhz.size() / 2, mhz.end());
            const double wave_cycles = cyc[cyc.size() / 2], clk = mhz.empty() ? 0.0 : mhz[mhz.size() / 2];
            const double insts = (double)waves * iters * per_iter;
            const double ns = (double)ms * 1e6 * (cus * 4.0) / insts;
            printf(" | %9.3f %5.2f %9.2f %5.0f %2d", ns, ns * clk * 1e-3, wave_cycles / ((double)iters * per_iter) / wps, clk, occ);
        }
        printf("\n");
        fflush(stdout);
    }
    return 0;
}
