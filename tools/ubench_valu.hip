// ubench_valu.hip -- issue-rate microbenchmark for the integer VALU instructions the fused
// SangNom2 kernel is made of (gfx950).  For each instruction: cycles per wave-instruction per
// SIMD at 1, 2 and 4 waves per SIMD (independent register chains, no memory traffic).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o /tmp/ubench_valu && /tmp/ubench_valu
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

#define DEFK(NAME, ASM)                                                                        \
    __global__ void __launch_bounds__(1024) k_##NAME(unsigned* out, int iters)                 \
    {                                                                                          \
        unsigned r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, \
                 r6 = r0 + 6, r7 = r0 + 7, a = r0 * 3 + 1, b = r0 * 5 + 2;                     \
        asm volatile("s_mov_b64 s[10:11], 0x5555\n\ts_mov_b64 vcc, 0x3333" ::: "s10", "s11", "s12", "vcc");           \
        for (int i = 0; i < iters; ++i) {                                                      \
            _Pragma("unroll") for (int u = 0; u < 8; ++u)                                      \
            {                                                                                  \
                asm volatile(ASM(0) : "+v"(r0) : "v"(a), "v"(b) : "vcc");                              \
                asm volatile(ASM(1) : "+v"(r1) : "v"(a), "v"(b) : "vcc");                              \
                asm volatile(ASM(2) : "+v"(r2) : "v"(a), "v"(b) : "vcc");                              \
                asm volatile(ASM(3) : "+v"(r3) : "v"(a), "v"(b) : "vcc");                              \
                asm volatile(ASM(4) : "+v"(r4) : "v"(a), "v"(b) : "vcc");                              \
                asm volatile(ASM(5) : "+v"(r5) : "v"(a), "v"(b) : "vcc");                              \
                asm volatile(ASM(6) : "+v"(r6) : "v"(a), "v"(b) : "vcc");                              \
                asm volatile(ASM(7) : "+v"(r7) : "v"(a), "v"(b) : "vcc");                              \
            }                                                                                  \
        }                                                                                      \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;   \
    }

#define A_ADD(n) "v_add_u32 %0, %0, %1"
#define A_ADD3(n) "v_add3_u32 %0, %0, %1, %2"
#define A_SAD(n) "v_sad_u16 %0, %0, %1, %2"
#define A_BFE(n) "v_bfe_u32 %0, %0, 3, 8"
#define A_LSHLOR(n) "v_lshl_or_b32 %0, %0, 16, %1"
#define A_MIN(n) "v_min_u32 %0, %0, %1"
#define A_MIN3(n) "v_min3_u32 %0, %0, %1, %2"
#define A_ADDDPP(n) "v_add_u32_dpp %0, %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0"
#define A_ADDDPPROW(n) "v_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0"
#define A_MOVDPP(n) "v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0"
#define A_ADDSDWA(n) "v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1"
#define A_PKADD(n) "v_pk_add_u16 %0, %0, %1"
#define A_PKSUB(n) "v_pk_sub_i16 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0]"
#define A_PKMAX(n) "v_pk_max_i16 %0, %0, %1"
#define A_CNDMASK(n) "v_cndmask_b32 %0, %0, %1, vcc"
#define A_MUL24(n) "v_mul_u32_u24 %0, %0, %1"
#define A_MAD24(n) "v_mad_u32_u24 %0, %0, %1, %2"
#define A_AND(n) "v_and_b32 %0, %0, %1"
#define A_ANDOR(n) "v_and_or_b32 %0, %0, %1, %2"
#define A_PERM(n) "v_perm_b32 %0, %0, %1, %2"
#define A_ALIGNBYTE(n) "v_alignbyte_b32 %0, %0, %1, 1"
#define A_LERP(n) "v_lerp_u8 %0, %0, %1, %2"
#define A_SADU8(n) "v_sad_u8 %0, %0, %1, %2"
#define A_FMA(n) "v_fma_f32 %0, %0, %1, %2"
#define A_MOV(n) "v_mov_b32 %0, %1"
#define A_PKMAD(n) "v_pk_mad_u16 %0, %0, %1, %2"
#define A_PKLSHR(n) "v_pk_lshrrev_b16 %0, 4, %0"
#define A_DOT4(n) "v_dot4_u32_u8 %0, %0, %1, %2"
#define A_MSAD(n) "v_msad_u8 %0, %0, %1, %2"

DEFK(add, A_ADD) DEFK(add3, A_ADD3) DEFK(sad_u16, A_SAD) DEFK(bfe, A_BFE) DEFK(lshl_or, A_LSHLOR)
DEFK(min, A_MIN) DEFK(min3, A_MIN3) DEFK(add_dpp_wave, A_ADDDPP) DEFK(add_dpp_row, A_ADDDPPROW)
DEFK(mov_dpp_wave, A_MOVDPP) DEFK(add_sdwa, A_ADDSDWA) DEFK(pk_add_u16, A_PKADD) DEFK(pk_sub_opsel, A_PKSUB)
DEFK(pk_max_i16, A_PKMAX) DEFK(cndmask, A_CNDMASK) DEFK(mul_u24, A_MUL24) DEFK(mad_u24, A_MAD24)
DEFK(and_b32, A_AND) DEFK(and_or, A_ANDOR) DEFK(perm, A_PERM) DEFK(alignbyte, A_ALIGNBYTE) DEFK(lerp_u8, A_LERP)
DEFK(sad_u8, A_SADU8) DEFK(fma_f32, A_FMA) DEFK(mov, A_MOV) DEFK(pk_mad_u16, A_PKMAD) DEFK(pk_lshr, A_PKLSHR)
DEFK(dot4_u8, A_DOT4) DEFK(msad_u8, A_MSAD)


#define A_SUB(n) "v_sub_u32 %0, %0, %1"
#define A_SUBREV(n) "v_subrev_u32 %0, %0, %1"
#define A_OR(n) "v_or_b32 %0, %0, %1"
#define A_XOR(n) "v_xor_b32 %0, %0, %1"
#define A_LSHL(n) "v_lshlrev_b32 %0, 3, %0"
#define A_LSHR(n) "v_lshrrev_b32 %0, 3, %0"
#define A_ASHR(n) "v_ashrrev_i32 %0, 3, %0"
#define A_MAXU(n) "v_max_u32 %0, %0, %1"
#define A_MAXI(n) "v_max_i32 %0, %0, %1"
#define A_MINI(n) "v_min_i32 %0, %0, %1"
#define A_ADDF(n) "v_add_f32 %0, %0, %1"
#define A_SUBF(n) "v_sub_f32 %0, %0, %1"
#define A_MULF(n) "v_mul_f32 %0, %0, %1"
#define A_MAXF(n) "v_max_f32 %0, %0, %1"
#define A_MINF(n) "v_min_f32 %0, %0, %1"
#define A_ADDFABS(n) "v_add_f32 %0, %0, |%1|"
#define A_FMAC(n) "v_fmac_f32 %0, %1, %2"
#define A_MIN3F(n) "v_min3_f32 %0, %0, %1, %2"
#define A_MED3F(n) "v_med3_f32 %0, %0, %1, %2"
#define A_FLOORF(n) "v_floor_f32 %0, %0"
#define A_FRACTF(n) "v_fract_f32 %0, %0"
#define A_CVTU(n) "v_cvt_u32_f32 %0, %0"
#define A_CVTF(n) "v_cvt_f32_u32 %0, %0"
#define A_CVTUB0(n) "v_cvt_f32_ubyte0 %0, %1"
#define A_CVTUB2(n) "v_cvt_f32_ubyte2 %0, %1"
#define A_CVTPKU8(n) "v_cvt_pk_u8_f32 %0, %1, 1, %0"
#define A_ADDU16(n) "v_add_u16 %0, %0, %1"
#define A_SUBU16(n) "v_sub_u16 %0, %0, %1"
#define A_MAXU16(n) "v_max_u16 %0, %0, %1"
#define A_MINU16(n) "v_min_u16 %0, %0, %1"
#define A_LSHRB16(n) "v_lshrrev_b16 %0, 4, %0"
#define A_MULLOU16(n) "v_mul_lo_u16 %0, %0, %1"
#define A_MADU16(n) "v_mad_u16 %0, %0, %1, %2"
#define A_CNDS(n) "v_cndmask_b32 %0, %0, %1, s[10:11]"
#define A_ADDCO(n) "v_add_co_u32 %0, vcc, %0, %1"
#define A_LSHLADD(n) "v_lshl_add_u32 %0, %0, 2, %1"
#define A_ADDLSHL(n) "v_add_lshl_u32 %0, %0, %1, 2"
#define A_BFI(n) "v_bfi_b32 %0, %0, %1, %2"
#define A_ALIGNBIT(n) "v_alignbit_b32 %0, %0, %1, 16"
#define A_PKMIN(n) "v_pk_min_u16 %0, %0, %1"
#define A_PKADDF16(n) "v_pk_add_f16 %0, %0, %1"
#define A_PKFMAF16(n) "v_pk_fma_f16 %0, %0, %1, %2"
#define A_PKADDF32(n) "v_pk_add_f32 %0, %0, %1"
#define A_XAD(n) "v_xad_u32 %0, %0, %1, %2"
#define A_MAX3U(n) "v_max3_u32 %0, %0, %1, %2"
#define A_READLANE(n) "v_readlane_b32 s12, %0, 3"
#define A_MBCNT(n) "v_mbcnt_lo_u32_b32 %0, %1, %0"
#define A_ADDF16(n) "v_add_f16 %0, %0, %1"
#define A_MAXF16(n) "v_max_f16 %0, %0, %1"
#define A_CVTF16(n) "v_cvt_f16_f32 %0, %0"
#define A_ADDE64(n) "v_add_u32_e64 %0, %0, %1"
#define A_ANDE64(n) "v_and_b32_e64 %0, %0, %1"

DEFK(sub, A_SUB) DEFK(subrev, A_SUBREV) DEFK(or_b32, A_OR) DEFK(xor_b32, A_XOR) DEFK(lshl, A_LSHL) DEFK(lshr, A_LSHR) DEFK(ashr, A_ASHR) DEFK(max_u32, A_MAXU) DEFK(max_i32, A_MAXI) DEFK(min_i32, A_MINI) DEFK(add_f32, A_ADDF) DEFK(sub_f32, A_SUBF) DEFK(mul_f32, A_MULF) DEFK(max_f32, A_MAXF) DEFK(min_f32, A_MINF) DEFK(add_f32_abs, A_ADDFABS) DEFK(fmac_f32, A_FMAC) DEFK(min3_f32, A_MIN3F) DEFK(med3_f32, A_MED3F) DEFK(floor_f32, A_FLOORF) DEFK(fract_f32, A_FRACTF) DEFK(cvt_u32_f32, A_CVTU) DEFK(cvt_f32_u32, A_CVTF) DEFK(cvt_f32_ubyte0, A_CVTUB0) DEFK(cvt_f32_ubyte2, A_CVTUB2) DEFK(cvt_pk_u8_f32, A_CVTPKU8) DEFK(add_u16, A_ADDU16) DEFK(sub_u16, A_SUBU16) DEFK(max_u16, A_MAXU16) DEFK(min_u16, A_MINU16) DEFK(lshr_b16, A_LSHRB16) DEFK(mul_lo_u16, A_MULLOU16) DEFK(mad_u16, A_MADU16) DEFK(cndmask_sgpr, A_CNDS) DEFK(add_co, A_ADDCO) DEFK(lshl_add, A_LSHLADD) DEFK(add_lshl, A_ADDLSHL) DEFK(bfi, A_BFI) DEFK(alignbit, A_ALIGNBIT) DEFK(pk_min_u16, A_PKMIN) DEFK(pk_add_f16, A_PKADDF16) DEFK(pk_fma_f16, A_PKFMAF16) DEFK(xad, A_XAD) DEFK(max3_u32, A_MAX3U) DEFK(mbcnt, A_MBCNT) DEFK(add_f16, A_ADDF16) DEFK(max_f16, A_MAXF16) DEFK(cvt_f16_f32, A_CVTF16) DEFK(add_e64, A_ADDE64) DEFK(and_e64, A_ANDE64)

// 64-bit result forms (qsad): separate kernel shape
__global__ void __launch_bounds__(1024) k_qsad(unsigned* out, int iters)
{
    unsigned long long r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, s = r0 * 7 + 3;
    unsigned b = threadIdx.x * 5 + 2;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            asm volatile("v_qsad_pk_u16_u8 %0, %1, %2, %0" : "+v"(r0) : "v"(s), "v"(b));
            asm volatile("v_qsad_pk_u16_u8 %0, %1, %2, %0" : "+v"(r1) : "v"(s), "v"(b));
            asm volatile("v_qsad_pk_u16_u8 %0, %1, %2, %0" : "+v"(r2) : "v"(s), "v"(b));
            asm volatile("v_qsad_pk_u16_u8 %0, %1, %2, %0" : "+v"(r3) : "v"(s), "v"(b));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (unsigned)(r0 ^ r1 ^ r2 ^ r3);
}

typedef void (*kern_t)(unsigned*, int);
struct Entry { const char* name; kern_t k; int per_iter; };

int main(int argc, char** argv)
{
    Entry es[] = {
#define E(n) {#n, k_##n, 64},
        E(add) E(add3) E(sad_u16) E(bfe) E(lshl_or) E(min) E(min3) E(add_dpp_wave) E(add_dpp_row) E(mov_dpp_wave)
        E(add_sdwa) E(pk_add_u16) E(pk_sub_opsel) E(pk_max_i16) E(cndmask) E(mul_u24) E(mad_u24) E(and_b32) E(and_or)
        E(perm) E(alignbyte) E(lerp_u8) E(sad_u8) E(fma_f32) E(mov) E(pk_mad_u16) E(pk_lshr) E(dot4_u8) E(msad_u8)
        E(sub) E(subrev) E(or_b32) E(xor_b32) E(lshl) E(lshr) E(ashr) E(max_u32) E(max_i32) E(min_i32) E(add_f32) E(sub_f32) E(mul_f32) E(max_f32) E(min_f32) E(add_f32_abs) E(fmac_f32) E(min3_f32) E(med3_f32) E(floor_f32) E(fract_f32) E(cvt_u32_f32) E(cvt_f32_u32) E(cvt_f32_ubyte0) E(cvt_f32_ubyte2) E(cvt_pk_u8_f32) E(add_u16) E(sub_u16) E(max_u16) E(min_u16) E(lshr_b16) E(mul_lo_u16) E(mad_u16) E(cndmask_sgpr) E(add_co) E(lshl_add) E(add_lshl) E(bfi) E(alignbit) E(pk_min_u16) E(pk_add_f16) E(pk_fma_f16) E(xad) E(max3_u32) E(mbcnt) E(add_f16) E(max_f16) E(cvt_f16_f32) E(add_e64) E(and_e64)
        {"qsad_pk_u16_u8", k_qsad, 64},
    };
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const double ghz = prop.clockRate * 1e-6;
    unsigned* out;
    hipMalloc(&out, (size_t)cus * 1024 * 4 * sizeof(unsigned));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 4000;
    printf("# %s, %d CUs, clock %.2f GHz; cycles per wave-instruction per SIMD (nominal clock)\n", prop.name, cus, ghz);
    printf("%-18s %8s %8s %8s\n", "instruction", "1w/SIMD", "2w/SIMD", "4w/SIMD");
    for (auto& e : es) {
        printf("%-18s", e.name);
        for (int wps : {1, 2, 4}) {
            const int threads = wps * 4 * 64;  // waves per CU = wps * 4 SIMDs, one block per CU
            hipLaunchKernelGGL(e.k, dim3(cus), dim3(threads), 0, 0, out, 10);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(e.k, dim3(cus), dim3(threads), 0, 0, out, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double instr_per_simd = (double)wps * iters * e.per_iter;
            printf(" %8.2f", ms * 1e-3 * ghz * 1e9 / instr_per_simd);
        }
        printf("\n");
    }
    return 0;
}
