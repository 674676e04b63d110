// sn_fused_u8_parts.h -- the pieces of the 8-bit fused sweeps that do not depend on the sweep's mode: a kept line in
// packed form (two pixels per register), stage 1's operands, the 7-tap box with the line buffer's clamps, and stage 3 in
// the byte domain.  Shared by sn_fused_u8_v3.hip (a register's halves = two column strips of one plane) and
// sn_fused_u8_uv.hip (a register's halves = the same strip of the U pass and of the V pass of a 4:2:0 frame).
// Reference semantics: /root/reference/src/SangNom2.cpp:25-34 (loadPixel), :60-65 (calculateSangNom), :74-124
// (prepareBuffers_c), :144-150 (the box of processBuffers_c), :204-249 (finalizePlane_c's ladder).
#pragma once

#include <type_traits>

#include "sn_fused_v3_common.h"

namespace sn {
namespace v3 {

using namespace v3c;

constexpr unsigned kLo = 0x0000ffffu, kHi = 0xffff0000u, kByte = 0x00ff00ffu;

// One kept line: P[i] = pixel (x0 - 3 + i) of both strips, F / B = the two SangNom values per pixel
// (calculateSangNom, SangNom2.cpp:60-65), all packed lo | hi << 16.
struct Line {
    unsigned P[PXL + 6];
    unsigned FB[PXL];  // F | B << 8 in each 16-bit half (both are 8-bit values)
    __device__ __forceinline__ unsigned F(int j) const { return FB[j] & kByte; }
    __device__ __forceinline__ unsigned B(int j) const { return pk_lshr8(FB[j]); }  // F < 256: one packed shift, no mask
};
// The same with F and B in registers of their own: eight registers more per line, no extraction per use.  For the
// sweeps whose stage 3 runs in the byte domain: there a line in this form only serves stage 1, two of them are live,
// and the kernel has the registers.
struct WideLine {
    unsigned P[PXL + 6];
    unsigned Fv[PXL], Bv[PXL];
    __device__ __forceinline__ unsigned F(int j) const { return Fv[j]; }
    __device__ __forceinline__ unsigned B(int j) const { return Bv[j]; }
};

struct RawHalf {  // left dword, own 8 bytes, right dword of one strip
    uint32_t l, m0, m1, r;
};
struct Raw {
    RawHalf h[2];
};


// One 16-byte buffer load per strip and line: bytes [x0 - 4, x0 + 12) = left dword, own 8 bytes,
// right dword.  Row base in soffset (wave-uniform), column in voffset; dead lanes carry an
// out-of-range voffset and read zeros, so there is no branch around memory operations.
// The lane that owns column 0 loads from column 0 instead (unpack() shifts its dwords).
__device__ __forceinline__ RawHalf load_half(__amdgpu_buffer_rsrc_t rs, int voff, int soff)
{
    const u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
    return RawHalf{q.x, q.y, q.z, q.w};
}

struct LaneRole {
    bool edge_wave;      // wave holds column 0 or column w-1: clamps needed
    unsigned first_mask; // 0xffff in the half that owns column 0 (else 0)
    unsigned last_mask;  // 0xffff (shifted) in the half that owns column w-1 of the sweep
    unsigned line_last_mask;  // ... that owns the last column of the source lines (kChroma: region_w - 1)
    unsigned inside_mask;     // kChroma: halves whose columns lie inside the chroma region
    unsigned key_mask;        // 0x0ff00ff0 in a VGPR (operand of the and-or that forms the ladder keys)
};

// byte k of the lo word -> bits 0..7, byte k of the hi word -> bits 16..23
__device__ __forceinline__ unsigned pair_byte(uint32_t hi_word, uint32_t lo_word, int k)
{
    return __builtin_amdgcn_perm(hi_word, lo_word, 0x0c040c00u + (unsigned)k * 0x00010001u);
}

// loadPixel's clamp (SangNom2.cpp:25-34) for the two image-edge lanes: afterwards l / m0 m1 / r are the four pixels
// left of the lane's eight, the eight, and the four right of them in every lane
__device__ __forceinline__ Raw clamp_edges(Raw q, const LaneRole& role)
{
    if (role.edge_wave) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (role.first_mask & (h ? kHi : kLo)) {  // loaded from column 0: dwords are one slot early
                q.h[h].r = q.h[h].m1;
                q.h[h].m1 = q.h[h].m0;
                q.h[h].m0 = q.h[h].l;
                q.h[h].l = (q.h[h].m0 & 0xff) * 0x01010101u;
            }
            if (role.line_last_mask & (h ? kHi : kLo)) q.h[h].r = (q.h[h].m1 >> 24) * 0x01010101u;
        }
    }
    return q;
}

// q: after clamp_edges()
template <class LineT>
__device__ __forceinline__ void unpack(LineT& L, const Raw& q)
{
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        L.P[k] = pair_byte(q.h[1].l, q.h[0].l, k + 1);
        L.P[PXL + 3 + k] = pair_byte(q.h[1].r, q.h[0].r, k);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        L.P[3 + k] = pair_byte(q.h[1].m0, q.h[0].m0, k);
        L.P[7 + k] = pair_byte(q.h[1].m1, q.h[0].m1, k);
    }
    // F = ((4a + 5b - c) >> 3) mod 256, B = ((4c + 5b - a) >> 3) mod 256 with a bias of 2048 per half
    unsigned Q4[PXL + 2], M[PXL + 2];  // positions 2 .. PXL+3
#pragma unroll
    for (int i = 0; i < PXL + 2; ++i) {
        const unsigned p = L.P[i + 2];
        const unsigned p2 = p + p;
        Q4[i] = p2 + p2;
        M[i] = 0x08000800u - p;
    }
#pragma unroll
    for (int j = 0; j < PXL; ++j) {
        const unsigned q5 = Q4[j + 1] + L.P[j + 3];
        const unsigned f = ((Q4[j] + q5 + M[j + 2]) >> 3) & kByte;
        if constexpr (std::is_same<LineT, WideLine>::value) {
            L.Fv[j] = f;
            L.Bv[j] = ((Q4[j + 2] + q5 + M[j]) >> 3) & kByte;
        } else {
            const unsigned b = ((Q4[j + 2] + q5 + M[j]) << 5) & 0xff00ff00u;  // (x >> 3 & 255) << 8
            L.FB[j] = f | b;
        }
    }
}

// Stage 1, buffer BUF, packed position j, pair (c, n): the two values whose difference is the cost; Buffers enum of
// SangNom2.h:8-20.
template <int BUF, class LineT>
__device__ __forceinline__ void cost_operands(const LineT& c, const LineT& n, int j, unsigned& x, unsigned& y)
{
    const int i = j + 3;
    if constexpr (BUF == 0) { x = c.P[i - 3]; y = n.P[i + 3]; }
    else if constexpr (BUF == 1) { x = c.P[i - 2]; y = n.P[i + 2]; }
    else if constexpr (BUF == 2) { x = c.P[i - 1]; y = n.P[i + 1]; }
    else if constexpr (BUF == 3) { x = c.F(j); y = n.B(j); }  // |forwardSangNom1 - forwardSangNom2|
    else if constexpr (BUF == 4) { x = c.P[i]; y = n.P[i]; }
    else if constexpr (BUF == 5) { x = c.B(j); y = n.F(j); }  // |backwardSangNom1 - backwardSangNom2|
    else if constexpr (BUF == 6) { x = c.P[i + 1]; y = n.P[i - 1]; }
    else if constexpr (BUF == 7) { x = c.P[i + 2]; y = n.P[i - 2]; }
    else { x = c.P[i + 3]; y = n.P[i - 3]; }
}
template <int BUF, class LineT>
__device__ __forceinline__ unsigned cost(const LineT& c, const LineT& n, int j)
{
    unsigned x, y;
    cost_operands<BUF>(c, n, j, x, y);
    return pk_absdiff(x, y);
}

// ---- Stage 3 in the byte domain ---------------------------------------------------------------
// The ladder's winner picks, per pixel, ONE byte of the upper line and ONE byte of the lower line (SangNom2.cpp:214-249:
// c(k) and n(-k) for k = -3 .. 3, or the two SangNom values), and the result is their rounded average.  In the packed
// pair layout that is nine tap sums and an eight-select tree per register (31 instructions per pixel pair).  On the
// lines as they lie in memory -- four pixels per dword -- it is a handful of byte permutes whose SELECTORS are data:
//   * the rank code of the winner (low nibble of the minimum key) indexes a byte table through v_perm_b32
//     (selector 0..7 -> table byte, 12 -> 0x00): sel_c = 3 + k;  the lower line's selector is 6 - sel_c;
//   * two neighbouring pixels x, x+1 find all their candidates c(x-3) .. c(x+4) in ONE 8-byte window, so a v_perm_b32
//     over that window with selectors (3 + k0, 4 + k1) fetches both; the windows of a lane's eight pixels are six
//     dwords per strip, cut with v_alignbyte_b32 once per line and kept in LDS for the two rows that use the line;
//   * the SangNom candidates are two more byte permutes (take the F / B byte where the code says so);
//   * v_lerp_u8 with 0x01010101 is (a + b + 1) >> 1 on four pixels at once.
// 18 instructions per four pixels of a strip instead of 124, and the result is already in memory order.
struct RawLine {
    unsigned W[2][6];  // per strip: [p-3..p0] [p1..p4] [p-1..p2] [p3..p6] [p5..p8] [p7..p10] (p0 = the lane's first pixel)
    unsigned F[2][2], B[2][2];  // forward / backward SangNom values of p0..p3 and p4..p7, one byte each
};
// rank codes: P4 (and the `minBuf > aaf` arm, same result) 0, P5 1, P3 2, P6 3, P2 4, P7 5, P1 6, P8 7, P0 12
constexpr unsigned kLutLo = 0x04030303u, kLutHi = 0x06010502u;  // code -> 3 + k; code 12 reads 0x00 = 3 + (-3)

template <class LineT>
__device__ __forceinline__ void make_raw(RawLine& R, const Raw& q, const LineT& L)
{
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const RawHalf& x = q.h[h];
        R.W[h][0] = __builtin_amdgcn_alignbyte(x.m0, x.l, 1);
        R.W[h][1] = __builtin_amdgcn_alignbyte(x.m1, x.m0, 1);
        R.W[h][2] = __builtin_amdgcn_alignbyte(x.m0, x.l, 3);
        R.W[h][3] = __builtin_amdgcn_alignbyte(x.m1, x.m0, 3);
        R.W[h][4] = __builtin_amdgcn_alignbyte(x.r, x.m1, 1);
        R.W[h][5] = __builtin_amdgcn_alignbyte(x.r, x.m1, 3);
    }
    // F[j] = [lo strip, 0, hi strip, 0]: pixels j, j+1 side by side, then the four of a strip into one dword
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const unsigned f01 = L.F(4 * g + 0) | (L.F(4 * g + 1) << 8), f23 = L.F(4 * g + 2) | (L.F(4 * g + 3) << 8);  // [lo0 lo1 hi0 hi1]
        const unsigned b01 = L.B(4 * g + 0) | (L.B(4 * g + 1) << 8), b23 = L.B(4 * g + 2) | (L.B(4 * g + 3) << 8);
        R.F[0][g] = __builtin_amdgcn_perm(f23, f01, 0x05040100u);
        R.F[1][g] = __builtin_amdgcn_perm(f23, f01, 0x07060302u);
        R.B[0][g] = __builtin_amdgcn_perm(b23, b01, 0x05040100u);
        R.B[1][g] = __builtin_amdgcn_perm(b23, b01, 0x07060302u);
    }
}

// four interpolated pixels of strip h, group g (pixels 4g .. 4g+3), from the rank codes of their winners
__device__ __forceinline__ unsigned interpolate4(const RawLine& c, const RawLine& n, int h, int g, unsigned codes)
{
    const unsigned sc = __builtin_amdgcn_perm(kLutHi, kLutLo, codes) + 0x01000100u;  // [3 + k0, 4 + k1, 3 + k2, 4 + k3]
    const unsigned sn = 0x08060806u - sc;                                          // [3 - k0, 4 - k1, 3 - k2, 4 - k3]
    const int a = g == 0 ? 0 : 1, b = g == 0 ? 1 : 4;  // window of pixels 0, 1 of the group: dwords W[a], W[b]
    const int d = g == 0 ? 2 : 3, e = g == 0 ? 3 : 5;  // ... of pixels 2, 3
    unsigned cc = bfi(0x0000ffffu, __builtin_amdgcn_perm(c.W[h][b], c.W[h][a], sc), __builtin_amdgcn_perm(c.W[h][e], c.W[h][d], sc));
    unsigned nn = bfi(0x0000ffffu, __builtin_amdgcn_perm(n.W[h][b], n.W[h][a], sn), __builtin_amdgcn_perm(n.W[h][e], n.W[h][d], sn));
    // P5 (code 1): avg(backwardSangNom1, backwardSangNom2) = c.B, n.F;  P3 (code 2): avg(forward1, forward2) = c.F, n.B
    const unsigned sb = __builtin_amdgcn_perm(0u, 0x00000400u, codes) + 0x03020100u;  // byte i: i, or 4 + i where the code is 1
    const unsigned sf = __builtin_amdgcn_perm(0u, 0x00040000u, codes) + 0x03020100u;  // ... where the code is 2
    cc = __builtin_amdgcn_perm(c.B[h][g], cc, sb);
    nn = __builtin_amdgcn_perm(n.F[h][g], nn, sb);
    cc = __builtin_amdgcn_perm(c.F[h][g], cc, sf);
    nn = __builtin_amdgcn_perm(n.B[h][g], nn, sf);
    return __builtin_amdgcn_lerp(cc, nn, 0x01010101u);  // (a + b + 1) >> 1 per byte, SangNom2.cpp:48-52
}

// the code of buffer BUF in the ladder keys (see RawLine): the reference's order P4, P5, P3, P6, P2, P7, P1, P8, P0
// (SangNom2.cpp:214-249) as 0, 1, 2, 3, 4, 5, 6, 7, 12 -- smaller wins a tie
template <int BUF, int MODE>
constexpr unsigned rank_of()
{
    constexpr unsigned code[9] = {12, 6, 4, 2, 0, 1, 3, 5, 7};
    return code[BUF] * 0x00010001u;
}

// The 7-tap box over S with the line buffer's clamps (SangNom2.cpp:144-150), the same instructions in every wave:
//   * column 0 is lane 0 of the first strip, and lane 0 has no left neighbour: its DPP move keeps the `old` operand,
//     S[0] -- the clamp.  (Lane 0 of every other strip is the outermost ghost lane, whose value is wrong by design.)
//   * the last column sits in some lane of the last strip: a select per right-hand tap there (RCLAMP; every wave runs it
//     unless SN_RCLAMP_BRANCH asks for a wave-uniform branch around it).
// Round 2 branched on "this wave holds an image edge" around two whole variants of the box: the edge waves paid 29
// instructions per buffer instead of 20, the branch cut every buffer step into three scheduling regions, and the seam
// refresh made every wave wait for the slowest (knocking the edge variant out -- wrong at the edges -- ran 12 % faster).
template <bool RCLAMP>
__device__ __forceinline__ void box7(const unsigned (&S)[PXL], unsigned (&Bx)[PXL], const LaneRole& role)
{
    unsigned L[3], R[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        L[k] = dpp_from_left_or(S[0], S[PXL - 3 + k]);
        R[k] = dpp_from_right(S[k]);
        if constexpr (RCLAMP) R[k] = bfi(role.last_mask, S[PXL - 1], R[k]);  // clamp to column w-1
    }
    auto X = [&](int i) -> unsigned { return i < 0 ? L[i + 3] : i >= PXL ? R[i - PXL] : S[i]; };
    Bx[0] = S[0] + S[1] + S[2] + S[3] + X(-1) + X(-2) + X(-3);
#pragma unroll
    for (int j = 0; j + 1 < PXL; ++j) Bx[j + 1] = Bx[j] - X(j - 3) + X(j + 4);
}
#ifndef SN_RCLAMP_BRANCH
#define SN_RCLAMP_BRANCH 0
#endif
__device__ __forceinline__ void box7_any(const unsigned (&S)[PXL], unsigned (&Bx)[PXL], const LaneRole& role)
{
    if (SN_RCLAMP_BRANCH && !role.edge_wave) box7<false>(S, Bx, role);
    else box7<true>(S, Bx, role);
}

// A RawLine parked in LDS for the two rows that use it: five uint4 per thread and line, `nthreads` apart (each thread
// reads back exactly what it wrote, so no barrier is involved); `v` = the line's slot.
__device__ __forceinline__ void park_raw_at(uint4* v, int nthreads, int tid, const RawLine& R)
{
    uint4* to = v + tid;
    to[0 * nthreads] = make_uint4(R.W[0][0], R.W[0][1], R.W[0][2], R.W[0][3]);
    to[1 * nthreads] = make_uint4(R.W[0][4], R.W[0][5], R.W[1][0], R.W[1][1]);
    to[2 * nthreads] = make_uint4(R.W[1][2], R.W[1][3], R.W[1][4], R.W[1][5]);
    to[3 * nthreads] = make_uint4(R.F[0][0], R.F[0][1], R.B[0][0], R.B[0][1]);
    to[4 * nthreads] = make_uint4(R.F[1][0], R.F[1][1], R.B[1][0], R.B[1][1]);
}
__device__ __forceinline__ void unpark_raw_at(const uint4* v, int nthreads, int tid, RawLine& R)
{
    const uint4* from = v + tid;
    const uint4 a = from[0 * nthreads], b = from[1 * nthreads], c = from[2 * nthreads], d = from[3 * nthreads], e = from[4 * nthreads];
    R.W[0][0] = a.x; R.W[0][1] = a.y; R.W[0][2] = a.z; R.W[0][3] = a.w;
    R.W[0][4] = b.x; R.W[0][5] = b.y; R.W[1][0] = b.z; R.W[1][1] = b.w;
    R.W[1][2] = c.x; R.W[1][3] = c.y; R.W[1][4] = c.z; R.W[1][5] = c.w;
    R.F[0][0] = d.x; R.F[0][1] = d.y; R.B[0][0] = d.z; R.B[0][1] = d.w;
    R.F[1][0] = e.x; R.F[1][1] = e.y; R.B[1][0] = e.z; R.B[1][1] = e.w;
}

struct Out {
    uint32_t lo[2], hi[2];  // 8 interpolated bytes of each half's strip
};

}  // namespace v3
}  // namespace sn
