// sn_fused_u8.hip -- the fused 8-bit kernel: frame assembly + stage 1 + stage 2 + stage 3 of one
// plane in ONE pass, with the nine cost buffers never leaving registers.
//
// Reference semantics: /root/reference/src/SangNom2.cpp:74-124 (prepareBuffers_c), :126-159
// (processBuffers_c), :161-257 (finalizePlane_c), :361-391 (GetFrame's copies).
//
// Mapping (DESIGN.md "Fused kernel"):
//   * one workgroup per (frame, plane); the workgroup sweeps the plane top to bottom because
//     stage 2 is a vertical recurrence  O[r] = box7(O[r-1] + D[r] + D[r+1]) / 16 (mod 256);
//   * a lane owns PXL (4 or 8) consecutive pixels of every row and keeps, per cost buffer, only
//     A[r] = O[r-1] + D[r], two pixels per register:  S[r] = A[r] + D[r+1],
//     O[r] = (box7(S[r]) >> 4) & 255,  A[r+1] = O[r] + D[r+1];
//   * the +-3 horizontal taps of the 7-tap box come from the neighbouring lanes with DPP
//     wave_shr:1 / wave_shl:1 folded into the adds (wavefront shuffles, no LDS);
//   * the first / last GH lanes of a wave are "ghost" lanes: they recompute everything for the
//     columns the neighbouring wave owns.  A ghost zone of G = GH * PXL pixels stays exact in
//     its innermost 3 pixels for floor(G / 3) rows (the missing outer neighbour corrupts 3 more
//     pixels per row), so the waves of a workgroup only meet every K = floor(G / 3) rows: the
//     seam lanes publish their whole A state to an LDS mailbox, one s_barrier, the ghosts reload
//     it.  The sweep is exact across wave seams (no speculation);
//   * image edges clamp S to the first / last column (loadPixel on the line buffer,
//     SangNom2.cpp:144-150): handled by per-lane selects in the two edge waves only;
//   * stage 3's priority ladder is a single unsigned minimum over keys
//     (O << 16) | (rank << 12) | (a + b + 1): smallest cost wins, ties go to the reference's
//     priority order, and the winner's average is (key & 0x1ff) >> 1.  The `minBuf > aaf` test
//     is a tenth key with cost aaf + 1 and rank 0 that selects avg(c0, n0).
//
// Everything is integer; results are bit-exact to the pool path and to the opt=0 reference.
#include <stdlib.h>

#include <type_traits>

#include "sn_internal.h"

namespace sn {

namespace {

struct FusedArgs {
    const uint8_t* src;
    uint8_t* dst;
    int64_t src_frame_stride;
    int64_t dst_frame_stride;
    int32_t src_pitch;
    int32_t dst_pitch;
    int32_t w;       // pixels, multiple of 32
    int32_t nk;      // kept lines = h_out / 2
    int32_t offset;  // first kept line in dst
    int32_t dh;      // source line k is src row k (dh) or src row offset + 2k
    int32_t thr;     // aaf as integer
    int32_t nl;      // real lanes = w / PXL
    int32_t nw;      // waves per workgroup
    int32_t dbg;     // timing experiments only (SN_FUSED_DEBUG): 1 = no seam refresh (wrong results)
};

typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int dpp_from_left(int v)  // lane i receives lane i-1 (0 for lane 0)
{
    return __builtin_amdgcn_update_dpp(0, v, 0x138 /* wave_shr:1 */, 0xf, 0xf, true);
}
__device__ __forceinline__ int dpp_from_right(int v)  // lane i receives lane i+1 (0 for lane 63)
{
    return __builtin_amdgcn_update_dpp(0, v, 0x130 /* wave_shl:1 */, 0xf, 0xf, true);
}
__device__ __forceinline__ int ubfe(unsigned v, int off, int width) { return (int)__builtin_amdgcn_ubfe(v, off, width); }
__device__ __forceinline__ int sad(int a, int b) { return (int)__builtin_amdgcn_sad_u16((unsigned)a, (unsigned)b, 0u); }

// One kept line as a lane sees it: its PXL pixels plus three on each side, and the two SangNom
// values per pixel (calculateSangNom, SangNom2.cpp:60-65), packed F | B << 16:
//   F[j] = sg(p[-1], p[0], p[+1])   B[j] = sg(p[+1], p[0], p[-1])
// For a pair (c, n): forwardSangNom1 = F(c), forwardSangNom2 = B(n), backwardSangNom1 = B(c),
// backwardSangNom2 = F(n)  (SangNom2.cpp:100-103).
template <int PXL>
struct Line {
    int p[PXL + 6];  // p[i] = pixel x0 - 3 + i (edge-clamped)
    unsigned FB[PXL];
};

// A line as loaded: left dword (x0-4..x0-1), own PXL bytes, right dword (x0+PXL..x0+PXL+3).
template <int PXL>
struct Raw {
    uint32_t l, r;
    uint32_t m[PXL / 4];
};

template <int PXL>
__device__ __forceinline__ Raw<PXL> load_raw(const uint8_t* row, int x0, int w, bool live)
{
    Raw<PXL> q;
    q.l = q.r = 0;
#pragma unroll
    for (int i = 0; i < PXL / 4; ++i) q.m[i] = 0;
    if (live) {
        if constexpr (PXL == 8) {
            const uint2 m = *reinterpret_cast<const uint2*>(row + x0);
            q.m[0] = m.x;
            q.m[1] = m.y;
        } else {
            q.m[0] = *reinterpret_cast<const uint32_t*>(row + x0);
        }
        // the image-edge lanes have no left / right dword: unpack() replicates their edge pixel
        // (doing it here would make the edge waves wait for the load they have just issued)
        if (x0 > 0) q.l = *reinterpret_cast<const uint32_t*>(row + x0 - 4);
        if (x0 + PXL < w) q.r = *reinterpret_cast<const uint32_t*>(row + x0 + PXL);
    }
    return q;
}

template <int PXL>
__device__ __forceinline__ void store_own(uint8_t* row, int x0, const Raw<PXL>& q)
{
    if constexpr (PXL == 8) *reinterpret_cast<uint2*>(row + x0) = make_uint2(q.m[0], q.m[1]);
    else *reinterpret_cast<uint32_t*>(row + x0) = q.m[0];
}

struct LaneRole {
    bool edge_wave;   // wave holds the first or the last real lane: box needs the clamp selects
    bool first_real;  // lane owns column 0
    bool last_real;   // lane owns column w - 1
};

template <int PXL>
__device__ __forceinline__ void unpack(Line<PXL>& L, Raw<PXL> q, const LaneRole& role)
{
    if (role.edge_wave) {  // loadPixel's clamp (SangNom2.cpp:25-34) for the two image-edge lanes
        if (role.first_real) q.l = (q.m[0] & 0xff) * 0x01010101u;
        if (role.last_real) q.r = (q.m[PXL / 4 - 1] >> 24) * 0x01010101u;
    }
    L.p[0] = ubfe(q.l, 8, 8);
    L.p[1] = ubfe(q.l, 16, 8);
    L.p[2] = (int)(q.l >> 24);
#pragma unroll
    for (int i = 0; i < PXL / 4; ++i) {
        L.p[3 + 4 * i] = (int)(q.m[i] & 0xff);
        L.p[4 + 4 * i] = ubfe(q.m[i], 8, 8);
        L.p[5 + 4 * i] = ubfe(q.m[i], 16, 8);
        L.p[6 + 4 * i] = (int)(q.m[i] >> 24);
    }
    L.p[PXL + 3] = (int)(q.r & 0xff);
    L.p[PXL + 4] = ubfe(q.r, 8, 8);
    L.p[PXL + 5] = ubfe(q.r, 16, 8);
#pragma unroll
    for (int j = 0; j < PXL; ++j) {
        const int a = L.p[j + 2], b = L.p[j + 3], c = L.p[j + 4];
        const int x5 = b * 5;
        const unsigned F = (unsigned)ubfe((unsigned)(4 * a + x5 - c), 3, 8);  // arithmetic >> 3, wrap to 8 bits
        const unsigned B = (unsigned)ubfe((unsigned)(4 * c + x5 - a), 3, 8);
        L.FB[j] = F | (B << 16);
    }
}

// Stage 1 for buffer BUF of pixel j of the pair (c, n) -- the Buffers enum order of
// /root/reference/src/SangNom2.h:8-20.  Buffers 3 / 5 (SangNom costs) come from sg_costs().
template <int BUF, int PXL>
__device__ __forceinline__ int cost(const Line<PXL>& c, const Line<PXL>& n, int j)
{
    const int i = j + 3;
    if constexpr (BUF == 0) return sad(c.p[i - 3], n.p[i + 3]);  // ADIFF_M3_P3
    if constexpr (BUF == 1) return sad(c.p[i - 2], n.p[i + 2]);  // ADIFF_M2_P2
    if constexpr (BUF == 2) return sad(c.p[i - 1], n.p[i + 1]);  // ADIFF_M1_P1
    if constexpr (BUF == 4) return sad(c.p[i], n.p[i]);          // ADIFF_P0_M0
    if constexpr (BUF == 6) return sad(c.p[i + 1], n.p[i - 1]);  // ADIFF_P1_M1
    if constexpr (BUF == 7) return sad(c.p[i + 2], n.p[i - 2]);  // ADIFF_P2_M2
    if constexpr (BUF == 8) return sad(c.p[i + 3], n.p[i - 3]);  // ADIFF_P3_M3
    return 0;
}

// Stage 3 candidate of buffer BUF: (rank << 12) + a + b + 1, rank = position in the reference's
// if/else ladder (SangNom2.cpp:214-249): P4, P5, P3, P6, P2, P7, P1, P8, P0 -> 1..9.
// Buffers 3 / 5 add the halves of sg_sums().
template <int BUF, int PXL>
__device__ __forceinline__ int candidate(const Line<PXL>& c, const Line<PXL>& n, int j)
{
    const int i = j + 3;
    if constexpr (BUF == 0) return c.p[i - 3] + n.p[i + 3] + ((9 << 12) + 1);
    if constexpr (BUF == 1) return c.p[i - 2] + n.p[i + 2] + ((7 << 12) + 1);
    if constexpr (BUF == 2) return c.p[i - 1] + n.p[i + 1] + ((5 << 12) + 1);
    if constexpr (BUF == 3) return (3 << 12) + 1;
    if constexpr (BUF == 4) return c.p[i] + n.p[i] + ((1 << 12) + 1);
    if constexpr (BUF == 5) return (2 << 12) + 1;
    if constexpr (BUF == 6) return c.p[i + 1] + n.p[i - 1] + ((4 << 12) + 1);
    if constexpr (BUF == 7) return c.p[i + 2] + n.p[i - 2] + ((6 << 12) + 1);
    return c.p[i + 3] + n.p[i - 3] + ((8 << 12) + 1);
}

// |F(c) - B(n)| | |B(c) - F(n)| << 16: the SG_FORWARD / SG_REVERSE costs of one pixel, two at a time
// with packed 16-bit math (op_sel swaps n's halves for free).
__device__ __forceinline__ unsigned sg_costs(unsigned fb_c, unsigned fb_n)
{
    const s16x2 a = __builtin_bit_cast(s16x2, fb_c), b = __builtin_bit_cast(s16x2, fb_n);
    const s16x2 d = a - __builtin_shufflevector(b, b, 1, 0);
    return __builtin_bit_cast(unsigned, __builtin_elementwise_abs(d));
}
// (F(c) + B(n)) | (B(c) + F(n)) << 16: the two SangNom averages' sums.
__device__ __forceinline__ unsigned sg_sums(unsigned fb_c, unsigned fb_n)
{
    const u16x2 a = __builtin_bit_cast(u16x2, fb_c), b = __builtin_bit_cast(u16x2, fb_n);
    return __builtin_bit_cast(unsigned, a + __builtin_shufflevector(b, b, 1, 0));
}

// 7-tap box of S over x (SangNom2.cpp:141-152) for the lane's pixels; the three values on either
// side come from the neighbouring lanes by DPP.  Sliding window: B[j+1] = B[j] - X[j-3] + X[j+4].
template <bool EDGE, int PXL>
__device__ __forceinline__ void box7(const int (&S)[PXL], int (&Bx)[PXL], const LaneRole& role)
{
    int L[3], R[3];  // L[k] = S[PXL-3+k] of the left lane, R[k] = S[k] of the right lane
    if constexpr (EDGE) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            L[k] = dpp_from_left(S[PXL - 3 + k]);
            R[k] = dpp_from_right(S[k]);
            if (role.first_real) L[k] = S[0];       // clamp to column 0
            if (role.last_real) R[k] = S[PXL - 1];  // clamp to column w-1
        }
    }
    auto X = [&](int i) -> int {  // S at lane-relative pixel i in [-3, PXL+3)
        if (i < 0) return EDGE ? L[i + 3] : dpp_from_left(S[PXL + i]);
        if (i >= PXL) return EDGE ? R[i - PXL] : dpp_from_right(S[i - PXL]);
        return S[i];
    };
    Bx[0] = S[0] + S[1] + S[2] + S[3] + X(-1) + X(-2) + X(-3);
#pragma unroll
    for (int j = 0; j + 1 < PXL; ++j) Bx[j + 1] = Bx[j] - X(j - 3) + X(j + 4);
}

// One cost buffer of one row: D[r+1] (or 0 past the last line pair), S, box, O, new A, and the
// stage-3 key of this buffer folded into the running minimum.  A is kept as 16-bit pairs
// (pixel j | pixel j+PXL/2 << 16); d35 / s35 carry the packed SangNom costs / sums of the row.
template <int BUF, bool HAS_NEXT, int PXL>
__device__ __forceinline__ void buffer_step(unsigned (&A)[PXL / 2], unsigned (&kmin)[PXL], const Line<PXL>& c,
                                            const Line<PXL>& n, const Line<PXL>& nn, const unsigned (&d35)[PXL],
                                            const unsigned (&s35)[PXL], const LaneRole& role)
{
    constexpr int H = PXL / 2;
    int D[PXL], S[PXL], Bx[PXL], O[PXL];
#pragma unroll
    for (int j = 0; j < PXL; ++j) {
        if constexpr (!HAS_NEXT) D[j] = 0;
        else if constexpr (BUF == 3) D[j] = (int)(d35[j] & 0xffffu);
        else if constexpr (BUF == 5) D[j] = (int)(d35[j] >> 16);
        else D[j] = cost<BUF>(n, nn, j);
    }
#pragma unroll
    for (int j = 0; j < H; ++j) {
        S[j] = (int)(A[j] & 0xffffu) + D[j];
        S[j + H] = (int)(A[j] >> 16) + D[j + H];
    }
    if (role.edge_wave) box7<true>(S, Bx, role);  // wave-uniform branch: only the box differs
    else box7<false>(S, Bx, role);
#pragma unroll
    for (int j = 0; j < PXL; ++j) O[j] = ubfe((unsigned)Bx[j], 4, 8);  // (box / 16) mod 256, SangNom2.cpp:152
#pragma unroll
    for (int j = 0; j < H; ++j) A[j] = (unsigned)(O[j] + D[j]) | ((unsigned)(O[j + H] + D[j + H]) << 16);
#pragma unroll
    for (int j = 0; j < PXL; ++j) {
        unsigned cand = (unsigned)candidate<BUF>(c, n, j);
        if constexpr (BUF == 3) cand += s35[j] & 0xffffu;
        if constexpr (BUF == 5) cand += s35[j] >> 16;
        const unsigned key = ((unsigned)O[j] << 16) | cand;
        kmin[j] = key < kmin[j] ? key : kmin[j];
    }
}

template <bool HAS_NEXT, int PXL>
__device__ __forceinline__ Raw<PXL> row_step(unsigned (&A)[kBuffers][PXL / 2], const Line<PXL>& c, const Line<PXL>& n,
                                             const Line<PXL>& nn, const LaneRole& role, unsigned thr_key)
{
    unsigned kmin[PXL], d35[PXL], s35[PXL];
#pragma unroll
    for (int j = 0; j < PXL; ++j) {
        // the `minBuf > aaf` arm: cost aaf+1, rank 0, value avg(c0, n0)
        kmin[j] = thr_key + (unsigned)(c.p[j + 3] + n.p[j + 3]);
        d35[j] = HAS_NEXT ? sg_costs(n.FB[j], nn.FB[j]) : 0u;
        s35[j] = sg_sums(c.FB[j], n.FB[j]);
    }
    buffer_step<3, HAS_NEXT>(A[3], kmin, c, n, nn, d35, s35, role);
    buffer_step<5, HAS_NEXT>(A[5], kmin, c, n, nn, d35, s35, role);
    buffer_step<0, HAS_NEXT>(A[0], kmin, c, n, nn, d35, s35, role);
    buffer_step<1, HAS_NEXT>(A[1], kmin, c, n, nn, d35, s35, role);
    buffer_step<2, HAS_NEXT>(A[2], kmin, c, n, nn, d35, s35, role);
    buffer_step<4, HAS_NEXT>(A[4], kmin, c, n, nn, d35, s35, role);
    buffer_step<6, HAS_NEXT>(A[6], kmin, c, n, nn, d35, s35, role);
    buffer_step<7, HAS_NEXT>(A[7], kmin, c, n, nn, d35, s35, role);
    buffer_step<8, HAS_NEXT>(A[8], kmin, c, n, nn, d35, s35, role);
    Raw<PXL> o;
    o.l = o.r = 0;
#pragma unroll
    for (int i = 0; i < PXL / 4; ++i) {
        uint32_t v = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) v |= (uint32_t)ubfe(kmin[4 * i + j], 1, 8) << (8 * j);  // (a + b + 1) >> 1
        o.m[i] = v;
    }
    return o;
}

// Ghost geometry: GH ghost lanes on each inner side of a wave, refresh period K rows.
template <int PXL>
struct Seam {
    static constexpr int GH = 2;                          // ghost lanes per side
    static constexpr int K = (GH * PXL) / 3;              // rows between two refreshes (5 for 16 px)
    static constexpr int kFirst = 64 - GH;                // real lanes of wave 0 (no left ghosts)
    static constexpr int kInner = 64 - 2 * GH;            // real lanes of every later wave
    static constexpr int kMaxWaves = PXL == 8 ? 8 : 16;
};

// LDS mailbox: [refresh parity][wave][side][ghost lane][buffer][packed A registers]
template <int PXL>
struct Mailbox {
    unsigned v[2][Seam<PXL>::kMaxWaves][2][Seam<PXL>::GH][kBuffers][PXL / 2];
};

template <int PXL>
__global__ void __launch_bounds__(PXL == 8 ? 512 : 1024) k_fused_u8(FusedArgs a)
{
    constexpr int H = PXL / 2;
    constexpr int GH = Seam<PXL>::GH;
    constexpr int K = Seam<PXL>::K;
    __shared__ Mailbox<PXL> mb;
    const int f = blockIdx.x;
    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    const int nw = a.nw;

    // lane -> global lane (column group): wave 0 owns lanes 0..63-GH (all 64 if it is the only
    // wave), later waves own lanes GH..63-GH (the last one up to 63); the other lanes are ghosts
    // that shadow the neighbouring wave's seam lanes.
    int gl;
    bool ghost;
    if (wave == 0) {
        gl = lane;
        ghost = (nw > 1) && lane >= 64 - GH;
    } else {
        gl = Seam<PXL>::kFirst + Seam<PXL>::kInner * (wave - 1) + (lane - GH);
        ghost = lane < GH || (lane >= 64 - GH && wave < nw - 1);
    }
    const bool live = gl < a.nl;       // has source pixels (real or ghost)
    const bool real = live && !ghost;  // owns output pixels
    const int x0 = gl * PXL;
    LaneRole role;
    role.first_real = gl == 0;
    role.last_real = gl == a.nl - 1;
    role.edge_wave = __builtin_amdgcn_readfirstlane(wave == 0 || wave == nw - 1);

    const uint8_t* src = a.src + (int64_t)f * a.src_frame_stride;
    uint8_t* dst = a.dst + (int64_t)f * a.dst_frame_stride;
    const int64_t src_step = (int64_t)(a.dh ? 1 : 2) * a.src_pitch;                 // kept line k -> k + 1
    const uint8_t* src_line = src + (int64_t)(a.dh ? 0 : a.offset) * a.src_pitch;  // kept line 0
    const int64_t dst_step = (int64_t)2 * a.dst_pitch;
    uint8_t* dst_line = dst + (int64_t)a.offset * a.dst_pitch;  // kept line 0 in dst
    auto keep = [&](uint8_t* row, const Raw<PXL>& q) {  // GetFrame's field copy, SangNom2.cpp:365 / :376
        if (real) store_own<PXL>(row, x0, q);
    };

    const int nk = a.nk;
    const int nr = nk - 1;
    const unsigned thr_key = ((unsigned)(a.thr + 1) << 16) + 1u;

    Line<PXL> L0, L1, L2;
    Raw<PXL> q0 = load_raw<PXL>(src_line, x0, a.w, live);
    Raw<PXL> q1 = nk > 1 ? load_raw<PXL>(src_line + src_step, x0, a.w, live) : q0;
    keep(dst_line, q0);
    if (a.offset == 1) keep(dst, q0);  // the line that cannot be interpolated, SangNom2.cpp:386-391
    if (nk > 1) keep(dst_line + dst_step, q1);
    unpack(L0, q0, role);
    unpack(L1, q1, role);

    // A[1] = O[0] + D[1] = D[1] (pool row 0 is never written: zero)
    unsigned A[kBuffers][H];
    {
        unsigned d35[PXL];
#pragma unroll
        for (int j = 0; j < PXL; ++j) d35[j] = sg_costs(L0.FB[j], L1.FB[j]);
#define SN_INIT(BUF)                                                                                       \
    _Pragma("unroll") for (int j = 0; j < H; ++j)                                                          \
    {                                                                                                      \
        const unsigned lo =                                                                                \
            BUF == 3 ? (d35[j] & 0xffffu) : BUF == 5 ? (d35[j] >> 16) : (unsigned)cost<BUF>(L0, L1, j);    \
        const unsigned hi = BUF == 3   ? (d35[j + H] & 0xffffu)                                            \
                            : BUF == 5 ? (d35[j + H] >> 16)                                                \
                                       : (unsigned)cost<BUF>(L0, L1, j + H);                               \
        A[BUF][j] = nr > 0 ? (lo | (hi << 16)) : 0u;                                                       \
    }
        SN_INIT(0) SN_INIT(1) SN_INIT(2) SN_INIT(3) SN_INIT(4) SN_INIT(5) SN_INIT(6) SN_INIT(7) SN_INIT(8)
#undef SN_INIT
    }

    // source / destination row cursors: src_next = kept line r+2 at the top of step r
    const uint8_t* src_next = src_line + 2 * src_step;
    uint8_t* dst_keep = dst_line + 2 * dst_step;  // where kept line r+1 goes (step r)
    uint8_t* out_row = dst_line + a.dst_pitch;    // interpolated line of step r
    Raw<PXL> qn = nk > 2 ? load_raw<PXL>(src_next, x0, a.w, live) : q1;  // kept line 2 = line r+1 of step 1
    src_next += src_step;

    // mailbox roles.  Seam lanes (the GH real lanes next to a seam) publish; ghost lanes receive.
    const int last_real_lane0 = 64 - 2 * GH;  // first of the GH real lanes at the right seam
    const bool pub_right = real && wave < nw - 1 && lane >= last_real_lane0 && lane < 64 - GH;
    const bool pub_left = real && wave > 0 && lane >= GH && lane < 2 * GH;
    const bool ghost_left = ghost && lane < GH;
    const bool ghost_right = ghost && lane >= 64 - GH;
    const int slot = ghost_left ? lane : ghost_right ? lane - (64 - GH) : pub_right ? lane - last_real_lane0 : lane - GH;

    // The interpolated line of row r is stored at the top of row r+1, after that row has waited
    // for its prefetched source line: a store issued at the end of the row would sit between the
    // prefetch and its wait (vmcnt counts loads and stores in order) and expose its latency.
    Raw<PXL> pending;
    pending.l = pending.r = 0;
#pragma unroll
    for (int i = 0; i < PXL / 4; ++i) pending.m[i] = 0;

    // One pool row r: c = K[r-1], n = K[r], nn = K[r+1].
    auto step = [&](int r, const Line<PXL>& c, const Line<PXL>& n, Line<PXL>& nn, auto has_next_tag) {
        constexpr bool HAS_NEXT = decltype(has_next_tag)::value;
        Raw<PXL> qnext = qn;
        if constexpr (HAS_NEXT) {
            unpack(nn, qn, role);  // waits for the line prefetched one row ago
            keep(dst_keep, qn);
            dst_keep += dst_step;
        }
        if (r > 1) {
            keep(out_row, pending);
            out_row += dst_step;
        }
        if constexpr (HAS_NEXT) {
            if (r + 2 <= nr) qnext = load_raw<PXL>(src_next, x0, a.w, live);  // prefetch K[r+2]
            src_next += src_step;
        }
        const int par = (r / K) & 1;
        if (r > 1 && (r - 1) % K == 0 && !(a.dbg & 1)) {
            // refresh: the ghosts reload the A state their owners published at the end of row r-1
            __syncthreads();
            if (ghost_left) {
#pragma unroll
                for (int b = 0; b < kBuffers; ++b)
#pragma unroll
                    for (int k = 0; k < H; ++k) A[b][k] = mb.v[par][wave][0][slot][b][k];
            }
            if (ghost_right) {
#pragma unroll
                for (int b = 0; b < kBuffers; ++b)
#pragma unroll
                    for (int k = 0; k < H; ++k) A[b][k] = mb.v[par][wave][1][slot][b][k];
            }
        }
        pending = row_step<HAS_NEXT>(A, c, n, nn, role, thr_key);
        if constexpr (HAS_NEXT) {
            if (r % K == 0 && !(a.dbg & 1)) {
                // row r+1 is a refresh row: publish A[r+1] of the seam lanes
                const int wpar = ((r + 1) / K) & 1;
                if (pub_right) {
#pragma unroll
                    for (int b = 0; b < kBuffers; ++b)
#pragma unroll
                        for (int k = 0; k < H; ++k) mb.v[wpar][wave + 1][0][slot][b][k] = A[b][k];
                }
                if (pub_left) {
#pragma unroll
                    for (int b = 0; b < kBuffers; ++b)
#pragma unroll
                        for (int k = 0; k < H; ++k) mb.v[wpar][wave - 1][1][slot][b][k] = A[b][k];
                }
            }
        }
        qn = qnext;
    };
    using T = std::integral_constant<bool, true>;
    using F = std::integral_constant<bool, false>;

    // rows 1 .. nr-1 have a following line pair; the last row does not (pool row bh is zero).
    // Lines rotate through register copies: one loop body keeps the code small enough for the
    // instruction cache (three unrolled rotations did not).
    for (int r = 1; r < nr; ++r) {
        step(r, L0, L1, L2, T{});
        L0 = L1;
        L1 = L2;
    }
    if (nr >= 1) {
        step(nr, L0, L1, L2, F{});
        keep(out_row, pending);
    }

    // the line that cannot be interpolated when the top field is kept, SangNom2.cpp:380-385:
    // dst row h-1 := dst row h-2 = K[nk-1]
    if (a.offset == 0 && real) {
        const Raw<PXL> q = load_raw<PXL>(src_line + (int64_t)(nk - 1) * src_step, x0, a.w, true);
        store_own<PXL>(dst + (int64_t)(2 * nk - 1) * a.dst_pitch, x0, q);
    }
}

template <int PXL>
int waves_for_t(int nl)
{
    return nl <= 64 ? 1 : 1 + (nl - Seam<PXL>::kFirst + Seam<PXL>::kInner - 1) / Seam<PXL>::kInner;
}
int waves_for(int nl, int pxl) { return pxl == 8 ? waves_for_t<8>(nl) : waves_for_t<4>(nl); }

// Pixels per lane for a plane of width w: 8 (8 waves, 16-pixel ghost zones refreshed every 5 rows)
// when the plane fits, else 4.  SN_FUSED_PXL=4 forces the 4-pixel variant for experiments.
int choose_pxl(int w)
{
    static const int forced = [] {
        const char* e = getenv("SN_FUSED_PXL");
        return e ? atoi(e) : 0;
    }();
    if (forced == 4 && waves_for(w / 4, 4) <= 16) return 4;
    if (waves_for(w / 8, 8) <= 8) return 8;
    return 4;
}

}  // namespace

// ---- host side -------------------------------------------------------------------------------

bool fused_v2_plane_ok(int w)
{
    if (w % 32 != 0) return false;
    return waves_for(w / 4, 4) <= 16 || waves_for(w / 8, 8) <= 8;
}

// Eligible when the clip is 8-bit, narrow enough for one workgroup, and either every processed plane is
// as large as the pool (no pass can see another pass's leftovers: SURVEY.md 0.7) or the chroma planes are
// subsampled AND luma is processed first (then the fused kernel couples the passes through scratch pools;
// without a luma pass the chroma passes would see the previous FRAME's leftovers: pool path).
static bool chroma_subsampled_and_processed(const sn_config& c)
{
    const int np = c.num_planes < 3 ? c.num_planes : 3;
    return np == 3 && (c.dh || c.chroma) && (c.sub_w != 0 || c.sub_h != 0);
}

bool fused_needs_pools(const sn_config& c) { return chroma_subsampled_and_processed(c); }

bool fused_plane_eligible(int bytes_per_sample, int w)
{
    if (bytes_per_sample == 1) return fused_v3_plane_ok(w) || fused_v2_plane_ok(w);
    if (bytes_per_sample == 2) return fused_u16_plane_ok(w);
    return fused_f32_plane_ok(w);
}

bool fused_padded_plane_eligible(int bytes_per_sample, int w)
{
    const int stride = (w + 31) & ~31;
    if (w % 8 != 0) return false;  // the plane must end on a lane boundary (8 columns per lane)
    if (bytes_per_sample == 1) return fused_v3_plane_ok(stride);
    if (bytes_per_sample == 2) return fused_u16_plane_ok(stride);
    return fused_f32_plane_ok(stride);
}

bool fused_eligible(const sn_config& c)
{
    if (c.bytes_per_sample == 2) {
        if (!fused_u16_plane_ok(c.width)) return false;
    } else if (c.bytes_per_sample == 1) {
        if (!fused_v3_plane_ok(c.width) && !fused_v2_plane_ok(c.width)) return false;
    } else {
        return fused_f32_plane_ok(c.width) && !chroma_subsampled_and_processed(c);
    }
    if (chroma_subsampled_and_processed(c)) {
        if (!(c.dh || c.luma)) return false;
        if (c.bytes_per_sample == 1 && !fused_v3_plane_ok(c.width)) return false;
        if ((c.width >> c.sub_w) % 8 != 0) return false;
    }
    return true;
}

hipError_t launch_fused_u8(hipStream_t st, const PlaneArgs& p, double threshold, int nframes)
{
    FusedArgs a{};
    a.src = p.src;
    a.dst = p.dst;
    a.src_frame_stride = p.src_frame_stride;
    a.dst_frame_stride = p.dst_frame_stride;
    a.src_pitch = p.src_pitch;
    a.dst_pitch = p.dst_pitch;
    a.w = p.w;
    a.nk = p.h_out / 2;
    a.offset = p.offset;
    a.dh = p.dh;
    a.thr = (int)threshold;
    const int pxl = choose_pxl(p.w);
    a.nl = p.w / pxl;
    a.nw = waves_for(a.nl, pxl);
    static const int dbg = [] { const char* e = getenv("SN_FUSED_DEBUG"); return e ? atoi(e) : 0; }();
    a.dbg = dbg;
    if (pxl == 8) hipLaunchKernelGGL(k_fused_u8<8>, dim3(nframes), dim3(a.nw * 64), 0, st, a);
    else hipLaunchKernelGGL(k_fused_u8<4>, dim3(nframes), dim3(a.nw * 64), 0, st, a);
    return hipGetLastError();
}

// Pointer / pitch alignment the 8-byte vector accesses need.
bool fused_layout_ok(const PlaneArgs& p)
{
    auto a8 = [](uintptr_t v) { return (v & 7) == 0; };
    return a8((uintptr_t)p.src) && a8((uintptr_t)p.dst) && a8((uintptr_t)p.src_pitch) && a8((uintptr_t)p.dst_pitch) &&
           a8((uintptr_t)p.src_frame_stride) && a8((uintptr_t)p.dst_frame_stride);
}

}  // namespace sn
