// sn_fused_u8.hip -- the fused 8-bit kernel: frame assembly + stage 1 + stage 2 + stage 3 of one
// plane in ONE pass, with the nine cost buffers never leaving registers.
//
// Reference semantics: /root/reference/src/SangNom2.cpp:74-124 (prepareBuffers_c), :126-159
// (processBuffers_c), :161-257 (finalizePlane_c), :361-391 (GetFrame's copies).
//
// Mapping (DESIGN.md "Fused kernel"):
//   * one workgroup per (frame, plane); the workgroup sweeps the plane top to bottom because
//     stage 2 is a vertical recurrence  O[r] = box7(O[r-1] + D[r] + D[r+1]) / 16 (mod 256);
//   * a lane owns PXL = 8 consecutive pixels of every row and keeps, per cost buffer, only
//     A[r] = O[r-1] + D[r] (8 registers): S[r] = A[r] + D[r+1], O[r] = (box7(S[r]) >> 4) & 255,
//     A[r+1] = O[r] + D[r+1];
//   * the +-3 horizontal taps of the 7-tap box come from the neighbouring lanes with DPP
//     wave_shr:1 / wave_shl:1 folded into the adds (wavefront shuffles, no LDS);
//   * lanes 0 and 63 of a wave are "ghost" lanes: they recompute the cost values D of the lane
//     that the neighbouring wave owns and receive that lane's A state through a tiny LDS mailbox
//     once per row (one s_barrier per row), so the sweep is exact across wave seams;
//   * image edges clamp S to the first / last column (loadPixel on the line buffer,
//     SangNom2.cpp:144-150): handled by per-lane selects in the two edge waves only;
//   * stage 3's priority ladder is a single unsigned minimum over keys
//     (O << 16) | (rank << 12) | (a + b + 1): smallest cost wins, ties go to the reference's
//     priority order, and the winner's average is (key & 0x1ff) >> 1.  The `minBuf > aaf` test
//     is a tenth key with cost aaf + 1 and rank 0 that selects avg(c0, n0).
//
// Everything is integer; results are bit-exact to the pool path and to the opt=0 reference.
#include <type_traits>

#include "sn_internal.h"

namespace sn {

namespace {

constexpr int PXL = 8;        // pixels per lane
constexpr int kMaxWaves = 8;  // 512 threads -> 256 VGPRs per lane available
constexpr int kTaps = PXL + 6;

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
};

__device__ __forceinline__ int dpp_from_left(int v)  // lane i receives lane i-1 (0 for lane 0)
{
    return __builtin_amdgcn_update_dpp(0, v, 0x138 /* wave_shr:1 */, 0xf, 0xf, true);
}
__device__ __forceinline__ int dpp_from_right(int v)  // lane i receives lane i+1 (0 for lane 63)
{
    return __builtin_amdgcn_update_dpp(0, v, 0x130 /* wave_shl:1 */, 0xf, 0xf, true);
}
__device__ __forceinline__ int ubfe(int v, int off, int width) { return (int)__builtin_amdgcn_ubfe((unsigned)v, off, width); }
__device__ __forceinline__ int sad(int a, int b) { return (int)__builtin_amdgcn_sad_u16((unsigned)a, (unsigned)b, 0u); }

// One kept line as a lane sees it: its 8 pixels plus three on each side, and the two SangNom
// values per pixel (calculateSangNom, SangNom2.cpp:60-65):
//   F[j] = sg(p[-1], p[0], p[+1])   B[j] = sg(p[+1], p[0], p[-1])
// For a pair (c, n): forwardSangNom1 = F(c), forwardSangNom2 = B(n), backwardSangNom1 = B(c),
// backwardSangNom2 = F(n)  (SangNom2.cpp:100-103).
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

struct Line {
    int p[kTaps];       // p[i] = pixel x0 - 3 + i (edge-clamped)
    unsigned FB[PXL];   // F | B << 16 (packed to halve the registers a line occupies)
};

struct Raw {  // a line as loaded: left dword (x0-4..x0-1), own 8 bytes, right dword (x0+8..x0+11)
    uint32_t l, m0, m1, r;
};

__device__ __forceinline__ Raw load_raw(const uint8_t* row, int x0, int w, bool live)
{
    Raw q{0, 0, 0, 0};
    if (live) {
        const uint2 m = *reinterpret_cast<const uint2*>(row + x0);
        q.m0 = m.x;
        q.m1 = m.y;
        q.l = x0 > 0 ? *reinterpret_cast<const uint32_t*>(row + x0 - 4) : (m.x & 0xff) * 0x01010101u;
        q.r = x0 + PXL < w ? *reinterpret_cast<const uint32_t*>(row + x0 + PXL) : (m.y >> 24) * 0x01010101u;
    }
    return q;
}

__device__ __forceinline__ void unpack(Line& L, const Raw& q)
{
    L.p[0] = ubfe(q.l, 8, 8);
    L.p[1] = ubfe(q.l, 16, 8);
    L.p[2] = (int)(q.l >> 24);
    L.p[3] = (int)(q.m0 & 0xff);
    L.p[4] = ubfe(q.m0, 8, 8);
    L.p[5] = ubfe(q.m0, 16, 8);
    L.p[6] = (int)(q.m0 >> 24);
    L.p[7] = (int)(q.m1 & 0xff);
    L.p[8] = ubfe(q.m1, 8, 8);
    L.p[9] = ubfe(q.m1, 16, 8);
    L.p[10] = (int)(q.m1 >> 24);
    L.p[11] = (int)(q.r & 0xff);
    L.p[12] = ubfe(q.r, 8, 8);
    L.p[13] = ubfe(q.r, 16, 8);
#pragma unroll
    for (int j = 0; j < PXL; ++j) {
        const int a = L.p[j + 2], b = L.p[j + 3], c = L.p[j + 4];
        const int x5 = b * 5;
        const unsigned F = (unsigned)ubfe(4 * a + x5 - c, 3, 8);  // arithmetic >> 3, wrap to 8 bits == bits 3..10
        const unsigned B = (unsigned)ubfe(4 * c + x5 - a, 3, 8);
        L.FB[j] = F | (B << 16);
    }
}

// Stage 1 for buffer BUF of pixel j of the pair (c, n) -- the Buffers enum order of
// /root/reference/src/SangNom2.h:8-20.
template <int BUF>
__device__ __forceinline__ int cost(const Line& c, const Line& n, int j)
{
    const int i = j + 3;
    if constexpr (BUF == 0) return sad(c.p[i - 3], n.p[i + 3]);  // ADIFF_M3_P3
    if constexpr (BUF == 1) return sad(c.p[i - 2], n.p[i + 2]);  // ADIFF_M2_P2
    if constexpr (BUF == 2) return sad(c.p[i - 1], n.p[i + 1]);  // ADIFF_M1_P1
    if constexpr (BUF == 3) return 0;                            // SG_FORWARD: see sg_costs()
    if constexpr (BUF == 4) return sad(c.p[i], n.p[i]);          // ADIFF_P0_M0
    if constexpr (BUF == 5) return 0;                            // SG_REVERSE: see sg_costs()
    if constexpr (BUF == 6) return sad(c.p[i + 1], n.p[i - 1]);  // ADIFF_P1_M1
    if constexpr (BUF == 7) return sad(c.p[i + 2], n.p[i - 2]);  // ADIFF_P2_M2
    return sad(c.p[i + 3], n.p[i - 3]);                          // ADIFF_P3_M3
}

// Stage 3 candidate of buffer BUF: (rank << 12) + a + b + 1, rank = position in the reference's
// if/else ladder (SangNom2.cpp:214-249): P4, P5, P3, P6, P2, P7, P1, P8, P0 -> 1..9.
template <int BUF>
__device__ __forceinline__ int candidate(const Line& c, const Line& n, int j)
{
    const int i = j + 3;
    if constexpr (BUF == 0) return c.p[i - 3] + n.p[i + 3] + ((9 << 12) + 1);
    if constexpr (BUF == 1) return c.p[i - 2] + n.p[i + 2] + ((7 << 12) + 1);
    if constexpr (BUF == 2) return c.p[i - 1] + n.p[i + 1] + ((5 << 12) + 1);
    if constexpr (BUF == 3) return (3 << 12) + 1;  // + low half of sg_sums()
    if constexpr (BUF == 4) return c.p[i] + n.p[i] + ((1 << 12) + 1);
    if constexpr (BUF == 5) return (2 << 12) + 1;  // + high half of sg_sums()
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

struct LaneRole {
    bool edge_wave;   // wave holds the first or the last real lane: box needs the clamp selects
    bool first_real;  // lane owns column 0
    bool last_real;   // lane owns column w - 1
};

// 7-tap box of S over x (SangNom2.cpp:141-152) for the lane's 8 pixels; neighbours by DPP.
template <bool EDGE>
__device__ __forceinline__ void box7(const int (&S)[PXL], int (&Bx)[PXL], const LaneRole& role)
{
    int L5, L6, L7, R0, R1, R2;
    if constexpr (EDGE) {
        L5 = dpp_from_left(S[5]);
        L6 = dpp_from_left(S[6]);
        L7 = dpp_from_left(S[7]);
        R0 = dpp_from_right(S[0]);
        R1 = dpp_from_right(S[1]);
        R2 = dpp_from_right(S[2]);
        if (role.first_real) L5 = L6 = L7 = S[0];  // clamp to column 0
        if (role.last_real) R0 = R1 = R2 = S[7];   // clamp to column w-1
        const int q = S[0] + S[1] + S[2] + S[3];
        Bx[0] = q + L5 + L6 + L7;
        Bx[1] = Bx[0] - L5 + S[4];
        Bx[2] = Bx[1] - L6 + S[5];
        Bx[3] = Bx[2] - L7 + S[6];
        Bx[4] = Bx[3] - S[0] + S[7];
        Bx[5] = Bx[4] - S[1] + R0;
        Bx[6] = Bx[5] - S[2] + R1;
        Bx[7] = Bx[6] - S[3] + R2;
    } else {
        const int q = S[0] + S[1] + S[2] + S[3];
        Bx[0] = q + dpp_from_left(S[7]) + dpp_from_left(S[6]) + dpp_from_left(S[5]);
        Bx[1] = Bx[0] - dpp_from_left(S[5]) + S[4];
        Bx[2] = Bx[1] - dpp_from_left(S[6]) + S[5];
        Bx[3] = Bx[2] - dpp_from_left(S[7]) + S[6];
        Bx[4] = Bx[3] - S[0] + S[7];
        Bx[5] = Bx[4] - S[1] + dpp_from_right(S[0]);
        Bx[6] = Bx[5] - S[2] + dpp_from_right(S[1]);
        Bx[7] = Bx[6] - S[3] + dpp_from_right(S[2]);
    }
}

// One cost buffer of one row: D[r+1] (or 0 past the last line pair), S, box, O, new A, and the
// stage-3 key of this buffer folded into the running minimum.  A is kept as 16-bit pairs
// (pixel j | pixel j+4 << 16); d35 / s35 carry the packed SangNom costs / sums of the row.
template <int BUF, bool HAS_NEXT>
__device__ __forceinline__ void buffer_step(unsigned (&A)[PXL / 2], unsigned (&kmin)[PXL], const Line& c, const Line& n,
                                            const Line& nn, const unsigned (&d35)[PXL], const unsigned (&s35)[PXL],
                                            const LaneRole& role)
{
    int D[PXL], S[PXL], Bx[PXL];
#pragma unroll
    for (int j = 0; j < PXL; ++j) {
        if constexpr (!HAS_NEXT) D[j] = 0;
        else if constexpr (BUF == 3) D[j] = (int)(d35[j] & 0xffffu);
        else if constexpr (BUF == 5) D[j] = (int)(d35[j] >> 16);
        else D[j] = cost<BUF>(n, nn, j);
    }
#pragma unroll
    for (int j = 0; j < PXL / 2; ++j) {
        S[j] = (int)(A[j] & 0xffffu) + D[j];
        S[j + 4] = (int)(A[j] >> 16) + D[j + 4];
    }
    if (role.edge_wave) box7<true>(S, Bx, role);  // wave-uniform branch: only the box differs
    else box7<false>(S, Bx, role);
    int O[PXL];
#pragma unroll
    for (int j = 0; j < PXL; ++j) O[j] = ubfe(Bx[j], 4, 8);  // (box / 16) mod 256, SangNom2.cpp:152
#pragma unroll
    for (int j = 0; j < PXL / 2; ++j) A[j] = (unsigned)(O[j] + D[j]) | ((unsigned)(O[j + 4] + D[j + 4]) << 16);
#pragma unroll
    for (int j = 0; j < PXL; ++j) {
        unsigned cand = (unsigned)candidate<BUF>(c, n, j);
        if constexpr (BUF == 3) cand += s35[j] & 0xffffu;
        if constexpr (BUF == 5) cand += s35[j] >> 16;
        const unsigned key = ((unsigned)O[j] << 16) | cand;
        kmin[j] = key < kmin[j] ? key : kmin[j];
    }
}

}  // namespace

// LDS mailbox: [row parity][wave][side][buffer][3 packed A registers + pad]
struct Mailbox {
    unsigned v[2][kMaxWaves][2][kBuffers][4];
};

template <bool HAS_NEXT>
__device__ __forceinline__ void row_step(unsigned (&A)[kBuffers][PXL / 2], const Line& c, const Line& n, const Line& nn,
                                         const LaneRole& role, unsigned thr_key, uint8_t* out_row, int x0, bool real)
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
    if (real) {
        uint32_t lo = 0, hi = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            lo |= (uint32_t)ubfe((int)kmin[j], 1, 8) << (8 * j);      // (a + b + 1) >> 1
            hi |= (uint32_t)ubfe((int)kmin[j + 4], 1, 8) << (8 * j);
        }
        *reinterpret_cast<uint2*>(out_row + x0) = make_uint2(lo, hi);
    }
}

__global__ void __launch_bounds__(kMaxWaves * 64) k_fused_u8(FusedArgs a)
{
    __shared__ Mailbox mb;
    const int f = blockIdx.x;
    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    const int nw = a.nw;

    // lane -> global lane (column group) mapping: wave 0 owns lanes 0..62 (63 if it is the only
    // wave), later waves own lanes 1..62; lane 0 / lane 63 are ghosts of the neighbouring wave.
    int gl;
    bool ghost;
    if (wave == 0) {
        gl = lane;
        ghost = (nw > 1) && lane == 63;
    } else {
        gl = 63 + 62 * (wave - 1) + (lane - 1);
        ghost = lane == 0 || (lane == 63 && wave < nw - 1);
    }
    const bool live = gl < a.nl;         // has source pixels (real or ghost)
    const bool real = live && !ghost;    // owns output pixels
    const int x0 = gl * PXL;
    LaneRole role;
    role.first_real = gl == 0;
    role.last_real = gl == a.nl - 1;
    role.edge_wave = __builtin_amdgcn_readfirstlane(wave == 0 || wave == nw - 1);

    const uint8_t* src = a.src + (int64_t)f * a.src_frame_stride;
    uint8_t* dst = a.dst + (int64_t)f * a.dst_frame_stride;
    const int64_t src_step = (int64_t)(a.dh ? 1 : 2) * a.src_pitch;  // kept line k -> k + 1 in src
    const uint8_t* src_line = src + (int64_t)(a.dh ? 0 : a.offset) * a.src_pitch;  // kept line 0
    const int64_t dst_step = (int64_t)2 * a.dst_pitch;
    uint8_t* dst_line = dst + (int64_t)a.offset * a.dst_pitch;  // kept line 0 in dst
    auto keep = [&](uint8_t* row, const Raw& q) {  // GetFrame's field copy, SangNom2.cpp:365 / :376
        if (real) *reinterpret_cast<uint2*>(row + x0) = make_uint2(q.m0, q.m1);
    };

    const int nk = a.nk;
    const int nr = nk - 1;
    const unsigned thr_key = ((unsigned)(a.thr + 1) << 16) + 1u;

    Line L0, L1, L2;
    Raw q0 = load_raw(src_line, x0, a.w, live);
    Raw q1 = nk > 1 ? load_raw(src_line + src_step, x0, a.w, live) : q0;
    keep(dst_line, q0);
    if (a.offset == 1) keep(dst, q0);  // the line that cannot be interpolated, SangNom2.cpp:386-391
    if (nk > 1) keep(dst_line + dst_step, q1);
    unpack(L0, q0);
    unpack(L1, q1);

    // A[1] = O[0] + D[1] = D[1] (pool row 0 is never written: zero)
    unsigned A[kBuffers][PXL / 2];
    {
        unsigned d35[PXL];
#pragma unroll
        for (int j = 0; j < PXL; ++j) d35[j] = sg_costs(L0.FB[j], L1.FB[j]);
#define SN_INIT(BUF)                                                                                   \
    _Pragma("unroll") for (int j = 0; j < PXL / 2; ++j)                                                \
    {                                                                                                  \
        const unsigned lo = BUF == 3 ? (d35[j] & 0xffffu) : BUF == 5 ? (d35[j] >> 16) : (unsigned)cost<BUF>(L0, L1, j);        \
        const unsigned hi = BUF == 3 ? (d35[j + 4] & 0xffffu) : BUF == 5 ? (d35[j + 4] >> 16) : (unsigned)cost<BUF>(L0, L1, j + 4); \
        A[BUF][j] = nr > 0 ? (lo | (hi << 16)) : 0u;                                                   \
    }
        SN_INIT(0) SN_INIT(1) SN_INIT(2) SN_INIT(3) SN_INIT(4) SN_INIT(5) SN_INIT(6) SN_INIT(7) SN_INIT(8)
#undef SN_INIT
    }

    // source / destination row cursors: src_next = kept line r+2 at the top of step r
    const uint8_t* src_next = src_line + 2 * src_step;
    uint8_t* dst_keep = dst_line + 2 * dst_step;          // where kept line r+1 goes (step r)
    uint8_t* out_row = dst_line + a.dst_pitch;            // interpolated line of step r
    Raw qn = nk > 2 ? load_raw(src_next, x0, a.w, live) : q1;  // kept line 2 = line r+1 of step 1
    src_next += src_step;

    // mailbox roles
    const bool pub_right = real && wave < nw - 1 && lane == 62;  // feeds next wave's lane 0
    const bool pub_left = real && wave > 0 && lane == 1;         // feeds previous wave's lane 63
    const bool ghost_left = ghost && lane == 0;
    const bool ghost_right = ghost && lane == 63;

    // One pool row r: c = K[r-1], n = K[r], nn = K[r+1].
    auto step = [&](int r, const Line& c, const Line& n, Line& nn, auto has_next_tag) {
        constexpr bool HAS_NEXT = decltype(has_next_tag)::value;
        Raw qnext = qn;
        if constexpr (HAS_NEXT) {
            keep(dst_keep, qn);
            dst_keep += dst_step;
            if (r + 2 <= nr) qnext = load_raw(src_next, x0, a.w, live);  // prefetch K[r+2]
            src_next += src_step;
            unpack(nn, qn);
        }
        if (r > 1) {
            // receive the neighbour wave's A state for this row (published at the end of row r-1)
            __syncthreads();
            if (ghost_left) {
#pragma unroll
                for (int b = 0; b < kBuffers; ++b) {
                    A[b][1] = mb.v[r & 1][wave][0][b][0];
                    A[b][2] = mb.v[r & 1][wave][0][b][1];
                    A[b][3] = mb.v[r & 1][wave][0][b][2];
                }
            }
            if (ghost_right) {
#pragma unroll
                for (int b = 0; b < kBuffers; ++b) {
                    A[b][0] = mb.v[r & 1][wave][1][b][0];
                    A[b][1] = mb.v[r & 1][wave][1][b][1];
                    A[b][2] = mb.v[r & 1][wave][1][b][2];
                }
            }
        }
        row_step<HAS_NEXT>(A, c, n, nn, role, thr_key, out_row, x0, real);
        out_row += dst_step;
        if constexpr (HAS_NEXT) {
            // publish A[r+1] of the seam lanes for the neighbouring waves' ghosts: the left ghost
            // needs pixels 5..7 (high halves of A[1..3]), the right ghost pixels 0..2 (low halves
            // of A[0..2]); whole packed registers travel.
            if (pub_right) {
#pragma unroll
                for (int b = 0; b < kBuffers; ++b) {
                    mb.v[(r + 1) & 1][wave + 1][0][b][0] = A[b][1];
                    mb.v[(r + 1) & 1][wave + 1][0][b][1] = A[b][2];
                    mb.v[(r + 1) & 1][wave + 1][0][b][2] = A[b][3];
                }
            }
            if (pub_left) {
#pragma unroll
                for (int b = 0; b < kBuffers; ++b) {
                    mb.v[(r + 1) & 1][wave - 1][1][b][0] = A[b][0];
                    mb.v[(r + 1) & 1][wave - 1][1][b][1] = A[b][1];
                    mb.v[(r + 1) & 1][wave - 1][1][b][2] = A[b][2];
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
    if (nr >= 1) step(nr, L0, L1, L2, F{});

    // the line that cannot be interpolated when the top field is kept, SangNom2.cpp:380-385:
    // dst row h-1 := dst row h-2 = K[nk-1]
    if (a.offset == 0 && real) {
        const uint2 m = *reinterpret_cast<const uint2*>(src_line + (int64_t)(nk - 1) * src_step + x0);
        *reinterpret_cast<uint2*>(dst + (int64_t)(2 * nk - 1) * a.dst_pitch + x0) = m;
    }
}

// ---- host side -------------------------------------------------------------------------------

static int waves_for(int nl) { return nl <= 64 ? 1 : 1 + (nl - 63 + 61) / 62; }

bool fused_plane_ok(int w, int bytes)
{
    if (bytes != 1 || w % 32 != 0) return false;
    return waves_for(w / PXL) <= kMaxWaves;
}

// Eligible when every processed plane is 8-bit, as large as the pool (so that no pass can see
// another pass's leftovers: SURVEY.md 0.7) and narrow enough for one workgroup.
bool fused_eligible(const sn_config& c)
{
    if (c.bytes_per_sample != 1) return false;
    if (!fused_plane_ok(c.width, 1)) return false;
    const int np = c.num_planes < 3 ? c.num_planes : 3;
    for (int p = 1; p < np; ++p) {
        const bool processed = c.dh || c.chroma;
        if (processed && (c.sub_w != 0 || c.sub_h != 0)) return false;
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
    a.nl = p.w / PXL;
    a.nw = waves_for(a.nl);
    hipLaunchKernelGGL(k_fused_u8, dim3(nframes), dim3(a.nw * 64), 0, st, a);
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
