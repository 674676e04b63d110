// sn_fused_f32_v3.hip -- the fused sweep for 32-bit float samples (Y32, YUV444PS; no chroma coupling).
//
// Same structure as sn_fused_u16_v3.hip (read the header of sn_fused_u8_v3.hip for the algorithm and the
// exactness argument): one workgroup sweeps one plane top to bottom, a lane owns 8 consecutive pixels of
// a row, one pixel per register, the nine cost buffers live in registers / thread-private LDS as
// A[r] = O[r-1] + D[r], ghost lanes keep the wave seams exact with one barrier every 5 rows.
//
// What float changes, all of it to keep the reference's rounding (one rounding per operation, no
// contraction, SangNom2.cpp:36-72 with T = float):
//   * the 7-tap sum is evaluated left to right for every pixel, (((((m3 + m2) + m1) + c0) + p1) + p2) + p3
//     (SangNom2.cpp:152) -- a sliding sum would round differently;
//   * S = (O[r-1] + D[r]) + D[r+1] in that order (:141): A holds the parenthesis;
//   * sum / 16 is a multiplication by 0.0625 (exact for every float, denormals are kept: the library is built
//     with -fno-gpu-flush-denormals-to-zero);
//   * stage 3 cannot pack a rank into the value, so the sweep visits the buffers from the lowest rung of
//     the reference's ladder to the highest (P0, P8, P1, P7, P2, P6, P3, P5, P4) and keeps
//     (minimum, rank of the last buffer that was <= minimum): ties end on the higher rung, as :211-249.
// Inputs are assumed finite, as the reference does (its min / == ladder has no NaN handling either).
#include <stdlib.h>

#include <type_traits>

#include "sn_fused_v3_common.h"

namespace sn {
namespace f32 {

using namespace v3c;
constexpr int kMaxWaves = 8;

__device__ __forceinline__ unsigned bits(float x) { return __builtin_bit_cast(unsigned, x); }
__device__ __forceinline__ float flt(unsigned x) { return __builtin_bit_cast(float, x); }

struct Line {
    float P[PXL + 6];  // P[i] = pixel x0 - 3 + i (edge-clamped)
    float F[PXL];      // forward / backward SangNom values (calculateSangNom, SangNom2.cpp:66-72)
    float B[PXL];
};

struct Raw {  // pixels x0-4 .. x0+11 as 16 dwords
    u32x4 q[4];
};

struct LaneRole {
    bool edge_wave;  // wave holds the first or last column
    bool first;      // lane owns column 0
    bool last;       // lane owns the last column of the sweep
    bool line_last;  // lane owns the last column of the source lines (PADDED: region_w - 1)
    float inside;    // PADDED: 1 where the lane's columns belong to the plane, 0 in the padding
    unsigned first_mask, last_mask;  // all ones where first / last
};

// sum = ((p1 * 4 + p2 * 5) - p3) * 0.125; p1 * 4 is exact, so the fma rounds exactly like the reference's
// multiply-then-add
__device__ __forceinline__ void sangnom_values(Line& L)
{
#pragma unroll
    for (int j = 0; j < PXL; ++j) {
        const float a = L.P[j + 2], b = L.P[j + 3], c = L.P[j + 4];
        const float q5 = b * 5.0f;
        L.F[j] = (__builtin_fmaf(a, 4.0f, q5) - c) * 0.125f;
        L.B[j] = (__builtin_fmaf(c, 4.0f, q5) - a) * 0.125f;
    }
}

__device__ __forceinline__ void unpack(Line& L, const Raw& q, const LaneRole& role)
{
    unsigned d[16] = {q.q[0].x, q.q[0].y, q.q[0].z, q.q[0].w, q.q[1].x, q.q[1].y, q.q[1].z, q.q[1].w,
                      q.q[2].x, q.q[2].y, q.q[2].z, q.q[2].w, q.q[3].x, q.q[3].y, q.q[3].z, q.q[3].w};
    if (role.edge_wave && role.first) {  // loaded from column 0 instead of x0 - 4: four slots early
#pragma unroll
        for (int i = 15; i >= 4; --i) d[i] = d[i - 4];
    }
#pragma unroll
    for (int i = 0; i < PXL + 6; ++i) L.P[i] = flt(d[i + 1]);
    if (role.edge_wave) {  // loadPixel's clamp, SangNom2.cpp:25-34
        if (role.first) L.P[0] = L.P[1] = L.P[2] = L.P[3];
        if (role.line_last) L.P[11] = L.P[12] = L.P[13] = L.P[10];
    }
    sangnom_values(L);
}

template <int BUF>
__device__ __forceinline__ float cost(const Line& c, const Line& n, int j)
{
    const int i = j + 3;
    if constexpr (BUF == 0) return __builtin_fabsf(c.P[i - 3] - n.P[i + 3]);
    if constexpr (BUF == 1) return __builtin_fabsf(c.P[i - 2] - n.P[i + 2]);
    if constexpr (BUF == 2) return __builtin_fabsf(c.P[i - 1] - n.P[i + 1]);
    if constexpr (BUF == 3) return __builtin_fabsf(c.F[j] - n.B[j]);
    if constexpr (BUF == 4) return __builtin_fabsf(c.P[i] - n.P[i]);
    if constexpr (BUF == 5) return __builtin_fabsf(c.B[j] - n.F[j]);
    if constexpr (BUF == 6) return __builtin_fabsf(c.P[i + 1] - n.P[i - 1]);
    if constexpr (BUF == 7) return __builtin_fabsf(c.P[i + 2] - n.P[i - 2]);
    return __builtin_fabsf(c.P[i + 3] - n.P[i - 3]);
}

template <int BUF>
__device__ __forceinline__ float tap_sum(const Line& c, const Line& n, int j)
{
    const int i = j + 3;
    if constexpr (BUF == 0) return c.P[i - 3] + n.P[i + 3];
    if constexpr (BUF == 1) return c.P[i - 2] + n.P[i + 2];
    if constexpr (BUF == 2) return c.P[i - 1] + n.P[i + 1];
    if constexpr (BUF == 3) return c.F[j] + n.B[j];
    if constexpr (BUF == 4) return c.P[i] + n.P[i];
    if constexpr (BUF == 5) return c.B[j] + n.F[j];
    if constexpr (BUF == 6) return c.P[i + 1] + n.P[i - 1];
    if constexpr (BUF == 7) return c.P[i + 2] + n.P[i - 2];
    return c.P[i + 3] + n.P[i - 3];
}

template <int BUF>
constexpr unsigned rank_of()  // P4, P5, P3, P6, P2, P7, P1, P8, P0 -> 1..9 (SangNom2.cpp:211-249)
{
    constexpr unsigned r[9] = {9, 7, 5, 3, 1, 2, 4, 6, 8};
    return r[BUF];
}

template <bool EDGE>
__device__ __forceinline__ void box7(const float (&S)[PXL], float (&Bx)[PXL], const LaneRole& role)
{
    float X[PXL + 6];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        unsigned l = dpp_from_left(bits(S[PXL - 3 + k])), r = dpp_from_right(bits(S[k]));
        if constexpr (EDGE) {  // bitwise selects: a DPP read must not run under a lane mask
            l = bfi(role.first_mask, bits(S[0]), l);        // clamp to column 0
            r = bfi(role.last_mask, bits(S[PXL - 1]), r);  // clamp to column w-1
        }
        X[k] = flt(l);
        X[PXL + 3 + k] = flt(r);
    }
#pragma unroll
    for (int j = 0; j < PXL; ++j) X[3 + j] = S[j];
#pragma unroll
    for (int j = 0; j < PXL; ++j)
        Bx[j] = (((((X[j] + X[j + 1]) + X[j + 2]) + X[j + 3]) + X[j + 4]) + X[j + 5]) + X[j + 6];
}

// PADDED: a plane narrower than its pool stride on a zero-filled pool (sn_config.fresh_pool, see Mode kPadded in
// sn_fused_v3_common.h): costs are zero in the padding columns -- a multiplication by 1 or 0 (costs are finite and
// not negative, so that is exact).
template <int BUF, bool S1, bool PADDED>
__device__ __forceinline__ void buffer_step(float (&A)[PXL], float (&vmin)[PXL], unsigned (&rank)[PXL], const Line& n,
                                            const Line& nn, const LaneRole& role)
{
    float D[PXL], S[PXL], Bx[PXL];
#pragma unroll
    for (int j = 0; j < PXL; ++j) D[j] = S1 ? (PADDED ? cost<BUF>(n, nn, j) * role.inside : cost<BUF>(n, nn, j)) : 0.0f;
#pragma unroll
    for (int j = 0; j < PXL; ++j) S[j] = A[j] + D[j];  // (O[r-1] + D[r]) + D[r+1]
    if (role.edge_wave) box7<true>(S, Bx, role);
    else box7<false>(S, Bx, role);
#pragma unroll
    for (int j = 0; j < PXL; ++j) {
        const float O = Bx[j] * 0.0625f;
        A[j] = O + D[j];
        const bool le = O <= vmin[j];
        vmin[j] = le ? O : vmin[j];
        rank[j] = le ? rank_of<BUF>() : rank[j];
    }
}

constexpr int kRegBuffers = 6;  // buffers 0..5 keep A in VGPRs, 6..8 in LDS between their steps (built without the SLP vectoriser the sweep needs ~212 VGPRs with two: six fit without scratch, seven touch the 256 limit)

template <int NT>
struct Parked {
    uint4* v;  // [4][NT]: the 14 pixels of the parked line (its SangNom values are recomputed: LDS is the scarcer store)
    uint4* a;  // [9 - kRegBuffers][2][NT]
    __device__ __forceinline__ void load_A(int tid, int b, float (&A)[PXL]) const
    {
        const uint4 x = a[((b - kRegBuffers) * 2 + 0) * NT + tid], y = a[((b - kRegBuffers) * 2 + 1) * NT + tid];
        A[0] = flt(x.x); A[1] = flt(x.y); A[2] = flt(x.z); A[3] = flt(x.w);
        A[4] = flt(y.x); A[5] = flt(y.y); A[6] = flt(y.z); A[7] = flt(y.w);
    }
    __device__ __forceinline__ void store_A(int tid, int b, const float (&A)[PXL]) const
    {
        a[((b - kRegBuffers) * 2 + 0) * NT + tid] = make_uint4(bits(A[0]), bits(A[1]), bits(A[2]), bits(A[3]));
        a[((b - kRegBuffers) * 2 + 1) * NT + tid] = make_uint4(bits(A[4]), bits(A[5]), bits(A[6]), bits(A[7]));
    }
    __device__ __forceinline__ void park(int tid, const Line& L) const
    {
        v[0 * NT + tid] = make_uint4(bits(L.P[0]), bits(L.P[1]), bits(L.P[2]), bits(L.P[3]));
        v[1 * NT + tid] = make_uint4(bits(L.P[4]), bits(L.P[5]), bits(L.P[6]), bits(L.P[7]));
        v[2 * NT + tid] = make_uint4(bits(L.P[8]), bits(L.P[9]), bits(L.P[10]), bits(L.P[11]));
        v[3 * NT + tid] = make_uint4(bits(L.P[12]), bits(L.P[13]), 0u, 0u);
    }
    __device__ __forceinline__ void unpark(int tid, Line& L) const
    {
        const uint4 p0 = v[0 * NT + tid], p1 = v[1 * NT + tid], p2 = v[2 * NT + tid], p3 = v[3 * NT + tid];
        L.P[0] = flt(p0.x); L.P[1] = flt(p0.y); L.P[2] = flt(p0.z); L.P[3] = flt(p0.w);
        L.P[4] = flt(p1.x); L.P[5] = flt(p1.y); L.P[6] = flt(p1.z); L.P[7] = flt(p1.w);
        L.P[8] = flt(p2.x); L.P[9] = flt(p2.y); L.P[10] = flt(p2.z); L.P[11] = flt(p2.w);
        L.P[12] = flt(p3.x); L.P[13] = flt(p3.y);
        sangnom_values(L);
    }
};

struct Out {
    u32x4 lo, hi;  // 8 interpolated float pixels
};

template <bool S1, bool PADDED, int NT>
__device__ __forceinline__ Out row_step(float (&A)[kRegBuffers][PXL], const Parked<NT>& pk, int tid, const Line& n,
                                        const Line& nn, const LaneRole& role, float aaf)
{
    float vmin[PXL];
    unsigned rank[PXL];
#pragma unroll
    for (int j = 0; j < PXL; ++j) {
        vmin[j] = __builtin_inff();
        rank[j] = 0u;
    }
    auto run = [&](auto buf) {
        constexpr int B = decltype(buf)::value;
        if constexpr (B < kRegBuffers) {
            buffer_step<B, S1, PADDED>(A[B], vmin, rank, n, nn, role);
        } else {
            float t[PXL];
            pk.load_A(tid, B, t);
            buffer_step<B, S1, PADDED>(t, vmin, rank, n, nn, role);
            pk.store_A(tid, B, t);
        }
    };
    // lowest rung of the ladder first: a later buffer with an equal cost takes over
    run(std::integral_constant<int, 0>{});
    run(std::integral_constant<int, 8>{});
    run(std::integral_constant<int, 1>{});
    run(std::integral_constant<int, 7>{});
    run(std::integral_constant<int, 2>{});
    run(std::integral_constant<int, 6>{});
    run(std::integral_constant<int, 3>{});
    run(std::integral_constant<int, 5>{});
    run(std::integral_constant<int, 4>{});

    Line c;
    pk.unpark(tid, c);
    unsigned v[PXL];
#pragma unroll
    for (int j = 0; j < PXL; ++j) {
        const unsigned wk = vmin[j] > aaf ? 1u : rank[j];  // `buf[4] == minbuf || minbuf > aaf`, SangNom2.cpp:211
        const unsigned m0 = 0u - (wk & 1u);
        const unsigned m1 = 0u - ((wk >> 1) & 1u);
        const unsigned m2 = 0u - ((wk >> 2) & 1u);
        const unsigned m3 = 0u - ((wk >> 3) & 1u);
        // ranks: 1 -> P4; 2 -> P5; 3 -> P3; 4 -> P6; 5 -> P2; 6 -> P7; 7 -> P1; 8 -> P8; 9 -> P0
        const unsigned a01 = bits(tap_sum<4>(c, n, j));
        const unsigned a23 = bfi(m0, bits(tap_sum<3>(c, n, j)), bits(tap_sum<5>(c, n, j)));
        const unsigned a45 = bfi(m0, bits(tap_sum<2>(c, n, j)), bits(tap_sum<6>(c, n, j)));
        const unsigned a67 = bfi(m0, bits(tap_sum<1>(c, n, j)), bits(tap_sum<7>(c, n, j)));
        const unsigned a89 = bfi(m0, bits(tap_sum<0>(c, n, j)), bits(tap_sum<8>(c, n, j)));
        const unsigned b0 = bfi(m1, a23, a01);
        const unsigned b1 = bfi(m1, a67, a45);
        const unsigned c0 = bfi(m2, b1, b0);
        const unsigned r = bfi(m3, a89, c0);
        v[j] = bits(flt(r) * 0.5f);  // (a + b) * 0.5, SangNom2.cpp:52-58
    }
    Out o;
    o.lo.x = v[0]; o.lo.y = v[1]; o.lo.z = v[2]; o.lo.w = v[3];
    o.hi.x = v[4]; o.hi.y = v[5]; o.hi.z = v[6]; o.hi.w = v[7];
    return o;
}

// LDS mailbox: [refresh parity][receiver][slot][72 A registers]; receivers are the left ghosts of waves
// 1 .. NW-1 (side 0) and the right ghosts of waves 0 .. NW-2 (side 1): entry wave * 2 + side - 1.  Sized exactly --
// at 8 waves the workgroup uses 163 584 of the CU's 163 840 bytes of LDS.
template <int NW>
struct Mailbox {
    unsigned* h;
    __device__ __forceinline__ unsigned* at(int par, int wave, int side, int slot) const
    {
        return h + ((par * (2 * NW - 2) + (wave * 2 + side - 1)) * GH + slot) * (kBuffers * PXL);
    }
};

__host__ __device__ constexpr int lds_bytes(int nw)
{
    return (4 + (kBuffers - kRegBuffers) * 2) * 16 * nw * 64 + 2 * (2 * nw - 2) * GH * kBuffers * PXL * 4;
}

template <int NW, bool PADDED>
__global__ void __launch_bounds__(NW * group_of(NW) * 64, 2) k_fused_f32_v3(Args a, float aaf)
{
    extern __shared__ __align__(16) unsigned char lds_raw[];
    constexpr int NT = NW * 64;
    const int sub = __builtin_amdgcn_readfirstlane((int)threadIdx.x) / (NW * 64);
    const int f_raw = (int)blockIdx.x * group_of(NW) + sub;
    const int f = f_raw < a.nframes ? f_raw : a.nframes - 1;  // see sn_fused_u16_v3.hip
    const int tid = (int)threadIdx.x - sub * (NW * 64);
    Parked<NT> parked;
    parked.v = reinterpret_cast<uint4*>(lds_raw + sub * lds_bytes(NW));
    parked.a = parked.v + 4 * NT;
    Mailbox<NW> mb;
    mb.h = reinterpret_cast<unsigned*>(parked.a + (kBuffers - kRegBuffers) * 2 * NT);
    const int wave = tid >> 6;
    const int lane = tid & 63;

    int gl;
    bool ghost;
    if (wave == 0) {
        gl = lane;
        ghost = (NW > 1) && lane >= 64 - GH;
    } else {
        gl = kFirst + kInner * (wave - 1) + (lane - GH);
        ghost = lane < GH || (lane >= 64 - GH && wave < NW - 1);
    }
    const bool live = gl < a.nl;
    const bool real = live && !ghost;
    const int x0 = gl * PXL;
    const int line_w = PADDED ? a.region_w : a.w;  // width of the source / destination plane
    const bool line_live = live && x0 < line_w;
    const bool line_real = real && x0 < line_w;
    LaneRole role;
    role.first = live && gl == 0;
    role.last = live && gl == a.nl - 1;
    role.line_last = line_live && x0 + PXL == line_w;
    role.inside = line_live ? 1.0f : 0.0f;
    role.first_mask = role.first ? 0xffffffffu : 0u;
    role.last_mask = role.last ? 0xffffffffu : 0u;
    role.edge_wave = __builtin_amdgcn_readfirstlane(__any((int)(role.first || role.last || role.line_last)) ? 1 : 0) != 0;

    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint8_t*>(a.src + (int64_t)f * a.src_frame_stride), 0, a.src_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rd =
        __builtin_amdgcn_make_buffer_rsrc(a.dst + (int64_t)f * a.dst_frame_stride, 0, a.dst_bytes, 0x00020000);
    const int vload = line_live ? (x0 > 0 ? 4 * (x0 - 4) : 0) : kOutOfRange;
    const int vstore = line_real ? 4 * x0 : kOutOfRange;
    const int src_step = (a.dh ? 1 : 2) * a.src_pitch;
    const int src_line = (a.dh ? 0 : a.offset) * a.src_pitch;
    const int dst_step = 2 * a.dst_pitch;
    const int dst_line = a.offset * a.dst_pitch;

    auto load_raw = [&](int row_off) {
        Raw q;
#pragma unroll
        for (int k = 0; k < 4; ++k) q.q[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, vload, row_off + 16 * k, 0);
        return q;
    };
    auto keep = [&](int row_off, const Raw& q) {  // GetFrame's field copy: the lane's own 8 pixels
        const bool early = role.first;            // loaded from column 0: own pixels come first
        store_b128(early ? q.q[0] : q.q[1], rd, vstore, row_off);
        store_b128(early ? q.q[1] : q.q[2], rd, vstore, row_off + 16);
    };
    auto put = [&](int row_off, const Out& o) {
        store_b128(o.lo, rd, vstore, row_off);
        store_b128(o.hi, rd, vstore, row_off + 16);
    };

    const int nk = a.nk;
    const int nr = nk - 1;

    Line L0, L1;
    Raw q0 = load_raw(src_line);
    Raw q1 = nk > 1 ? load_raw(src_line + src_step) : q0;
    keep(dst_line, q0);
    if (a.offset == 1) keep(0, q0);  // the line that cannot be interpolated, SangNom2.cpp:386-391
    if (nk > 1) keep(dst_line + dst_step, q1);
    unpack(L0, q0, role);
    unpack(L1, q1, role);
    parked.park(tid, L0);

    // A[1] = O[0] + P[1] = P[1] (pool row 0 is never written: zero, and 0 + x = x)
    float A[kRegBuffers][PXL];
    auto init_buf = [&](auto buf) {
        constexpr int B = decltype(buf)::value;
        float t[PXL];
#pragma unroll
        for (int j = 0; j < PXL; ++j) t[j] = nr > 0 ? (PADDED ? cost<B>(L0, L1, j) * role.inside : cost<B>(L0, L1, j)) : 0.0f;
        if constexpr (B < kRegBuffers) {
#pragma unroll
            for (int j = 0; j < PXL; ++j) A[B][j] = t[j];
        } else {
            parked.store_A(tid, B, t);
        }
    };
    init_buf(std::integral_constant<int, 0>{});
    init_buf(std::integral_constant<int, 1>{});
    init_buf(std::integral_constant<int, 2>{});
    init_buf(std::integral_constant<int, 3>{});
    init_buf(std::integral_constant<int, 4>{});
    init_buf(std::integral_constant<int, 5>{});
    init_buf(std::integral_constant<int, 6>{});
    init_buf(std::integral_constant<int, 7>{});
    init_buf(std::integral_constant<int, 8>{});

    int src_next = src_line + 2 * src_step;
    int dst_keep = dst_line + 2 * dst_step;
    int out_row = dst_line + a.dst_pitch;
    Raw qn = nk > 2 ? load_raw(src_next) : q1;
    src_next += src_step;

    // seam exchange: lanes 60, 61 feed the next wave's left ghosts, lanes 2, 3 the previous wave's right ghosts
    const bool pub_right = lane >= 64 - 2 * GH && lane < 64 - GH && wave < NW - 1;
    const bool pub_left = lane >= GH && lane < 2 * GH && wave > 0;
    const bool recv = ghost && live;
    const int slot = lane < GH ? lane : lane >= 64 - GH ? lane - (64 - GH) : pub_right ? lane - (64 - 2 * GH) : lane - GH;

    TurnTaking turns;
    turns.init(a.turn_shift);
    auto step = [&](int r, Line& n, Line& nn, auto s1_tag) {
        constexpr bool S1 = decltype(s1_tag)::value;
        turns.update();
        Raw qnext = qn;
        if constexpr (S1) {
            unpack(nn, qn, role);  // waits for the line prefetched one row ago
            keep(dst_keep, qn);
            dst_keep += dst_step;
        }
        if constexpr (S1) {
            if (r + 2 <= nr) qnext = load_raw(src_next);
            src_next += src_step;
        }
        const int par = (r / K) & 1;
        if (r > 1 && (r - 1) % K == 0) {
            __syncthreads();
            if (recv) {
                const unsigned* from = mb.at(par, wave, lane < GH ? 0 : 1, slot);
#pragma unroll
                for (int b = 0; b < kRegBuffers; ++b)
#pragma unroll
                    for (int j = 0; j < PXL; ++j) A[b][j] = flt(from[b * PXL + j]);
#pragma unroll
                for (int b = kRegBuffers; b < kBuffers; ++b) {
                    float t[PXL];
#pragma unroll
                    for (int j = 0; j < PXL; ++j) t[j] = flt(from[b * PXL + j]);
                    parked.store_A(tid, b, t);
                }
            }
        }
        put(out_row, row_step<S1, PADDED>(A, parked, tid, n, nn, role, aaf));
        out_row += dst_step;
        if constexpr (S1) parked.park(tid, n);  // n is the next row's c
        if (r < nr && r % K == 0) {
            const int wpar = ((r + 1) / K) & 1;
            if (pub_right || pub_left) {
                unsigned* to = pub_right ? mb.at(wpar, wave + 1, 0, slot) : mb.at(wpar, wave - 1, 1, slot);
#pragma unroll
                for (int b = 0; b < kRegBuffers; ++b)
#pragma unroll
                    for (int j = 0; j < PXL; ++j) to[b * PXL + j] = bits(A[b][j]);
#pragma unroll
                for (int b = kRegBuffers; b < kBuffers; ++b) {
                    float t[PXL];
                    parked.load_A(tid, b, t);
#pragma unroll
                    for (int j = 0; j < PXL; ++j) to[b * PXL + j] = bits(t[j]);
                }
            }
        }
        qn = qnext;
    };
    using T = std::integral_constant<bool, true>;
    using F = std::integral_constant<bool, false>;

    for (int r = 1; r < nr; ++r) {
        step(r, L1, L0, T{});
        L1 = L0;
    }
    if (nr >= 1) {
        step(nr, L1, L0, F{});
    }

    // dst row h-1 := K[nk-1] when the top field is kept, SangNom2.cpp:380-385
    if (a.offset == 0) {
        const Raw q = load_raw(src_line + (nk - 1) * src_step);
        keep((2 * nk - 1) * a.dst_pitch, q);
    }
}

}  // namespace f32

bool fused_f32_plane_ok(int w)
{
    if (w % 32 != 0) return false;
    return v3c::strips_for(w / v3c::PXL) <= f32::kMaxWaves;
}

hipError_t launch_fused_f32_v3(hipStream_t st, const PlaneArgs& p, double threshold, int nframes, int sweep_w)
{
    v3c::Args a{};
    a.src = p.src;
    a.dst = p.dst;
    a.src_frame_stride = p.src_frame_stride;
    a.dst_frame_stride = p.dst_frame_stride;
    a.src_pitch = p.src_pitch;
    a.dst_pitch = p.dst_pitch;
    a.w = sweep_w > 0 ? sweep_w : p.w;  // sweep_w: the pool stride a narrower plane is swept over (PADDED)
    a.region_w = p.w;
    a.nk = p.h_out / 2;
    a.offset = p.offset;
    a.dh = p.dh;
    a.nl = a.w / v3c::PXL;
    a.nvw = v3c::strips_for(a.nl);
    a.nw = a.nvw;
    a.turn_shift = v3c::turn_shift_for(a.nk, a.nw * v3c::group_of(a.nw));
    a.nframes = nframes;
    a.src_bytes = (int)((int64_t)p.src_pitch * p.h_in);
    a.dst_bytes = (int)((int64_t)p.dst_pitch * p.h_out);
    const float aaf = (float)threshold;
    const int g = v3c::group_of(a.nw);
    const int lds = f32::lds_bytes(a.nw) * g;
    hipError_t e = hipSuccess;
#define SN_LAUNCH(NW)                                                                                              \
    case NW:                                                                                                       \
        if (sweep_w > 0) {                                                                                         \
            if (lds > 64 * 1024)                                                                                   \
                e = hipFuncSetAttribute((const void*)f32::k_fused_f32_v3<NW, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
            if (e == hipSuccess) hipLaunchKernelGGL((f32::k_fused_f32_v3<NW, true>), dim3((nframes + g - 1) / g), dim3(NW * g * 64), lds, st, a, aaf); \
        } else {                                                                                                   \
            if (lds > 64 * 1024)                                                                                   \
                e = hipFuncSetAttribute((const void*)f32::k_fused_f32_v3<NW, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
            if (e == hipSuccess) hipLaunchKernelGGL((f32::k_fused_f32_v3<NW, false>), dim3((nframes + g - 1) / g), dim3(NW * g * 64), lds, st, a, aaf); \
        }                                                                                                          \
        break;
    switch (a.nw) {
        SN_LAUNCH(1) SN_LAUNCH(2) SN_LAUNCH(3) SN_LAUNCH(4) SN_LAUNCH(5) SN_LAUNCH(6) SN_LAUNCH(7) SN_LAUNCH(8)
    default: return hipErrorInvalidValue;
    }
#undef SN_LAUNCH
    if (e != hipSuccess) return e;
    return hipGetLastError();
}

}  // namespace sn
