// sn_fused_u8_v4.hip -- the 8-bit sweep for planes on their own (a Y clip, luma-only processing, 4:4:4, isolated planes):
// the same sweep as sn_fused_u8_v3.hip (read that file's header first), re-cut for FOUR waves per SIMD.
//
// Reference semantics: /root/reference/src/SangNom2.cpp:74-124 (prepareBuffers_c), :126-159 (processBuffers_c),
// :161-257 (finalizePlane_c), :361-391 (GetFrame's copies).
//
// Why: profiles/r2_ubench_valu_issue_rates.txt.  On gfx950 a wave issues one VALU instruction per ~4.2 cycles at best, and a
// stream that mixes the plain two-operand forms with packed / DPP / three-operand ones -- every stream this sweep
// can be written as -- costs the SIMD ~4.0 cycles per instruction with TWO waves resident (hardly better than one
// wave alone), 2.7 with three, 2.9 with four and 1.4 with eight.  v3 keeps 16 pixels per lane and needs ~240 VGPRs:
// two waves per SIMD, the worst point of that curve.  Here a lane owns 4 consecutive pixels in each of two column
// strips (8 pixels): all nine buffers' state fits 36 registers, the whole kernel 128, a 3840-wide plane takes 8
// waves, and two workgroups (two frames) give every SIMD four waves.  What it costs: the +-3 halo of the box filter
// and of the taps is amortised over 4 pixels instead of 8 (about 8 % more instructions per pixel), and a ghost zone
// of 2 lanes x 4 pixels stays exact for only floor(8 / 3) = 2 rows, so the waves of a workgroup meet every 2 rows
// instead of every 5.
//
// Everything else is v3's: packed halves (bits 0..15 strip W, bits 16..31 strip W + NW), A = O + D as the only state,
// DPP taps, ghost lanes + LDS mailbox, key minimum for the ladder, buffer descriptors with out-of-range voffsets.
#include <type_traits>

#include "sn_fused_v3_common.h"

// Timing experiments of tools/ only (results are WRONG with any of them): -DSN_V4_X=<bits>, 1 = no seam refresh,
// 2 = six cost buffers instead of nine; -DSN_V4_MINWAVES=<n> = waves per SIMD the register allocation aims at.
#ifndef SN_V4_X
#define SN_V4_X 0
#endif
#ifndef SN_V4_MINWAVES
#define SN_V4_MINWAVES 4
#endif

namespace sn {
namespace v4 {

using v3c::and_or;
using v3c::bfi;
using v3c::dpp_from_left;
using v3c::dpp_from_right;
using v3c::kOutOfRange;
using v3c::pk_absdiff;
using v3c::pk_avg_from_sum;
using v3c::pk_bit_mask;
using v3c::pk_lshr4;
using v3c::pk_min;
using v3c::u32x2;
using v3c::u32x4;
typedef unsigned u32x3 __attribute__((ext_vector_type(3)));

constexpr int PX = 4;                // pixels per lane and per strip
constexpr int GH = 2;                // ghost lanes on each inner side of a strip
constexpr int K = GH * PX / 3;       // rows between two seam refreshes (2)
constexpr int kFirst = 64 - GH;      // real lanes of strip 0
constexpr int kInner = 64 - 2 * GH;  // real lanes of every later strip
constexpr int kMaxWaves = 16;        // 32 strips: 7680 columns
constexpr unsigned kLo = 0x0000ffffu, kHi = 0xffff0000u, kByte = 0x00ff00ffu;

struct Line {
    unsigned P[PX + 6];  // P[i] = pixel x0 - 3 + i of both strips
    unsigned FB[PX];     // the two SangNom values per pixel (calculateSangNom, SangNom2.cpp:60-65): F | B << 8 per half
    __device__ __forceinline__ unsigned F(int j) const { return FB[j] & kByte; }
    __device__ __forceinline__ unsigned B(int j) const { return (FB[j] >> 8) & kByte; }
};

struct RawHalf {  // left dword, own 4 bytes, right dword of one strip
    uint32_t l, m, r;
};
struct Raw {
    RawHalf h[2];
};

struct LaneRole {
    bool edge_wave;           // wave holds column 0 or column w-1: clamps needed
    unsigned first_mask;      // 0xffff in the half that owns column 0 (else 0)
    unsigned last_mask;       // ... that owns column w-1
    unsigned key_mask;        // 0x0ff00ff0 in a VGPR (operand of the and-or that forms the ladder keys)
};

__device__ __forceinline__ RawHalf load_half(__amdgpu_buffer_rsrc_t rs, int voff, int soff)
{
    const u32x3 q = __builtin_amdgcn_raw_buffer_load_b96(rs, voff, soff, 0);
    return RawHalf{q.x, q.y, q.z};
}

// byte k of the lo word -> bits 0..7, byte k of the hi word -> bits 16..23
__device__ __forceinline__ unsigned pair_byte(uint32_t hi_word, uint32_t lo_word, int k)
{
    return __builtin_amdgcn_perm(hi_word, lo_word, 0x0c040c00u + (unsigned)k * 0x00010001u);
}

__device__ __forceinline__ void unpack(Line& L, Raw q, const LaneRole& role)
{
    if (role.edge_wave) {  // loadPixel's clamp (SangNom2.cpp:25-34) for the two image-edge lanes
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (role.first_mask & (h ? kHi : kLo)) {  // loaded from column 0: dwords are one slot early
                q.h[h].r = q.h[h].m;
                q.h[h].m = q.h[h].l;
                q.h[h].l = (q.h[h].m & 0xff) * 0x01010101u;
            }
            if (role.last_mask & (h ? kHi : kLo)) q.h[h].r = (q.h[h].m >> 24) * 0x01010101u;
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        L.P[k] = pair_byte(q.h[1].l, q.h[0].l, k + 1);
        L.P[PX + 3 + k] = pair_byte(q.h[1].r, q.h[0].r, k);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) L.P[3 + k] = pair_byte(q.h[1].m, q.h[0].m, k);
    // F = ((4a + 5b - c) >> 3) mod 256, B = ((4c + 5b - a) >> 3) mod 256 with a bias of 2048 per half
    unsigned Q4[PX + 2], M[PX + 2];  // positions 2 .. PX+3
#pragma unroll
    for (int i = 0; i < PX + 2; ++i) {
        const unsigned p = L.P[i + 2];
        const unsigned p2 = p + p;
        Q4[i] = p2 + p2;
        M[i] = 0x08000800u - p;
    }
#pragma unroll
    for (int j = 0; j < PX; ++j) {
        const unsigned q5 = Q4[j + 1] + L.P[j + 3];
        const unsigned f = ((Q4[j] + q5 + M[j + 2]) >> 3) & kByte;
        const unsigned b = ((Q4[j + 2] + q5 + M[j]) << 5) & 0xff00ff00u;  // (x >> 3 & 255) << 8
        L.FB[j] = f | b;
    }
}

// Stage 1, buffer BUF, packed position j, pair (c, n); Buffers enum of SangNom2.h:8-20.
template <int BUF>
__device__ __forceinline__ unsigned cost(const Line& c, const Line& n, int j)
{
    const int i = j + 3;
    if constexpr (BUF == 0) return pk_absdiff(c.P[i - 3], n.P[i + 3]);
    if constexpr (BUF == 1) return pk_absdiff(c.P[i - 2], n.P[i + 2]);
    if constexpr (BUF == 2) return pk_absdiff(c.P[i - 1], n.P[i + 1]);
    if constexpr (BUF == 3) return pk_absdiff(c.F(j), n.B(j));  // |forwardSangNom1 - forwardSangNom2|
    if constexpr (BUF == 4) return pk_absdiff(c.P[i], n.P[i]);
    if constexpr (BUF == 5) return pk_absdiff(c.B(j), n.F(j));  // |backwardSangNom1 - backwardSangNom2|
    if constexpr (BUF == 6) return pk_absdiff(c.P[i + 1], n.P[i - 1]);
    if constexpr (BUF == 7) return pk_absdiff(c.P[i + 2], n.P[i - 2]);
    return pk_absdiff(c.P[i + 3], n.P[i - 3]);
}

// Stage 3: a + b of the candidate that belongs to buffer BUF (SangNom2.cpp:214-249).
template <int BUF>
__device__ __forceinline__ unsigned tap_sum(const Line& c, const Line& n, int j)
{
    const int i = j + 3;
    if constexpr (BUF == 0) return c.P[i - 3] + n.P[i + 3];
    if constexpr (BUF == 1) return c.P[i - 2] + n.P[i + 2];
    if constexpr (BUF == 2) return c.P[i - 1] + n.P[i + 1];
    if constexpr (BUF == 3) return c.F(j) + n.B(j);
    if constexpr (BUF == 4) return c.P[i] + n.P[i];
    if constexpr (BUF == 5) return c.B(j) + n.F(j);
    if constexpr (BUF == 6) return c.P[i + 1] + n.P[i - 1];
    if constexpr (BUF == 7) return c.P[i + 2] + n.P[i - 2];
    return c.P[i + 3] + n.P[i - 3];
}

// rank of buffer BUF in the reference's ladder: P4, P5, P3, P6, P2, P7, P1, P8, P0 -> 1..9
template <int BUF>
constexpr unsigned rank_of()
{
    constexpr unsigned r[9] = {9, 7, 5, 3, 1, 2, 4, 6, 8};
    return r[BUF] * 0x00010001u;
}

template <bool EDGE>
__device__ __forceinline__ void box7(const unsigned (&S)[PX], unsigned (&Bx)[PX], const LaneRole& role)
{
    unsigned L[3], R[3];
    if constexpr (EDGE) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            L[k] = bfi(role.first_mask, S[0], dpp_from_left(S[PX - 3 + k]));   // clamp to column 0
            R[k] = bfi(role.last_mask, S[PX - 1], dpp_from_right(S[k]));       // clamp to column w-1
        }
    }
    auto X = [&](int i) -> unsigned {
        if (i < 0) return EDGE ? L[i + 3] : dpp_from_left(S[PX + i]);
        if (i >= PX) return EDGE ? R[i - PX] : dpp_from_right(S[i - PX]);
        return S[i];
    };
    Bx[0] = S[0] + S[1] + S[2] + S[3] + X(-1) + X(-2) + X(-3);
#pragma unroll
    for (int j = 0; j + 1 < PX; ++j) Bx[j + 1] = Bx[j] - X(j - 3) + X(j + 4);
}

// S1: the costs of row r+1 come from the lines (n, nn); otherwise they are zero (row bh is never written).
template <int BUF, bool S1>
__device__ __forceinline__ void buffer_step(unsigned (&A)[PX], unsigned (&kmin)[PX], const Line& n, const Line& nn, const LaneRole& role)
{
    unsigned D[PX], S[PX], Bx[PX];
#pragma unroll
    for (int j = 0; j < PX; ++j) D[j] = S1 ? cost<BUF>(n, nn, j) : 0u;
#pragma unroll
    for (int j = 0; j < PX; ++j) S[j] = A[j] + D[j];
    if (role.edge_wave) box7<true>(S, Bx, role);
    else box7<false>(S, Bx, role);
#pragma unroll
    for (int j = 0; j < PX; ++j) {
        const unsigned key = and_or(Bx[j], role.key_mask, rank_of<BUF>());  // (sum / 16 mod 256) << 4 | rank
        A[j] = pk_lshr4(key) + D[j];  // O + D[r+1]; (sum / 16) wraps to uint8_t, SangNom2.cpp:152
        kmin[j] = pk_min(kmin[j], key);
    }
}

struct Out {
    uint32_t lo, hi;  // 4 interpolated bytes of each strip
};

// The line above the pair being interpolated (c = K[r-1]) is needed only by stage 3, after the nine buffer steps: parked
// in thread-private LDS slots meanwhile (no barrier involved).
struct Parked {
    uint4* v;  // [4][NT]
    int nt;
    __device__ __forceinline__ void park(int tid, const Line& L) const
    {
        v[0 * nt + tid] = make_uint4(L.P[0], L.P[1], L.P[2], L.P[3]);
        v[1 * nt + tid] = make_uint4(L.P[4], L.P[5], L.P[6], L.P[7]);
        v[2 * nt + tid] = make_uint4(L.P[8], L.P[9], L.FB[0], L.FB[1]);
        v[3 * nt + tid] = make_uint4(L.FB[2], L.FB[3], 0u, 0u);
    }
    __device__ __forceinline__ void unpark(int tid, Line& L) const
    {
        const uint4 a = v[0 * nt + tid], b = v[1 * nt + tid], c = v[2 * nt + tid], d = v[3 * nt + tid];
        L.P[0] = a.x; L.P[1] = a.y; L.P[2] = a.z; L.P[3] = a.w;
        L.P[4] = b.x; L.P[5] = b.y; L.P[6] = b.z; L.P[7] = b.w;
        L.P[8] = c.x; L.P[9] = c.y; L.FB[0] = c.z; L.FB[1] = c.w;
        L.FB[2] = d.x; L.FB[3] = d.y;
    }
};

template <bool S1>
__device__ __forceinline__ Out row_step(unsigned (&A)[kBuffers][PX], const Parked& pk, int tid, const Line& n, const Line& nn,
                                        const LaneRole& role, unsigned thr_key)
{
    unsigned kmin[PX];
#pragma unroll
    for (int j = 0; j < PX; ++j) kmin[j] = thr_key;  // the `minBuf > aaf` arm: cost aaf + 1, rank 0
    buffer_step<0, S1>(A[0], kmin, n, nn, role);
    __builtin_amdgcn_sched_barrier(0);  // one buffer at a time: interleaving them costs registers
    buffer_step<1, S1>(A[1], kmin, n, nn, role);
    __builtin_amdgcn_sched_barrier(0);  // one buffer at a time: interleaving them costs registers
    buffer_step<2, S1>(A[2], kmin, n, nn, role);
    __builtin_amdgcn_sched_barrier(0);  // one buffer at a time: interleaving them costs registers
    buffer_step<3, S1>(A[3], kmin, n, nn, role);
    __builtin_amdgcn_sched_barrier(0);  // one buffer at a time: interleaving them costs registers
    buffer_step<4, S1>(A[4], kmin, n, nn, role);
    __builtin_amdgcn_sched_barrier(0);  // one buffer at a time: interleaving them costs registers
    buffer_step<5, S1>(A[5], kmin, n, nn, role);
    __builtin_amdgcn_sched_barrier(0);  // one buffer at a time: interleaving them costs registers
    if constexpr (!(SN_V4_X & 2)) buffer_step<6, S1>(A[6], kmin, n, nn, role);
    __builtin_amdgcn_sched_barrier(0);  // one buffer at a time: interleaving them costs registers
    if constexpr (!(SN_V4_X & 2)) buffer_step<7, S1>(A[7], kmin, n, nn, role);
    __builtin_amdgcn_sched_barrier(0);  // one buffer at a time: interleaving them costs registers
    if constexpr (!(SN_V4_X & 2)) buffer_step<8, S1>(A[8], kmin, n, nn, role);
    __builtin_amdgcn_sched_barrier(0);  // one buffer at a time: interleaving them costs registers

    // winner's rank -> tap sum -> average
    Line c;
    pk.unpark(tid, c);
    unsigned v[PX];
#pragma unroll
    for (int j = 0; j < PX; ++j) {
        const unsigned wk = kmin[j];
        const unsigned m0 = pk_bit_mask<0>(wk), m1 = pk_bit_mask<1>(wk), m2 = pk_bit_mask<2>(wk), m3 = pk_bit_mask<3>(wk);
        // ranks: 0,1 -> P4; 2 -> P5; 3 -> P3; 4 -> P6; 5 -> P2; 6 -> P7; 7 -> P1; 8 -> P8; 9 -> P0
        const unsigned a01 = tap_sum<4>(c, n, j);
        const unsigned a23 = bfi(m0, tap_sum<3>(c, n, j), tap_sum<5>(c, n, j));
        const unsigned a45 = bfi(m0, tap_sum<2>(c, n, j), tap_sum<6>(c, n, j));
        const unsigned a67 = bfi(m0, tap_sum<1>(c, n, j), tap_sum<7>(c, n, j));
        const unsigned a89 = bfi(m0, tap_sum<0>(c, n, j), tap_sum<8>(c, n, j));
        const unsigned b0 = bfi(m1, a23, a01);
        const unsigned b1 = bfi(m1, a67, a45);
        const unsigned c0 = bfi(m2, b1, b0);
        const unsigned r = bfi(m3, a89, c0);
        v[j] = pk_avg_from_sum(r);  // (a + b + 1) >> 1 in both halves
    }
    // v[j] = lo-strip byte | hi-strip byte << 16  ->  four bytes per strip
    const unsigned t01 = __builtin_amdgcn_perm(v[1], v[0], 0x06020400u);  // [v0.b0, v1.b0, v0.b2, v1.b2]
    const unsigned t23 = __builtin_amdgcn_perm(v[3], v[2], 0x06020400u);
    Out o;
    o.lo = __builtin_amdgcn_perm(t23, t01, 0x05040100u);
    o.hi = __builtin_amdgcn_perm(t23, t01, 0x07060302u);
    return o;
}

// LDS mailbox, receiver-ready (as in v3): word [refresh parity][wave 0..NW][side][slot][36][2 halves] 16-bit entries; what
// ghost lane `slot` on `side` of wave W loads into packed A register i.  Inside the plane a seam register goes whole to one
// ghost lane of the neighbouring wave; at the wrap seam (strip NW-1 | strip NW) a half crosses into the other half of its
// receiver and is stored on its own.
template <int NW>
struct Mailbox {
    unsigned short* h;
    __device__ __forceinline__ unsigned short* at(int par, int wave, int side, int slot) const
    {
        return h + ((((par * (NW + 1) + wave) * 2 + side) * GH + slot) * (kBuffers * PX)) * 2;
    }
};

__host__ __device__ constexpr int lds_bytes(int nw) { return 4 * 16 * nw * 64 + 2 * (nw + 1) * 2 * GH * kBuffers * PX * 4; }

// frames per workgroup: narrow planes share a workgroup so that it has 8 waves (two workgroups per CU, four waves per SIMD)
__host__ __device__ constexpr int group_of(int nw) { return nw == 1 ? 8 : nw == 2 ? 4 : nw <= 4 ? 2 : 1; }

template <int NW>
__global__ void __launch_bounds__(NW * group_of(NW) * 64, SN_V4_MINWAVES) k_fused_u8_v4(v3c::Args a)
{
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int sub = __builtin_amdgcn_readfirstlane((int)threadIdx.x) / (NW * 64);
    const int f_raw = (int)blockIdx.x * group_of(NW) + sub;
    const int f = f_raw < a.nframes ? f_raw : a.nframes - 1;  // past the end: repeat the last frame (same stores, harmless)
    const int tid = (int)threadIdx.x - sub * (NW * 64);
    Parked parked;
    parked.v = reinterpret_cast<uint4*>(lds_raw + sub * lds_bytes(NW));
    parked.nt = NW * 64;
    Mailbox<NW> mb;
    mb.h = reinterpret_cast<unsigned short*>(parked.v + 4 * NW * 64);
    const int wave = tid >> 6;
    const int lane = tid & 63;
    const int nvw = a.nvw;

    // per-half lane roles: virtual wavefront vw = wave + half * NW (strips in column order)
    int x0[2];
    bool live[2], real[2], ghost[2];
    LaneRole role;
    role.first_mask = 0;
    role.last_mask = 0;
    role.key_mask = 0x0ff00ff0u;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int vw = wave + h * NW;
        int gl;
        bool g;
        if (vw == 0) {
            gl = lane;
            g = (nvw > 1) && lane >= 64 - GH;
        } else {
            gl = kFirst + kInner * (vw - 1) + (lane - GH);
            g = lane < GH || (lane >= 64 - GH && vw < nvw - 1);
        }
        live[h] = vw < nvw && gl < a.nl;
        ghost[h] = g;
        real[h] = live[h] && !g;
        x0[h] = gl * PX;
        if (live[h] && gl == 0) role.first_mask |= h ? kHi : kLo;
        if (live[h] && gl == a.nl - 1) role.last_mask |= h ? kHi : kLo;
    }
    role.edge_wave = __builtin_amdgcn_readfirstlane(__any((int)(role.first_mask | role.last_mask)) ? 1 : 0) != 0;

    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint8_t*>(a.src + (int64_t)f * a.src_frame_stride), 0, a.src_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rd =
        __builtin_amdgcn_make_buffer_rsrc(a.dst + (int64_t)f * a.dst_frame_stride, 0, a.dst_bytes, 0x00020000);
    int vload[2], vstore[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        vload[h] = live[h] ? (x0[h] > 0 ? x0[h] - 4 : 0) : kOutOfRange;
        vstore[h] = real[h] ? x0[h] : kOutOfRange;
    }
    const int src_step = (a.dh ? 1 : 2) * a.src_pitch;        // kept line k -> k + 1
    const int src_line = (a.dh ? 0 : a.offset) * a.src_pitch;  // kept line 0
    const int dst_step = 2 * a.dst_pitch;
    const int dst_line = a.offset * a.dst_pitch;  // kept line 0 in dst

    auto load_raw = [&](int row_off) {
        Raw q;
        q.h[0] = load_half(rs, vload[0], row_off);
        q.h[1] = load_half(rs, vload[1], row_off);
        return q;
    };
    auto own = [&](const Raw& q, int h) {  // the 4 bytes the lane owns (the column-0 lane loaded them first)
        return (role.first_mask & (h ? kHi : kLo)) ? q.h[h].l : q.h[h].m;
    };
    auto keep = [&](int row_off, const Raw& q) {  // GetFrame's field copy, SangNom2.cpp:365 / :376
        __builtin_amdgcn_raw_buffer_store_b32(own(q, 0), rd, vstore[0], row_off, 0);
        __builtin_amdgcn_raw_buffer_store_b32(own(q, 1), rd, vstore[1], row_off, 0);
    };
    auto put = [&](int row_off, const Out& o) {
        __builtin_amdgcn_raw_buffer_store_b32(o.lo, rd, vstore[0], row_off, 0);
        __builtin_amdgcn_raw_buffer_store_b32(o.hi, rd, vstore[1], row_off, 0);
    };

    const int nk = a.nk;
    const int nr = nk - 1;
    const unsigned thr_key = (unsigned)((a.thr + 1) << 4) * 0x00010001u;

    Line L0, L1;
    Raw q0 = load_raw(src_line);
    Raw q1 = nk > 1 ? load_raw(src_line + src_step) : q0;
    keep(dst_line, q0);
    if (a.offset == 1) keep(0, q0);  // the line that cannot be interpolated, SangNom2.cpp:386-391
    if (nk > 1) keep(dst_line + dst_step, q1);
    unpack(L0, q0, role);
    unpack(L1, q1, role);
    parked.park(tid, L0);  // c of row 1

    // A[1] = O[0] + P[1] = P[1] (pool row 0 is never written: zero); P[1] = stage-1 costs of the first line pair
    unsigned A[kBuffers][PX];
    auto init_A = [&](auto buf) {
        constexpr int B = decltype(buf)::value;
#pragma unroll
        for (int j = 0; j < PX; ++j) A[B][j] = nr > 0 ? cost<B>(L0, L1, j) : 0u;
    };
    init_A(std::integral_constant<int, 0>{});
    init_A(std::integral_constant<int, 1>{});
    init_A(std::integral_constant<int, 2>{});
    init_A(std::integral_constant<int, 3>{});
    init_A(std::integral_constant<int, 4>{});
    init_A(std::integral_constant<int, 5>{});
    init_A(std::integral_constant<int, 6>{});
    init_A(std::integral_constant<int, 7>{});
    init_A(std::integral_constant<int, 8>{});

    int src_next = src_line + 2 * src_step;
    int dst_keep = dst_line + 2 * dst_step;
    int out_row = dst_line + a.dst_pitch;
    Raw qn = nk > 2 ? load_raw(src_next) : q1;
    src_next += src_step;

    // Seam exchange roles.  Lanes 60, 61 are the right seam lanes and lanes 2, 3 the left seam lanes of BOTH virtual
    // wavefronts of this wave, so they publish whole packed registers; the ghost lanes (0, 1 and 62, 63) take one half
    // from each of two published registers.
    const bool pub_right = lane >= 64 - 2 * GH && lane < 64 - GH;
    const bool pub_left = lane >= GH && lane < 2 * GH;
    const bool recv_left = lane < GH;         // left ghosts of both strips
    const bool recv_right = lane >= 64 - GH;  // right ghosts of both strips
    const unsigned ghost_mask = (ghost[0] && live[0] ? kLo : 0u) | (ghost[1] && live[1] ? kHi : 0u);
    const int slot = recv_left ? lane : recv_right ? lane - (64 - GH) : pub_right ? lane - (64 - 2 * GH) : lane - GH;

    Out pending{};
    // One pool row r: n = K[r], nn = K[r+1] (S1: the pair exists), c = K[r-1] parked.
    auto step = [&](int r, Line& n, Line& nn, auto s1_tag) {
        constexpr bool HAS_NEXT = decltype(s1_tag)::value;
        Raw qnext = qn;
        if constexpr (HAS_NEXT) {
            unpack(nn, qn, role);  // waits for the line prefetched one row ago
            keep(dst_keep, qn);
            dst_keep += dst_step;
        }
        if (r > 1) {
            put(out_row, pending);
            out_row += dst_step;
        }
        if constexpr (HAS_NEXT) {
            if (r + 2 <= nr) qnext = load_raw(src_next);  // prefetch K[r+2]
            src_next += src_step;
        }
        const int par = (r / K) & 1;
        if (r > 1 && (r - 1) % K == 0 && !(SN_V4_X & 1)) {
            __syncthreads();
            if (recv_left || recv_right) {
                const uint4* from = reinterpret_cast<const uint4*>(mb.at(par, wave, recv_left ? 0 : 1, slot));
#pragma unroll
                for (int b = 0; b < kBuffers; ++b) {
                    const uint4 g = from[b];
                    A[b][0] = bfi(ghost_mask, g.x, A[b][0]);
                    A[b][1] = bfi(ghost_mask, g.y, A[b][1]);
                    A[b][2] = bfi(ghost_mask, g.z, A[b][2]);
                    A[b][3] = bfi(ghost_mask, g.w, A[b][3]);
                }
            }
        }
        pending = row_step<HAS_NEXT>(A, parked, tid, n, nn, role, thr_key);
        if constexpr (HAS_NEXT) parked.park(tid, n);  // n is the next row's c
        if (r < nr && r % K == 0 && !(SN_V4_X & 1)) {
            const int wpar = ((r + 1) / K) & 1;
            if (pub_right || pub_left) {
                const bool whole = pub_right ? wave < NW - 1 : wave > 0;
                if (whole) {
                    uint4* to = reinterpret_cast<uint4*>(pub_right ? mb.at(wpar, wave + 1, 0, slot) : mb.at(wpar, wave - 1, 1, slot));
#pragma unroll
                    for (int b = 0; b < kBuffers; ++b) to[b] = make_uint4(A[b][0], A[b][1], A[b][2], A[b][3]);
                } else {
                    // wave NW-1, right seam lanes: low half (strip NW-1) -> high half of wave 0's left ghosts;
                    // wave 0, left seam lanes: high half (strip NW) -> low half of wave NW-1's right ghosts
                    unsigned short* to = pub_right ? mb.at(wpar, 0, 0, slot) + 1 : mb.at(wpar, NW - 1, 1, slot);
                    const int sh = pub_right ? 0 : 16;
#pragma unroll
                    for (int b = 0; b < kBuffers; ++b) {
#pragma unroll
                        for (int j = 0; j < PX; ++j) to[(b * PX + j) * 2] = (unsigned short)(A[b][j] >> sh);
                    }
                }
            }
        }
        qn = qnext;
    };
    using T = std::integral_constant<bool, true>;
    using F = std::integral_constant<bool, false>;

    // L1 = K[r] (n), L0 is reused for K[r+1] (nn); c lives in LDS.  Rows 1 .. nr-1 have a following line pair, row nr
    // does not (its next costs are zero: pool row bh is never written).
    for (int r = 1; r < nr; ++r) {
        step(r, L1, L0, T{});
        L1 = L0;
    }
    if (nr >= 1) {
        step(nr, L1, L0, F{});
        put(out_row, pending);
    }

    // dst row h-1 := K[nk-1] when the top field is kept, SangNom2.cpp:380-385
    if (a.offset == 0) {
        const Raw q = load_raw(src_line + (nk - 1) * src_step);
        keep((2 * nk - 1) * a.dst_pitch, q);
    }
}

static int strips_for(int nl) { return nl <= 64 ? 1 : 1 + (nl - kFirst + kInner - 1) / kInner; }

}  // namespace v4

bool fused_v4_plane_ok(int w)
{
    if (w % 32 != 0) return false;
    return v4::strips_for(w / v4::PX) <= 2 * v4::kMaxWaves;
}

int fused_v4_waves(int w) { return (v4::strips_for(w / v4::PX) + 1) / 2; }

hipError_t launch_fused_u8_v4(hipStream_t st, const PlaneArgs& p, double threshold, int nframes)
{
    v3c::Args a{};
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
    a.nl = a.w / v4::PX;
    a.nvw = v4::strips_for(a.nl);
    a.nw = (a.nvw + 1) / 2;
    a.src_bytes = (int)((int64_t)p.src_pitch * p.h_in);
    a.dst_bytes = (int)((int64_t)p.dst_pitch * p.h_out);
    a.nframes = nframes;
    const int g = v4::group_of(a.nw);
    const int lds = v4::lds_bytes(a.nw) * g;
    hipError_t e = hipSuccess;
#define SN_LAUNCH(NW)                                                                                              \
    case NW:                                                                                                       \
        if (lds > 64 * 1024)                                                                                       \
            e = hipFuncSetAttribute((const void*)v4::k_fused_u8_v4<NW>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
        if (e == hipSuccess) hipLaunchKernelGGL((v4::k_fused_u8_v4<NW>), dim3((nframes + g - 1) / g), dim3(NW * g * 64), lds, st, a); \
        break;
    switch (a.nw) {
        SN_LAUNCH(1) SN_LAUNCH(2) SN_LAUNCH(3) SN_LAUNCH(4) SN_LAUNCH(5) SN_LAUNCH(6) SN_LAUNCH(7) SN_LAUNCH(8)
        SN_LAUNCH(9) SN_LAUNCH(10) SN_LAUNCH(11) SN_LAUNCH(12) SN_LAUNCH(13) SN_LAUNCH(14) SN_LAUNCH(15) SN_LAUNCH(16)
    default: return hipErrorInvalidValue;
    }
#undef SN_LAUNCH
    if (e != hipSuccess) return e;
    return hipGetLastError();
}

}  // namespace sn
