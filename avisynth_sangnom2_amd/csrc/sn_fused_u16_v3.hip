// sn_fused_u16_v3.hip -- the fused sweep for 9..16-bit samples (uint16 containers).
//
// Same algorithm, exactness argument and structure as sn_fused_u8_v3.hip (read its header): one
// workgroup sweeps one plane top to bottom, a lane owns 8 consecutive pixels of a row, the nine cost
// buffers live in registers / thread-private LDS as A[r] = O[r-1] + D[r], ghost lanes keep wave seams
// exact with one barrier every 5 rows, subsampled chroma is coupled to the luma sweep through scratch
// pools (Mode).  16-bit sums need up to 21 bits, so a register holds ONE pixel
// here: one strip per wave, up to 8 waves = 3840 pixels.
//
// Reference semantics: /root/reference/src/SangNom2.cpp:60-65 (wrap to uint16_t, arithmetic >> 3),
// :74-124, :126-159 (sum / 16 wraps to uint16_t), :161-257, :361-391.
#include <stdlib.h>

#include <type_traits>

#include "sn_fused_v3_common.h"

namespace sn {
namespace w16 {

#ifdef SN_WAVE_TIMING  // tools/wave_timing.py: per mode and wave index, shader-clock ticks in the kernel and at its barriers
__device__ unsigned long long sn_wave_ticks[5][8][2];
#define SN_SYNC()                                                      \
    do {                                                               \
        const unsigned long long wt_a = __builtin_amdgcn_s_memtime();  \
        __syncthreads();                                               \
        wt_barrier += __builtin_amdgcn_s_memtime() - wt_a;             \
    } while (0)
#define SN_WT_FLUSH()                                                                                         \
    do {                                                                                                      \
        if (lane == 0) {                                                                                      \
            atomicAdd(&sn_wave_ticks[MODE][wave][0], __builtin_amdgcn_s_memtime() - wt_start);                \
            atomicAdd(&sn_wave_ticks[MODE][wave][1], wt_barrier);                                             \
        }                                                                                                     \
    } while (0)
#else
#define SN_SYNC() __syncthreads()
#define SN_WT_FLUSH() do {} while (0)
#endif

using namespace v3c;
constexpr int kMaxWaves = 8;
constexpr unsigned kVal = 0xffffu;

__device__ __forceinline__ unsigned absdiff(unsigned a, unsigned b)  // both < 65536
{
    return __builtin_amdgcn_sad_u16(a, b, 0u);
}
__device__ __forceinline__ unsigned umin(unsigned a, unsigned b) { return a < b ? a : b; }

struct Line {
    unsigned P[PXL + 6];  // P[i] = pixel x0 - 3 + i (edge-clamped)
    unsigned FB[PXL];     // F | B << 16: the two SangNom values (calculateSangNom, SangNom2.cpp:60-65)
    __device__ __forceinline__ unsigned F(int j) const { return FB[j] & kVal; }
    __device__ __forceinline__ unsigned B(int j) const { return FB[j] >> 16; }
};

// The same with F and B in registers of their own (no extraction per use): for the sweeps whose stage 3 gathers its two
// taps from LDS (planes on their own) -- there a line in this form only serves stage 1 and the kernel has the registers.
struct WideLine {
    unsigned P[PXL + 6];
    unsigned Fv[PXL], Bv[PXL];
    __device__ __forceinline__ unsigned F(int j) const { return Fv[j]; }
    __device__ __forceinline__ unsigned B(int j) const { return Bv[j]; }
};

struct Raw {  // bytes [2(x0-4), 2(x0+12)): pixels x0-4 .. x0+11 as 8 dwords
    u32x4 a, b;
};

// ---- Stage 3 as a gather (planes on their own) ------------------------------------------------------------------------
// The ladder's winner picks ONE sample of the line above and ONE of the line below (SangNom2.cpp:214-249: c(k) and n(-k),
// or the two SangNom values); the result is their rounded average.  The 8-bit sweep does that with byte permutes over
// registers; a 16-bit pixel pair's candidates are 16 bytes, more than v_perm_b32 sees, so here every lane READS its two
// taps from LDS with a computed address.  A line is kept in LDS as 32 halfwords per thread -- 0..15 the pixels x0-4 ..
// x0+11 exactly as loaded (edge-clamped), 16..23 the forward, 24..31 the backward SangNom values -- for the two rows that
// use it (a ring of three lines, each thread reads what it wrote; 17 dwords per thread so that the lanes spread over the
// banks).  The winner's rank code (P4 / threshold 0, P5 1, P3 2, P6 3, P2 4, P7 5, P1 6, P8 7, P0 12) goes through two
// byte tables (v_perm_b32; selector 12 reads 0x00, the tables are stored XOR 1 / XOR 7 so that 0x00 means P0's tap):
// halfword index of pixel j's upper tap = j + (4 + k | 16 | 24), of its lower tap = j + (4 - k | 24 | 16).
// 6 instructions and two ds_read_u16 per pixel where the select tree took 31 instructions.
constexpr int kGatherDwords = 17;  // per thread and line
constexpr unsigned kLutCLo = 0x04111905u, kLutCHi = 0x06030702u;  // code -> (4 + k, c -> B 24, c -> F 16) ^ 1
constexpr unsigned kLutNLo = 0x041f1703u, kLutNHi = 0x06010502u;  // code -> (4 - k, n -> F 16, n -> B 24) ^ 7

struct LaneRole {
    bool edge_wave;  // wave holds the first or last column of the sweep or of the source lines
    bool first;      // lane owns column 0
    bool last;       // lane owns the last column of the sweep
    bool line_last;  // lane owns the last column of the source lines (kChroma: region_w - 1)
    bool inside;     // kChroma: the lane's columns lie inside the chroma region
    unsigned first_mask, last_mask;  // all ones where first / last
    unsigned key_mask;               // 0xffff0 in a VGPR (operand of the and-or that forms the ladder keys)
};

struct GatherLine {  // what park_gather() writes: 8 dwords of pixels, 4 of F pairs, 4 of B pairs
    unsigned d[8], f[4], b[4];
};

template <class LineT>
__device__ __forceinline__ void unpack(LineT& L, const Raw& q, const LaneRole& role, GatherLine* G = nullptr)
{
    // dword i of (a, b) holds pixels x0 - 4 + 2i (low half) and x0 - 3 + 2i; the lane that owns column 0
    // loaded from column 0 instead, so its dwords are two slots early
    unsigned d[8] = {q.a.x, q.a.y, q.a.z, q.a.w, q.b.x, q.b.y, q.b.z, q.b.w};
    if (role.edge_wave && role.first) {
#pragma unroll
        for (int i = 7; i >= 2; --i) d[i] = d[i - 2];
    }
    if (role.edge_wave) {  // loadPixel's clamp, SangNom2.cpp:25-34, on the dwords (the gather reads them as they are)
        if (role.first) d[0] = d[1] = (d[2] & kVal) * 0x10001u;        // pixels x0-4 .. x0-1 := pixel x0
        if (role.line_last) d[6] = d[7] = (d[5] >> 16) * 0x10001u;     // pixels x0+8 .. x0+11 := pixel x0+7
    }
    L.P[0] = d[0] >> 16;
#pragma unroll
    for (int i = 1; i < 7; ++i) {
        L.P[2 * i - 1] = d[i] & kVal;
        L.P[2 * i] = d[i] >> 16;
    }
    L.P[13] = d[7] & kVal;
    // F = ((4a + 5b - c) >> 3) mod 65536, B likewise mirrored; a bias of 8 * 65536 keeps the sum positive
    // and drops out of the result
    unsigned fv[PXL], bv[PXL];
#pragma unroll
    for (int j = 0; j < PXL; ++j) {
        const unsigned a = L.P[j + 2], b = L.P[j + 3], c = L.P[j + 4];
        const unsigned x5 = 4 * b + b + 0x80000u;
        fv[j] = ((4 * a + x5 - c) >> 3) & kVal;
        bv[j] = ((4 * c + x5 - a) >> 3) & kVal;
        if constexpr (std::is_same<LineT, WideLine>::value) {
            L.Fv[j] = fv[j];
            L.Bv[j] = bv[j];
        } else {
            L.FB[j] = fv[j] | (bv[j] << 16);
        }
    }
    if (G) {
#pragma unroll
        for (int i = 0; i < 8; ++i) G->d[i] = d[i];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            G->f[i] = fv[2 * i] | (fv[2 * i + 1] << 16);
            G->b[i] = bv[2 * i] | (bv[2 * i + 1] << 16);
        }
    }
}

template <int BUF, class LineT>
__device__ __forceinline__ unsigned cost(const LineT& c, const LineT& n, int j)
{
    const int i = j + 3;
    if constexpr (BUF == 0) return absdiff(c.P[i - 3], n.P[i + 3]);
    if constexpr (BUF == 1) return absdiff(c.P[i - 2], n.P[i + 2]);
    if constexpr (BUF == 2) return absdiff(c.P[i - 1], n.P[i + 1]);
    if constexpr (BUF == 3) return absdiff(c.F(j), n.B(j));
    if constexpr (BUF == 4) return absdiff(c.P[i], n.P[i]);
    if constexpr (BUF == 5) return absdiff(c.B(j), n.F(j));
    if constexpr (BUF == 6) return absdiff(c.P[i + 1], n.P[i - 1]);
    if constexpr (BUF == 7) return absdiff(c.P[i + 2], n.P[i - 2]);
    return absdiff(c.P[i + 3], n.P[i - 3]);
}

// acc + cost<BUF>: v_sad_u16 adds its third operand, so S = A + D and A' = O + D are ONE instruction each and D itself
// never exists (with two waves per SIMD an instruction costs the same whatever it does: fewer is faster)
template <int BUF, class LineT>
__device__ __forceinline__ unsigned cost_acc(const LineT& c, const LineT& n, int j, unsigned acc)
{
    const int i = j + 3;
    if constexpr (BUF == 0) return __builtin_amdgcn_sad_u16(c.P[i - 3], n.P[i + 3], acc);
    if constexpr (BUF == 1) return __builtin_amdgcn_sad_u16(c.P[i - 2], n.P[i + 2], acc);
    if constexpr (BUF == 2) return __builtin_amdgcn_sad_u16(c.P[i - 1], n.P[i + 1], acc);
    if constexpr (BUF == 3) return __builtin_amdgcn_sad_u16(c.F(j), n.B(j), acc);
    if constexpr (BUF == 4) return __builtin_amdgcn_sad_u16(c.P[i], n.P[i], acc);
    if constexpr (BUF == 5) return __builtin_amdgcn_sad_u16(c.B(j), n.F(j), acc);
    if constexpr (BUF == 6) return __builtin_amdgcn_sad_u16(c.P[i + 1], n.P[i - 1], acc);
    if constexpr (BUF == 7) return __builtin_amdgcn_sad_u16(c.P[i + 2], n.P[i - 2], acc);
    return __builtin_amdgcn_sad_u16(c.P[i + 3], n.P[i - 3], acc);
}

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

#ifndef SN_U16_GATHER_ALL
#define SN_U16_GATHER_ALL 1
#endif
#ifndef SN_U16_POOL_RB
#define SN_U16_POOL_RB 9
#endif
#ifndef SN_U16_POOL_WIDE
#define SN_U16_POOL_WIDE 0
#endif
__host__ __device__ constexpr bool gather_stage3(int mode) { return SN_U16_GATHER_ALL || !has_pools(mode); }
__host__ __device__ constexpr bool wide_lines(int mode) { return !has_pools(mode) || SN_U16_POOL_WIDE; }
template <int MODE>
using LineOf = typename std::conditional<wide_lines(MODE), WideLine, Line>::type;

template <int BUF, int MODE>
constexpr unsigned rank_of()  // P4, P5, P3, P6, P2, P7, P1, P8, P0 -> 1..9 (SangNom2.cpp:214-249), or the gather's codes
{
    constexpr unsigned r[9] = {9, 7, 5, 3, 1, 2, 4, 6, 8};
    constexpr unsigned code[9] = {12, 6, 4, 2, 0, 1, 3, 5, 7};
    return gather_stage3(MODE) ? code[BUF] : r[BUF];
}

// The box of a plane on its own: two variants, the edge waves branch (measured: the uniform box below is 0.7 % slower
// there -- an 8-wave workgroup has one edge wave and one inner wave on the SIMDs concerned already)
template <bool EDGE>
__device__ __forceinline__ void box7_plain(const unsigned (&S)[PXL], unsigned (&Bx)[PXL], const LaneRole& role)
{
    unsigned L[3], R[3];
    if constexpr (EDGE) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            // bitwise selects, not ?: -- a DPP read executed under a lane mask would see the masked-off
            // source lanes as invalid (zero)
            L[k] = bfi(role.first_mask, S[0], dpp_from_left(S[PXL - 3 + k]));  // clamp to column 0
            R[k] = bfi(role.last_mask, S[PXL - 1], dpp_from_right(S[k]));      // clamp to column w-1
        }
    }
    auto X = [&](int i) -> unsigned {
        if (i < 0) return EDGE ? L[i + 3] : dpp_from_left(S[PXL + i]);
        if (i >= PXL) return EDGE ? R[i - PXL] : dpp_from_right(S[i - PXL]);
        return S[i];
    };
    Bx[0] = S[0] + S[1] + S[2] + S[3] + X(-1) + X(-2) + X(-3);
#pragma unroll
    for (int j = 0; j + 1 < PXL; ++j) Bx[j + 1] = Bx[j] - X(j - 3) + X(j + 4);
}

// One box for every wave (see sn_fused_u8_v3.hip, box7): column 0 is lane 0 of the first strip, whose DPP move keeps its
// `old` operand S[0] (lane 0 of the other strips is the outermost ghost lane); the last column takes a select.  No branch
// on "this wave holds an image edge" inside the buffer steps.
__device__ __forceinline__ void box7(const unsigned (&S)[PXL], unsigned (&Bx)[PXL], const LaneRole& role)
{
    unsigned L[3], R[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        L[k] = dpp_from_left_or(S[0], S[PXL - 3 + k]);                     // clamp to column 0
        R[k] = bfi(role.last_mask, S[PXL - 1], dpp_from_right(S[k]));      // clamp to column w-1
    }
    auto X = [&](int i) -> unsigned { return i < 0 ? L[i + 3] : i >= PXL ? R[i - PXL] : S[i]; };
    Bx[0] = S[0] + S[1] + S[2] + S[3] + X(-1) + X(-2) + X(-3);
#pragma unroll
    for (int j = 0; j + 1 < PXL; ++j) Bx[j + 1] = Bx[j] - X(j - 3) + X(j + 4);
}

// Scratch pools of the chroma coupling: [buffer][row][thread][4 dwords], dword k = O[2k] | O[2k+1] << 16.
struct PoolIO {
    __amdgpu_buffer_rsrc_t rin, rout;
    int v_a;    // slot this lane reads: its own, or its owner's for ghost lanes
    int v_out;  // slot this lane writes
    int row_stride, buf_stride;

    // va: v_a, or kOutOfRange where the row does not exist or the lane lies outside the dependency cone (Args::cone_*)
    __device__ __forceinline__ u32x4 issue(int b, int row, int va) const
    {
        return __builtin_amdgcn_raw_buffer_load_b128(rin, va, b * buf_stride + row * row_stride, 0);
    }
    __device__ __forceinline__ void finish(const u32x4& d, unsigned (&P)[PXL]) const
    {
        P[0] = d.x & kVal; P[1] = d.x >> 16;
        P[2] = d.y & kVal; P[3] = d.y >> 16;
        P[4] = d.z & kVal; P[5] = d.z >> 16;
        P[6] = d.w & kVal; P[7] = d.w >> 16;
    }
    __device__ __forceinline__ void store(int b, int row, int vout, const unsigned (&O)[PXL]) const
    {
        u32x4 d;
        d.x = O[0] | (O[1] << 16);
        d.y = O[2] | (O[3] << 16);
        d.z = O[4] | (O[5] << 16);
        d.w = O[6] | (O[7] << 16);
        store_b128(d, rout, vout, b * buf_stride + row * row_stride);
    }
};

// Chroma modes, waves of the region: the previous pass's row r + 1 of buffer B + kStaleAhead is fetched while buffer B is
// smoothed, and the last steps of a row fetch the first buffers of the next row's: the ring lives across rows (9 is a multiple
// of its size, so a buffer's slot is the same in every row).  A fetch -- also one whose every lane is out of range, as in the
// waves inside the region -- takes about 0.4 us under load, a step of a wave that has its SIMD to itself 0.3 us: one step
// ahead and a fresh start in every row (until round 4) made every step wait.
constexpr int kStaleAhead = 8;
static_assert(kBuffers % (kStaleAhead + 1) == 0, "a buffer's slot in the ring must not depend on the row");
struct StaleRing {
    u32x4 s[kStaleAhead + 1];
};

struct RowCtx {
    int r;
    int vin;  // kChroma: voffset of the loads of row r + 1 (out of range: row missing or outside the cone)
    int vin_next;  // ... and of row r + 2
    int vout;
    bool any_out;  // wave-uniform: some lane of this wave stores in this row
    int slot_c, slot_n;  // gather modes: LDS slots of the lines above / below the interpolated one
};

// STORE: the luma sweep's smoothed row goes to pool_out (rows past the hand-off run a loop without the packing: no
// branch inside a buffer step, see sn_fused_u8_v3.hip)
// FETCH (chroma modes): some lane of the wave takes the previous pass's value instead of a cost in this row.  A wave inside
// the region (three of the four of a 4:2:0 pass) does not: its step is the one of a plane on its own -- v_sad_u16 accumulates,
// D never exists -- and it fetches nothing (a fetch whose every lane is out of range is still a trip to the texture unit).
template <int BUF, int MODE, bool S1, bool STORE, bool FETCH>
__device__ __forceinline__ void buffer_step(unsigned (&A)[PXL], unsigned (&kmin)[PXL], const LineOf<MODE>& n, const LineOf<MODE>& nn,
                                            const LaneRole& role, const PoolIO& io, const RowCtx& rc, const u32x4& stale)
{
    unsigned D[PXL], S[PXL], Bx[PXL], O[PXL];
    if constexpr (MODE == kPlain || MODE == kLumaSpill) {
#pragma unroll
        for (int j = 0; j < PXL; ++j) S[j] = S1 ? cost_acc<BUF>(n, nn, j, A[j]) : A[j];
        if constexpr (MODE == kPlain) {
            if (role.edge_wave) box7_plain<true>(S, Bx, role);
            else box7_plain<false>(S, Bx, role);
        } else {
            box7(S, Bx, role);
        }
#pragma unroll
        for (int j = 0; j < PXL; ++j) {
            const unsigned key = and_or(Bx[j], role.key_mask, rank_of<BUF, MODE>());  // (sum / 16 mod 65536) << 4 | rank
            O[j] = key >> 4;                                                     // SangNom2.cpp:152
            A[j] = S1 ? cost_acc<BUF>(n, nn, j, O[j]) : O[j];
            kmin[j] = umin(kmin[j], key);
        }
        if constexpr (has_pools(MODE) && STORE) io.store(BUF, rc.r, rc.vout, O);  // lanes that keep nothing: out-of-range voffset
        return;
    }
    if constexpr (chroma_mode(MODE) && !FETCH) {
#pragma unroll
        for (int j = 0; j < PXL; ++j) S[j] = S1 ? cost_acc<BUF>(n, nn, j, A[j]) : A[j];
        box7(S, Bx, role);
#pragma unroll
        for (int j = 0; j < PXL; ++j) {
            const unsigned t = Bx[j] & 0xffff0u;
            O[j] = t >> 4;
            A[j] = S1 ? cost_acc<BUF>(n, nn, j, O[j]) : O[j];
            kmin[j] = umin(kmin[j], t | rank_of<BUF, MODE>());
        }
        if constexpr (MODE == kChroma) {
            if (rc.any_out) io.store(BUF, rc.r, rc.vout, O);
        }
        __builtin_amdgcn_sched_barrier(0);  // one step at a time: interleaved, the steps of a row spill
        return;
    }
    if constexpr (chroma_mode(MODE)) {
        io.finish(stale, D);
        if constexpr (S1) {
#pragma unroll
            for (int j = 0; j < PXL; ++j) D[j] = role.inside ? cost<BUF>(n, nn, j) : D[j];
        }
    } else if constexpr (MODE == kPadded) {
        const unsigned inside = role.inside ? 0xffffffffu : 0u;
#pragma unroll
        for (int j = 0; j < PXL; ++j) D[j] = S1 ? (cost<BUF>(n, nn, j) & inside) : 0u;
    } else {
#pragma unroll
        for (int j = 0; j < PXL; ++j) D[j] = S1 ? cost<BUF>(n, nn, j) : 0u;
    }
#pragma unroll
    for (int j = 0; j < PXL; ++j) S[j] = A[j] + D[j];
    box7(S, Bx, role);
#pragma unroll
    for (int j = 0; j < PXL; ++j) {
        const unsigned t = Bx[j] & 0xffff0u;  // O << 4; shared by O and the key: three full-rate ops
        O[j] = t >> 4;                        // (sum / 16) wraps to uint16_t, SangNom2.cpp:152
        A[j] = O[j] + D[j];
        kmin[j] = umin(kmin[j], t | rank_of<BUF, MODE>());
    }
    if constexpr (MODE == kChroma) io.store(BUF, rc.r, rc.vout, O);  // (kChromaLast hands nothing on)
}

// A wave of a chroma sweep whose columns all lie outside the chroma region has no lines, no stage 1 and no stage 3: the
// cost of the next row is what the previous pass left there, nothing else (see stale_wave_sweep).
template <int BUF>
__device__ __forceinline__ void stale_buffer_step(unsigned (&A)[PXL], const u32x4& stale, const LaneRole& role, const PoolIO& io, int r, int vout,
                                                  bool vout_any)
{
    unsigned D[PXL], S[PXL], Bx[PXL], O[PXL];
    io.finish(stale, D);
#pragma unroll
    for (int j = 0; j < PXL; ++j) S[j] = A[j] + D[j];
    box7(S, Bx, role);
#pragma unroll
    for (int j = 0; j < PXL; ++j) {
        O[j] = (Bx[j] >> 4) & kVal;  // (sum / 16) wraps to uint16_t, SangNom2.cpp:152
        A[j] = O[j] + D[j];
    }
    if (vout_any) io.store(BUF, r, vout, O);
}

__host__ __device__ constexpr int reg_buffers(int mode) { return has_pools(mode) ? (gather_stage3(mode) ? SN_U16_POOL_RB : 4) : 9; }

template <int NT, int RB>
struct Parked {
    uint4* v;  // [6][NT]: the parked line; gather modes: [3 lines][NT][kGatherDwords] dwords (see GatherLine)
    __device__ __forceinline__ void park_gather(int tid, int slot, const GatherLine& G) const
    {
        unsigned* to = reinterpret_cast<unsigned*>(v) + (slot * NT + tid) * kGatherDwords;
#pragma unroll
        for (int i = 0; i < 8; ++i) to[i] = G.d[i];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            to[8 + i] = G.f[i];
            to[12 + i] = G.b[i];
        }
    }
    // byte address (in the workgroup's LDS window) of halfword 0 of this thread's record of line `slot`
    __device__ __forceinline__ const unsigned short* gather_base(int tid, int slot) const
    {
        return reinterpret_cast<const unsigned short*>(reinterpret_cast<const unsigned*>(v) + (slot * NT + tid) * kGatherDwords);
    }
    __device__ __forceinline__ void park(int, const WideLine&) const {}  // never called: gather modes park a GatherLine
    uint4* a;  // [9 - RB][2][NT]: A of the LDS-resident buffers
    __device__ __forceinline__ void load_A(int tid, int b, unsigned (&A)[PXL]) const
    {
        const uint4 x = a[((b - RB) * 2 + 0) * NT + tid], y = a[((b - RB) * 2 + 1) * NT + tid];
        A[0] = x.x; A[1] = x.y; A[2] = x.z; A[3] = x.w;
        A[4] = y.x; A[5] = y.y; A[6] = y.z; A[7] = y.w;
    }
    __device__ __forceinline__ void store_A(int tid, int b, const unsigned (&A)[PXL]) const
    {
        a[((b - RB) * 2 + 0) * NT + tid] = make_uint4(A[0], A[1], A[2], A[3]);
        a[((b - RB) * 2 + 1) * NT + tid] = make_uint4(A[4], A[5], A[6], A[7]);
    }
    __device__ __forceinline__ void park(int tid, const Line& L) const
    {
        v[0 * NT + tid] = make_uint4(L.P[0], L.P[1], L.P[2], L.P[3]);
        v[1 * NT + tid] = make_uint4(L.P[4], L.P[5], L.P[6], L.P[7]);
        v[2 * NT + tid] = make_uint4(L.P[8], L.P[9], L.P[10], L.P[11]);
        v[3 * NT + tid] = make_uint4(L.P[12], L.P[13], L.FB[0], L.FB[1]);
        v[4 * NT + tid] = make_uint4(L.FB[2], L.FB[3], L.FB[4], L.FB[5]);
        v[5 * NT + tid] = make_uint4(L.FB[6], L.FB[7], 0u, 0u);
    }
    __device__ __forceinline__ void unpark(int tid, Line& L) const
    {
        const uint4 a0 = v[0 * NT + tid], b = v[1 * NT + tid], c = v[2 * NT + tid], d = v[3 * NT + tid],
                    e = v[4 * NT + tid], f = v[5 * NT + tid];
        L.P[0] = a0.x; L.P[1] = a0.y; L.P[2] = a0.z; L.P[3] = a0.w;
        L.P[4] = b.x; L.P[5] = b.y; L.P[6] = b.z; L.P[7] = b.w;
        L.P[8] = c.x; L.P[9] = c.y; L.P[10] = c.z; L.P[11] = c.w;
        L.P[12] = d.x; L.P[13] = d.y; L.FB[0] = d.z; L.FB[1] = d.w;
        L.FB[2] = e.x; L.FB[3] = e.y; L.FB[4] = e.z; L.FB[5] = e.w;
        L.FB[6] = f.x; L.FB[7] = f.y;
    }
};

struct Out {
    u32x4 v;  // 8 interpolated 16-bit pixels
};

template <int MODE, bool S1, bool S3, bool STORE, bool FETCH, int NT>
__device__ __forceinline__ Out row_step(unsigned (&A)[reg_buffers(MODE)][PXL], const Parked<NT, reg_buffers(MODE)>& pk,
                                        int tid, const LineOf<MODE>& n, const LineOf<MODE>& nn, const LaneRole& role, unsigned thr_key,
                                        const PoolIO& io, const RowCtx& rc, StaleRing& st)
{
    unsigned kmin[PXL];
#pragma unroll
    for (int j = 0; j < PXL; ++j) kmin[j] = thr_key;  // the `minBuf > aaf` arm: cost aaf + 1, rank 0
    auto run = [&](auto buf) {
        constexpr int B = decltype(buf)::value;
        if constexpr (chroma_mode(MODE) && FETCH) {  // see StaleRing
            constexpr int T = B + kStaleAhead;
            if constexpr (T < kBuffers) st.s[T % (kStaleAhead + 1)] = io.issue(T, rc.r + 1, rc.vin);
            else st.s[T % (kStaleAhead + 1)] = io.issue(T - kBuffers, rc.r + 2, rc.vin_next);
        }
        if constexpr (B < reg_buffers(MODE)) {
            buffer_step<B, MODE, S1, STORE, FETCH>(A[B], kmin, n, nn, role, io, rc, st.s[B % (kStaleAhead + 1)]);
        } else {
            unsigned t[PXL];
            pk.load_A(tid, B, t);
            buffer_step<B, MODE, S1, STORE, FETCH>(t, kmin, n, nn, role, io, rc, st.s[B % (kStaleAhead + 1)]);
            pk.store_A(tid, B, t);
        }
    };
    run(std::integral_constant<int, 0>{});
    run(std::integral_constant<int, 1>{});
    run(std::integral_constant<int, 2>{});
    run(std::integral_constant<int, 3>{});
    run(std::integral_constant<int, 4>{});
    run(std::integral_constant<int, 5>{});
    run(std::integral_constant<int, 6>{});
    run(std::integral_constant<int, 7>{});
    run(std::integral_constant<int, 8>{});
    Out o{};
    if constexpr (!S3) return o;
    if constexpr (gather_stage3(MODE)) {
        const unsigned short* cbase = pk.gather_base(tid, rc.slot_c);
        const unsigned short* nbase = pk.gather_base(tid, rc.slot_n);
        unsigned v[PXL];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            // the four winners' codes as bytes, then their taps' halfword offsets relative to the pixel
            const unsigned k01 = __builtin_amdgcn_perm(kmin[4 * g + 1], kmin[4 * g + 0], 0x0c0c0400u);
            const unsigned codes = __builtin_amdgcn_perm(kmin[4 * g + 3], kmin[4 * g + 2], 0x04000c0cu) | k01;
            const unsigned c4 = codes & 0x0f0f0f0fu;
            const unsigned tc = __builtin_amdgcn_perm(kLutCHi, kLutCLo, c4) ^ 0x01010101u;
            const unsigned tn = __builtin_amdgcn_perm(kLutNHi, kLutNLo, c4) ^ 0x07070707u;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int j = 4 * g + i;
                const unsigned oc = (tc >> (8 * i)) & 0xffu, on = (tn >> (8 * i)) & 0xffu;
                const unsigned ctap = cbase[oc + j], ntap = nbase[on + j];
                v[j] = (ctap + ntap + 1u) >> 1;  // (a + b + 1) >> 1, SangNom2.cpp:48-52
            }
        }
        o.v.x = v[0] | (v[1] << 16);
        o.v.y = v[2] | (v[3] << 16);
        o.v.z = v[4] | (v[5] << 16);
        o.v.w = v[6] | (v[7] << 16);
        return o;
    } else {
    Line c;
    pk.unpark(tid, c);
    unsigned v[PXL];
#pragma unroll
    for (int j = 0; j < PXL; ++j) {
        const unsigned wk = kmin[j];
        const unsigned m0 = 0u - (wk & 1u);
        const unsigned m1 = 0u - ((wk >> 1) & 1u);
        const unsigned m2 = 0u - ((wk >> 2) & 1u);
        const unsigned m3 = 0u - ((wk >> 3) & 1u);
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
        v[j] = (r + 1u) >> 1;  // (a + b + 1) >> 1, at most 65535
    }
    o.v.x = v[0] | (v[1] << 16);
    o.v.y = v[2] | (v[3] << 16);
    o.v.z = v[4] | (v[5] << 16);
    o.v.w = v[6] | (v[7] << 16);
    return o;
    }
}

// LDS mailbox: [refresh parity][wave][side][slot][72 A registers]; ghost lane `slot` on `side` of wave W
// loads what the seam lanes of the neighbouring wave published.
template <int NW>
struct Mailbox {
    unsigned* h;
    __device__ __forceinline__ unsigned* at(int par, int wave, int side, int slot) const
    {
        return h + (((par * (NW + 1) + wave) * 2 + side) * GH + slot) * (kBuffers * PXL);
    }
};

__host__ __device__ constexpr int lds_bytes(int nw, int mode)
{
    return ((gather_stage3(mode) ? (3 * kGatherDwords * 4 + 15) / 16 : 6) + (kBuffers - reg_buffers(mode)) * 2) * 16 * nw * 64
           + 2 * (nw + 1) * 2 * GH * kBuffers * PXL * 4;
}

template <int NW, int MODE, bool BAND>
__global__ void __launch_bounds__(NW * group_of(NW) * 64, 2) k_fused_u16_v3(Args a)
{
    extern __shared__ __align__(16) unsigned char lds_raw[];
    constexpr int RB = reg_buffers(MODE);
    constexpr int NT = NW * 64;
    // group_of(NW) frames per workgroup, each on its own NW waves and its own slice of the LDS; a frame index
    // past the end repeats the last frame (same inputs, same outputs: harmless, and the barrier counts agree)
    const int sub = __builtin_amdgcn_readfirstlane((int)threadIdx.x) / (NW * 64);
    const int f_raw = (int)blockIdx.x * group_of(NW) + sub;
    const int f = f_raw < a.nframes ? f_raw : a.nframes - 1;
    const int tid = (int)threadIdx.x - sub * (NW * 64);
    Parked<NT, RB> parked;
    parked.v = reinterpret_cast<uint4*>(lds_raw + sub * lds_bytes(NW, MODE));
    parked.a = parked.v + (gather_stage3(MODE) ? (3 * kGatherDwords * 4 + 15) / 16 : 6) * NT;
    Mailbox<NW> mb;
    mb.h = reinterpret_cast<unsigned*>(parked.a + (kBuffers - RB) * 2 * NT);
    const int wave = tid >> 6;
    const int lane = tid & 63;
#ifdef SN_WAVE_TIMING
    const unsigned long long wt_start = __builtin_amdgcn_s_memtime();
    unsigned long long wt_barrier = 0;
#endif

    // lane -> column group: wave 0 owns lanes 0..61 (all 64 if it is the only wave), later waves own
    // lanes 2..61 (the last one up to 63); the other lanes are ghosts of the neighbouring wave
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
    const int line_w = has_region(MODE) ? a.region_w : a.w;
    const bool line_live = live && x0 < line_w;
    const bool line_real = real && x0 < line_w;
    LaneRole role;
    role.first = live && gl == 0;
    role.last = live && gl == a.nl - 1;
    role.line_last = line_live && x0 + PXL == line_w;
    role.inside = line_live;
    role.key_mask = 0xffff0u;
    role.first_mask = role.first ? 0xffffffffu : 0u;
    role.last_mask = role.last ? 0xffffffffu : 0u;
    role.edge_wave = __builtin_amdgcn_readfirstlane(__any((int)(role.first || role.last || role.line_last)) ? 1 : 0) != 0;

    // buffer descriptors: row in the scalar offset, column in a constant per-lane voffset; dead lanes carry
    // an out-of-range voffset (loads return zero, stores are dropped)
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint8_t*>(a.src + (int64_t)f * a.src_frame_stride), 0, a.src_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rd =
        __builtin_amdgcn_make_buffer_rsrc(a.dst + (int64_t)f * a.dst_frame_stride, 0, a.dst_bytes, 0x00020000);
    const int vload = line_live ? (x0 > 0 ? 2 * (x0 - 4) : 0) : kOutOfRange;
    const int vstore = line_real ? 2 * x0 : kOutOfRange;
    const int src_step = (a.dh ? 1 : 2) * a.src_pitch;
    const int src_line = (a.dh ? 0 : a.offset) * a.src_pitch;
    const int dst_step = 2 * a.dst_pitch;
    const int dst_line = a.offset * a.dst_pitch;

    auto load_raw = [&](int row_off) {
        Raw q;
        q.a = __builtin_amdgcn_raw_buffer_load_b128(rs, vload, row_off, 0);
        q.b = __builtin_amdgcn_raw_buffer_load_b128(rs, vload, row_off + 16, 0);
        return q;
    };
    auto keep = [&](int row_off, const Raw& q, bool on = true) {  // GetFrame's field copy: the lane's own 8 pixels
        u32x4 v;
        if (role.first) v = q.a;                       // loaded from column 0: own pixels come first
        else { v.x = q.a.z; v.y = q.a.w; v.z = q.b.x; v.w = q.b.y; }
        store_b128(v, rd, on ? vstore : kOutOfRange, row_off);
    };
    auto put = [&](int row_off, const Out& o) { store_b128(o.v, rd, vstore, row_off); };

    PoolIO io{};
    if constexpr (has_pools(MODE)) {
        const bool linear = MODE == kLumaSpill && a.pool_row_bytes > 0;  // the pool path's layout
        io.row_stride = linear ? a.pool_row_bytes : NT * 16;
        const int pool_bytes = kBuffers * a.pool_rows * io.row_stride;
        io.buf_stride = a.pool_rows * io.row_stride;
        if (chroma_mode(MODE))
            io.rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(a.pool_in ? a.pool_in + (int64_t)f * a.pool_frame_stride : nullptr), 0,
                                                       a.pool_in ? pool_bytes : 0, 0x00020000);
        io.rout = __builtin_amdgcn_make_buffer_rsrc(a.pool_out ? a.pool_out + (int64_t)f * a.pool_frame_stride : nullptr, 0,
                                                    a.pool_out ? pool_bytes : 0, 0x00020000);
        int ta = tid;  // ghost lanes read the slot of the thread that owns their columns
        if (lane < GH && wave > 0) ta = (wave - 1) * 64 + (64 - 2 * GH) + lane;
        if (lane >= 64 - GH && wave < NW - 1) ta = (wave + 1) * 64 + GH + (lane - (64 - GH));
        io.v_a = ta * 16;
        io.v_out = real ? (linear ? x0 * 2 : tid * 16) : kOutOfRange;
    }

    // seam exchange: lanes 60, 61 feed the next wave's left ghosts, lanes 2, 3 the previous wave's right ghosts
    const bool pub_right = lane >= 64 - 2 * GH && lane < 64 - GH && wave < NW - 1;
    const bool pub_left = lane >= GH && lane < 2 * GH && wave > 0;
    const bool recv = ghost && live;
    const int slot = lane < GH ? lane : lane >= 64 - GH ? lane - (64 - GH) : pub_right ? lane - (64 - 2 * GH) : lane - GH;

    // Does this lane's slot of pool row q matter (Args::cone_*)?  Its first column is 8 * lane + 480 * wave.
    auto in_cone = [&](int q, int extra) -> bool {
        const int lim = a.cone_w + 3 * (a.cone_nr - q + 2) + extra;
        const int cols = lim < a.w ? lim : a.w;
        const int xa = (lane << 3) + wave * (kInner * PXL);
        return xa < cols && (q > a.cone_nr || xa + PXL > a.cone_w);
    };

    const int nk = a.nk;
    const int nr = nk - 1;
    // BAND (planes on their own only, sn_fused_v3_common.h): own rows ra .. rb, swept from row r0 on
    static_assert(!(BAND && chroma_mode(MODE)), "the chroma sweeps of the coupling are not cut");
    int r0 = 1, ra = 1, rb = nr;
    if constexpr (BAND) {
        ra = 1 + (int)blockIdx.y * a.band_rows;
        rb = ra + a.band_rows - 1 < nr ? ra + a.band_rows - 1 : nr;
        r0 = ra - a.band_warm > 1 ? ra - a.band_warm : 1;
        if (a.band_reset && blockIdx.y == 0 && tid == 0) a.band_flags[f] = 0;
    }
    const bool top = ra == 1;
    const int sweep = chroma_mode(MODE) ? a.sweep_rows : rb;
    const unsigned thr_key = (unsigned)(a.thr + 1) << 4;

    if constexpr (chroma_mode(MODE)) {
        // Waves entirely to the right of the chroma region (columns 480 * wave and up; with 4:2:0 the right half of the
        // workgroup) only re-smooth what the previous pass left: no lines, no costs, no ladder.  All nine buffers' state
        // stays in registers, the stale rows are fetched a whole row ahead, and the wave LEAVES once every column from
        // its first one on lies outside the dependency cone (Args::cone_*; the cone only shrinks): its SIMD partner --
        // a wave of the region, one per SIMD by construction (wave i and i + 4) -- then issues alone.
        if (wave * (kInner * PXL) >= a.region_w) {
            const int x_wave = wave * (kInner * PXL);
            unsigned As[kBuffers][PXL];
            const int v1 = (a.rows_in >= 1 && in_cone(1, a.cone_in)) ? io.v_a : kOutOfRange;
            u32x4 sa[kBuffers], sb[kBuffers];
#pragma unroll
            for (int b = 0; b < kBuffers; ++b) io.finish(io.issue(b, 1, v1), As[b]);
            auto fetch = [&](int q, u32x4 (&st)[kBuffers]) {  // pool row q of all nine buffers
                const int v = (q <= a.rows_in && in_cone(q, a.cone_in)) ? io.v_a : kOutOfRange;
#pragma unroll
                for (int b = 0; b < kBuffers; ++b) st[b] = io.issue(b, q, v);
            };
            fetch(2, sa);
            auto stale_row = [&](int r, const u32x4 (&cur)[kBuffers], u32x4 (&nxt)[kBuffers]) -> bool {
#ifdef SN_X_BALANCED  // DIAGNOSTIC (wrong results): every stale wave lives 5/9 of the sweep, the load of a balanced hand-over
                if (r > sweep * 5 / 9) return false;
#else
                if (x_wave >= a.cone_w + 3 * (a.cone_nr - (r - 1) + 2) + a.cone_in) return false;  // outside for good
#endif
                fetch(r + 2, nxt);
                if (r > 1 && (r - 1) % K == 0) {
                    SN_SYNC();
                    if (recv) {
                        const unsigned* from = mb.at((r / K) & 1, wave, lane < GH ? 0 : 1, slot);
#pragma unroll
                        for (int b = 0; b < kBuffers; ++b)
#pragma unroll
                            for (int j = 0; j < PXL; ++j) As[b][j] = from[b * PXL + j];
                    }
                }
                const int vout = (r <= a.rows_out && in_cone(r, a.cone_out)) ? io.v_out : kOutOfRange;
                const bool vout_any = __builtin_amdgcn_readfirstlane(__any(vout != kOutOfRange) ? 1 : 0) != 0;
                stale_buffer_step<0>(As[0], cur[0], role, io, r, vout, vout_any);
                stale_buffer_step<1>(As[1], cur[1], role, io, r, vout, vout_any);
                stale_buffer_step<2>(As[2], cur[2], role, io, r, vout, vout_any);
                stale_buffer_step<3>(As[3], cur[3], role, io, r, vout, vout_any);
                stale_buffer_step<4>(As[4], cur[4], role, io, r, vout, vout_any);
                stale_buffer_step<5>(As[5], cur[5], role, io, r, vout, vout_any);
                stale_buffer_step<6>(As[6], cur[6], role, io, r, vout, vout_any);
                stale_buffer_step<7>(As[7], cur[7], role, io, r, vout, vout_any);
                stale_buffer_step<8>(As[8], cur[8], role, io, r, vout, vout_any);
                if (r < sweep && r % K == 0) {
                    if (pub_right || pub_left) {
                        unsigned* to = pub_right ? mb.at(((r + 1) / K) & 1, wave + 1, 0, slot) : mb.at(((r + 1) / K) & 1, wave - 1, 1, slot);
#pragma unroll
                        for (int b = 0; b < kBuffers; ++b)
#pragma unroll
                            for (int j = 0; j < PXL; ++j) to[b * PXL + j] = As[b][j];
                    }
                }
                return true;
            };
            for (int r = 1; r <= sweep; r += 2) {
                if (!stale_row(r, sa, sb)) { SN_WT_FLUSH(); return; }
                if (r + 1 <= sweep && !stale_row(r + 1, sb, sa)) { SN_WT_FLUSH(); return; }
            }
            SN_WT_FLUSH();
            return;
        }
    }

    // a band copies the kept lines ra .. rb (the top band line 0 as well)
    LineOf<MODE> L0, L1;
    Raw q0 = load_raw(src_line + (r0 - 1) * src_step);
    Raw q1 = nk > 1 ? load_raw(src_line + r0 * src_step) : q0;
    keep(dst_line, q0, top);
    if (a.offset == 1) keep(0, q0, top);  // the line that cannot be interpolated, SangNom2.cpp:386-391
    if (nk > 1) keep(dst_line + r0 * dst_step, q1, r0 == ra);
    if constexpr (gather_stage3(MODE)) {
        GatherLine G;
        unpack(L0, q0, role, &G);
        parked.park_gather(tid, (r0 - 1) % 3, G);  // K[r0 - 1]: the upper line of row r0
        unpack(L1, q1, role, &G);
        parked.park_gather(tid, r0 % 3, G);        // K[r0]: its lower line
    } else {
        unpack(L0, q0, role);
        unpack(L1, q1, role);
        parked.park(tid, L0);
    }

    const bool first_in = chroma_mode(MODE) && a.rows_in >= 1 && in_cone(1, a.cone_in);

    // A[1] = O[0] + P[1] = P[1]
    unsigned A[RB][PXL];
    auto init_A = [&](auto buf, unsigned (&Ab)[PXL]) {
        constexpr int B = decltype(buf)::value;
        if constexpr (chroma_mode(MODE)) {
            io.finish(io.issue(B, 1, first_in ? io.v_a : kOutOfRange), Ab);
            if (nr > 0) {
#pragma unroll
                for (int j = 0; j < PXL; ++j) Ab[j] = role.inside ? cost<B>(L0, L1, j) : Ab[j];
            }
        } else if constexpr (MODE == kPadded) {
#pragma unroll
            for (int j = 0; j < PXL; ++j) Ab[j] = (r0 <= nr && role.inside) ? cost<B>(L0, L1, j) : 0u;
        } else {
#pragma unroll
            for (int j = 0; j < PXL; ++j) Ab[j] = r0 <= nr ? cost<B>(L0, L1, j) : 0u;
        }
    };
    auto init_buf = [&](auto buf) {
        constexpr int B = decltype(buf)::value;
        if constexpr (B < RB) {
            init_A(buf, A[B]);
        } else {
            unsigned t[PXL];
            init_A(buf, t);
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

    int src_next = src_line + (r0 + 1) * src_step;
    int dst_keep = dst_line + (r0 + 1) * dst_step;
    int out_row = dst_line + a.dst_pitch + (r0 - 1) * dst_step;  // where the pending row goes
    Raw qn = r0 + 1 <= nr ? load_raw(src_next) : q1;
    src_next += src_step;

    Out pending{};
    StaleRing stale_ring{};
    TurnTaking turns;
    turns.init(a.turn_shift);
    // fetch_tag (chroma modes): false for the rows in which no lane of this wave takes a stale value (buffer_step)
    auto step = [&](int r, LineOf<MODE>& n, LineOf<MODE>& nn, auto s1_tag, auto s3_tag, auto store_tag, auto fetch_tag) __attribute__((always_inline)) {
        constexpr bool S1 = decltype(s1_tag)::value;
        constexpr bool S3 = decltype(s3_tag)::value;
        constexpr bool STORE = decltype(store_tag)::value;
        turns.update((r - 1) % K);
        Raw qnext = qn;
        if constexpr (S1) {
            if constexpr (gather_stage3(MODE)) {
                GatherLine G;
                unpack(nn, qn, role, &G);  // waits for the line prefetched one row ago
                parked.park_gather(tid, (r + 1) % 3, G);  // K[r + 1]: lower line of the next row, upper line of the one after
            } else {
                unpack(nn, qn, role);
            }
            keep(dst_keep, qn, !BAND || (r + 1 >= ra && r < rb));
            dst_keep += dst_step;
        }
        if (r > r0 && r <= nr) {
            if (!BAND || r > ra) put(out_row, pending);  // stored here, after the prefetch wait (vmcnt counts in order)
            out_row += dst_step;
        }
        if constexpr (S1) {
            if (r + 2 <= nr) qnext = load_raw(src_next);
            src_next += src_step;
        }
        const int par = (r / K) & 1;
        if (r > r0 && (r - 1) % K == 0) {
            SN_SYNC();
            if (recv) {
                const unsigned* from = mb.at(par, wave, lane < GH ? 0 : 1, slot);
#pragma unroll
                for (int b = 0; b < RB; ++b)
#pragma unroll
                    for (int j = 0; j < PXL; ++j) A[b][j] = from[b * PXL + j];
#pragma unroll
                for (int b = RB; b < kBuffers; ++b) {
                    unsigned t[PXL];
#pragma unroll
                    for (int j = 0; j < PXL; ++j) t[j] = from[b * PXL + j];
                    parked.store_A(tid, b, t);
                }
            }
        }
        RowCtx rc;
        rc.r = r;
        rc.vin = rc.vout = kOutOfRange;
        rc.vin_next = kOutOfRange;
        if constexpr (chroma_mode(MODE)) {
            rc.vin = (r + 1 <= a.rows_in && in_cone(r + 1, a.cone_in)) ? io.v_a : kOutOfRange;
            rc.vin_next = (r + 2 <= a.rows_in && in_cone(r + 2, a.cone_in)) ? io.v_a : kOutOfRange;
        }
        rc.any_out = false;
        rc.slot_c = (r - 1) % 3;
        rc.slot_n = r % 3;
        if constexpr (has_pools(MODE)) {
            rc.vout = (r <= a.rows_out && (!BAND || r >= ra) && in_cone(r, a.cone_out)) ? io.v_out : kOutOfRange;
            rc.any_out = __builtin_amdgcn_readfirstlane(__any(rc.vout != kOutOfRange) ? 1 : 0) != 0;
        }
        pending = row_step<MODE, S1, S3, STORE, chroma_mode(MODE) && decltype(fetch_tag)::value>(A, parked, tid, n, nn, role, thr_key, io, rc, stale_ring);
        if constexpr (S1 && !gather_stage3(MODE)) parked.park(tid, n);  // n is the next row's c
        if (r < sweep && r % K == 0) {
            const int wpar = ((r + 1) / K) & 1;
            if (pub_right || pub_left) {
                unsigned* to = pub_right ? mb.at(wpar, wave + 1, 0, slot) : mb.at(wpar, wave - 1, 1, slot);
#pragma unroll
                for (int b = 0; b < RB; ++b)
#pragma unroll
                    for (int j = 0; j < PXL; ++j) to[b * PXL + j] = A[b][j];
#pragma unroll
                for (int b = RB; b < kBuffers; ++b) {
                    unsigned t[PXL];
                    parked.load_A(tid, b, t);
#pragma unroll
                    for (int j = 0; j < PXL; ++j) to[b * PXL + j] = t[j];
                }
            }
        }
        qn = qnext;
    };
    using T = std::integral_constant<bool, true>;
    using F = std::integral_constant<bool, false>;

    if constexpr (BAND) {
        // the state the band holds on entering a row: real columns only
        auto leave_state = [&](int which) {
            uint32_t* to = a.band_state + ((int64_t)(f * a.nbands + (int)blockIdx.y) * 2 + which) * (kBuffers * PXL * NT) + tid;
#pragma unroll
            for (int b = 0; b < RB; ++b)
#pragma unroll
                for (int j = 0; j < PXL; ++j) to[(b * PXL + j) * NT] = real ? A[b][j] : 0u;
#pragma unroll
            for (int b = RB; b < kBuffers; ++b) {
                unsigned t[PXL];
                parked.load_A(tid, b, t);
#pragma unroll
                for (int j = 0; j < PXL; ++j) to[(b * PXL + j) * NT] = real ? t[j] : 0u;
            }
        };
        using ST = std::integral_constant<bool, MODE == kLumaSpill>;
        int r = r0;
        for (; r < ra; ++r) {  // the run-up: nothing interpolated, nothing handed on
            step(r, L1, L0, T{}, F{}, F{}, T{});
            L1 = L0;
        }
        leave_state(0);
        const int own_next = rb < nr ? rb + 1 : nr;
        for (; r < own_next; ++r) {
            step(r, L1, L0, T{}, T{}, ST{}, T{});
            L1 = L0;
        }
        if (rb == nr) step(nr, L1, L0, F{}, T{}, ST{}, T{});
        else leave_state(1);
        put(out_row, pending);
    } else {
        // rows [from, to) with a following line pair: two rows per trip with the roles of the two line registers swapped
        // (no copy of a line per row); afterwards L1 is K[to] again
        auto rows = [&](int from, int to, auto store_tag, auto fetch_tag) __attribute__((always_inline)) {
            int r = from;
            for (; r + 1 < to; r += 2) {
                step(r, L1, L0, T{}, T{}, store_tag, fetch_tag);
                step(r + 1, L0, L1, T{}, T{}, store_tag, fetch_tag);
            }
            if (r < to) {
                step(r, L1, L0, T{}, T{}, store_tag, fetch_tag);
                L1 = L0;
            }
        };
        if constexpr (MODE == kLumaSpill) {
            // the rows a chroma pass can see first, with the hand-off; the rest of the plane without
            const int split = a.rows_out + 1 < nr ? a.rows_out + 1 : nr;
            // a wave left of the cone hands nothing on in the rows of the chroma region (only in the row or two below it, which
            // the next pass reads whole): it runs those rows without the packing and the stores, whose issue slots its SIMD
            // partner -- a wave of the right half, the ones the workgroup waits for -- can use
            const bool hands_on = __builtin_amdgcn_readfirstlane(__any(real && in_cone(1, a.cone_out)) ? 1 : 0) != 0;
            const int idle = hands_on ? 1 : (a.cone_nr + 1 < split ? a.cone_nr + 1 : split);
            rows(1, idle, F{}, T{});
            rows(idle, split, T{}, T{});
            rows(split, nr, F{}, T{});
            if (nr >= 1) {
                step(nr, L1, L0, F{}, T{}, T{}, T{});
                put(out_row, pending);
            }
        } else {
            int quiet = 1;  // rows [1, quiet): no lane of this wave takes a stale value (row r fetches pool row r + 1)
            if constexpr (chroma_mode(MODE)) {
                const int xa = (lane << 3) + wave * (kInner * PXL);
                const int lim = a.cone_w + 3 * (a.cone_nr - 2 + 2) + a.cone_in;  // in_cone(2, cone_in), the widest row the loop fetches
                const bool beyond = xa < (lim < a.w ? lim : a.w) && xa + PXL > a.cone_w;
                if (!__builtin_amdgcn_readfirstlane(__any(beyond) ? 1 : 0)) quiet = nr < a.cone_nr ? nr : a.cone_nr;
                if (quiet < 1) quiet = 1;
            }
            rows(1, quiet, F{}, F{});
            if constexpr (chroma_mode(MODE)) {  // the ring's first fill: the first buffers of the first row that fetches
                const int v = (quiet + 1 <= a.rows_in && in_cone(quiet + 1, a.cone_in)) ? io.v_a : kOutOfRange;
#pragma unroll
                for (int b = 0; b < kStaleAhead; ++b) stale_ring.s[b] = io.issue(b, quiet + 1, v);
            }
            rows(quiet, nr, F{}, T{});
            if (nr >= 1) {
                step(nr, L1, L0, F{}, T{}, F{}, T{});
                put(out_row, pending);
            }
            if constexpr (chroma_mode(MODE)) {
                for (int r = nr + 1; r <= sweep; ++r) step(r, L1, L0, F{}, F{}, F{}, T{});
            }
        }
    }

    // dst row h-1 := K[nk-1] when the top field is kept, SangNom2.cpp:380-385
    if (a.offset == 0 && rb == nr) {
        const Raw q = load_raw(src_line + (nk - 1) * src_step);
        keep((2 * nk - 1) * a.dst_pitch, q);
    }
    SN_WT_FLUSH();
}

#ifdef SN_WAVE_TIMING
}  // namespace w16
}  // namespace sn
extern "C" __attribute__((visibility("default"))) int sn_debug_wave_ticks(unsigned long long out[5 * 8 * 2], int reset)
{
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(sn::w16::sn_wave_ticks), 5 * 8 * 2 * sizeof(unsigned long long)) != hipSuccess) return 1;
    if (reset) {
        static const unsigned long long zero[5 * 8 * 2] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(sn::w16::sn_wave_ticks), zero, sizeof zero) != hipSuccess) return 1;
    }
    return 0;
}
namespace sn {
namespace w16 {
#endif

template <int MODE, bool BAND = false>
static hipError_t launch_mode(hipStream_t st, const Args& a, int nframes)
{
    const int g = v3c::group_of(a.nw);
    const int lds = lds_bytes(a.nw, MODE) * g;
    hipError_t e = hipSuccess;
#define SN_LAUNCH(NW)                                                                                              \
    case NW:                                                                                                       \
        if (lds > 64 * 1024)                                                                                       \
            e = hipFuncSetAttribute((const void*)k_fused_u16_v3<NW, MODE, BAND>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
        if (e == hipSuccess)                                                                                       \
            hipLaunchKernelGGL((k_fused_u16_v3<NW, MODE, BAND>), dim3((nframes + g - 1) / g, BAND ? a.nbands : 1), dim3(NW * g * 64), lds, st, a); \
        break;
    switch (a.nw) {
        SN_LAUNCH(1) SN_LAUNCH(2) SN_LAUNCH(3) SN_LAUNCH(4) SN_LAUNCH(5) SN_LAUNCH(6) SN_LAUNCH(7) SN_LAUNCH(8)
    default: return hipErrorInvalidValue;
    }
#undef SN_LAUNCH
    if (e != hipSuccess) return e;
    return hipGetLastError();
}

}  // namespace w16

bool fused_u16_plane_ok(int w)
{
    if (w % 32 != 0) return false;
    return v3c::strips_for(w / v3c::PXL) <= w16::kMaxWaves;
}

int fused_u16_waves(int sweep_w) { return v3c::strips_for(sweep_w / v3c::PXL); }

int64_t fused_u16_pool_bytes(int sweep_w, int rows) { return (int64_t)kBuffers * rows * fused_u16_waves(sweep_w) * 64 * 16; }

// Slot of thread t, dword k = O[2k] | O[2k+1] << 16 of the 8 columns the lane owns (PoolIO::store); ghost
// lanes and lanes past the sweep width own nothing.
void fused_u16_pool_unpack(const uint32_t* raw, int sweep_w, int rows, uint16_t* out)
{
    using namespace v3c;
    const int nl = sweep_w / PXL, nw = strips_for(nl), nt = nw * 64;
    for (int64_t br = 0; br < (int64_t)kBuffers * rows; ++br)
        for (int t = 0; t < nt; ++t) {
            const int wave = t / 64, lane = t % 64;
            const int gl = wave == 0 ? lane : kFirst + kInner * (wave - 1) + (lane - GH);
            const bool ghost = wave == 0 ? (nw > 1 && lane >= 64 - GH) : (lane < GH || (lane >= 64 - GH && wave < nw - 1));
            if (ghost || gl >= nl) continue;
            const uint32_t* d = raw + (br * nt + t) * 4;
            uint16_t* o = out + br * sweep_w + gl * PXL;
            for (int k = 0; k < 4; ++k) { o[2 * k] = (uint16_t)d[k]; o[2 * k + 1] = (uint16_t)(d[k] >> 16); }
        }
}

hipError_t launch_fused_u16_v3(hipStream_t st, const PlaneArgs& p, double threshold, int nframes, const FusedPool* pool)
{
    v3c::Args a{};
    a.src = p.src;
    a.dst = p.dst;
    a.src_frame_stride = p.src_frame_stride;
    a.dst_frame_stride = p.dst_frame_stride;
    a.src_pitch = p.src_pitch;
    a.dst_pitch = p.dst_pitch;
    a.w = pool && pool->mode != v3c::kPlain ? pool->sweep_w : p.w;
    a.nk = p.h_out / 2;
    a.offset = p.offset;
    a.dh = p.dh;
    a.thr = (int)threshold;
    a.nl = a.w / v3c::PXL;
    a.nvw = v3c::strips_for(a.nl);
    a.nw = a.nvw;
    a.turn_shift = v3c::turn_shift_for(a.nk, a.nw * v3c::group_of(a.nw), 2);
    a.nframes = nframes;
    a.src_bytes = (int)((int64_t)p.src_pitch * p.h_in);
    a.dst_bytes = (int)((int64_t)p.dst_pitch * p.h_out);
    if (!pool) return w16::launch_mode<v3c::kPlain>(st, a, nframes);
    if (pool->nbands > 1) {
        a.band_rows = pool->band_rows;
        a.band_warm = pool->band_warm;
        a.nbands = pool->nbands;
        a.band_state = pool->band_state;
        a.band_flags = pool->band_flags;
        a.band_reset = pool->band_reset;
        if (pool->mode == v3c::kPlain) return w16::launch_mode<v3c::kPlain, true>(st, a, nframes);
        if (pool->mode != v3c::kLumaSpill) return hipErrorInvalidValue;  // of the pool-coupled sweeps only the luma one is cut
    }
    if (pool->mode == v3c::kPlain) return w16::launch_mode<v3c::kPlain>(st, a, nframes);
    a.pool_in = pool->pool_in;
    a.pool_out = pool->pool_out;
    a.pool_frame_stride = pool->frame_stride;
    a.pool_rows = pool->pool_rows;
    a.rows_in = pool->rows_in;
    a.rows_out = pool->pool_out ? pool->rows_out : 0;
    a.region_w = p.w;
    a.sweep_rows = pool->sweep_rows;
    a.cone_w = pool->cone_w;
    a.cone_nr = pool->cone_nr;
    a.cone_in = pool->cone_in;
    a.cone_out = pool->cone_out;
    a.pool_row_bytes = pool->mode == v3c::kLumaSpill ? pool->pool_row_bytes : 0;
    if (pool->mode == v3c::kLumaSpill && a.nbands > 1) return w16::launch_mode<v3c::kLumaSpill, true>(st, a, nframes);
    if (pool->mode == v3c::kLumaSpill) return w16::launch_mode<v3c::kLumaSpill>(st, a, nframes);
    if (pool->mode == v3c::kPadded) return w16::launch_mode<v3c::kPadded>(st, a, nframes);
    if (!pool->pool_out) return w16::launch_mode<v3c::kChromaLast>(st, a, nframes);
    return w16::launch_mode<v3c::kChroma>(st, a, nframes);
}

}  // namespace sn
