// sn_fused_u8_v3.hip -- the fused sweep for 8-bit samples: frame assembly + stage 1 + stage 2 + stage 3 of one
// plane in ONE pass, with the nine cost buffers never leaving the CU.
//
// Reference semantics: /root/reference/src/SangNom2.cpp:74-124 (prepareBuffers_c), :126-159
// (processBuffers_c), :161-257 (finalizePlane_c), :361-391 (GetFrame's copies).
//
// The sweep (the 16-bit and float files share its shape):
//   * one workgroup per (frame, plane) sweeps the plane top to bottom, because stage 2 is a vertical recurrence
//     O[r] = box7(O[r-1] + D[r] + D[r+1]) / 16 (mod 256);
//   * a lane owns 8 consecutive pixels of every row IN EACH OF TWO COLUMN STRIPS: every quantity of stages 1 and 2 fits
//     13 bits, so a 32-bit register holds two pixels -- bits 0..15 strip `wave`, bits 16..31 strip `wave + NW` -- and one
//     instruction serves both.  Per cost buffer only A[r] = O[r-1] + D[r] is kept (8 registers), all nine in VGPRs;
//   * per buffer and register the row body is nine instructions: U = x -sat y, V = y -sat x (v_pk_sub_u16 clamp; the cost
//     |x - y| = U + V is never formed), S = A + U + V (v_add3), the sliding box B[j+1] = B[j] - X[j-3] + X[j+4] (two plain
//     adds), key = (B & 0x0ff00ff0) | code (v_and_or), O = key >> 4 as a PACKED shift (the code falls off both halves),
//     A' = O + U + V (v_add3), kmin = min(kmin, key) (v_pk_min_u16).  On gfx950 every packed, DPP and three-operand form
//     issues in the same 4-cycle class and one such form in eight drags a whole stream there (profiles/r3_ubench_valu_*),
//     so what counts is the NUMBER of instructions, and each of these forms replaces two to four simple ones;
//   * the +-3 horizontal taps of the box come from the neighbouring lanes with DPP wave_shr:1 / wave_shl:1 (one move
//     serves both strips).  Column 0 is lane 0 of the first strip and has no left neighbour: its DPP move keeps the `old`
//     operand, S[0] -- loadPixel's clamp (SangNom2.cpp:25-34) for free; the last column takes one select per right-hand tap.
//     The same box in every wave: no branch inside a buffer step;
//   * the first / last GH lanes of a strip are ghost lanes that recompute the neighbouring strip's 16 columns.  A ghost
//     zone stays exact in its innermost 3 pixels for floor(16 / 3) = 5 rows (the missing outer neighbour corrupts 3 more
//     pixels per row), so the waves meet only every K = 5 rows: seam lanes publish their A state to an LDS mailbox, one
//     s_barrier, the ghosts reload it.  Exact across seams, no speculation;
//   * stage 3's ladder (SangNom2.cpp:204-249) is the minimum over keys (O << 4) | code -- smallest cost wins, ties follow
//     the reference's priority order, the `minBuf > aaf` arm is a tenth key -- and what the winner selects is ONE byte
//     of the line above and ONE of the line below (or the two SangNom values): byte permutes (v_perm_b32) whose selectors
//     come from the winner's code through a byte table, then v_lerp_u8 = (a + b + 1) >> 1 on four pixels at once, on the
//     lines as they lie in memory (RawLine: windows cut once per line with v_alignbyte, kept in LDS for its two rows);
//   * memory: one buffer resource per frame plane, the row in the scalar offset, the column in a per-lane voffset that
//     never changes; dead and ghost lanes carry an out-of-range voffset (loads read zero, stores are dropped): no branch
//     around any memory operation.  One 16-byte load per strip and line, prefetched a row ahead.
// Modes (sn_fused_v3_common.h): planes on their own (kPlain, kPadded), the pool-coupled sweeps of subsampled chroma
// (kLumaSpill, kChroma, kChromaLast), and row bands (BAND) for launches of a few frames.  Everything is integer; results
// are bit-exact to the pool path and to the opt=0 reference.
#include <stdlib.h>

#include <type_traits>

#include "sn_fused_v3_common.h"
#include "sn_fused_u8_parts.h"  // lines, stage 1 operands, the box, stage 3 in the byte domain: shared with sn_fused_u8_uv.hip

namespace sn {
namespace v3 {

using namespace v3c;
#ifdef SN_ROW_TIMING  // tools/row_timing.py: where a wave of the plain sweep spends its shader-clock cycles, by phase of a row
__device__ unsigned long long sn_row_cycles[8];
#define SN_RT(k)                                                       \
    do {                                                               \
        __builtin_amdgcn_sched_barrier(0); /* no instruction of one phase is scheduled into another */ \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();  \
        rt_acc[k] += now_ - rt_last;                                   \
        rt_last = now_;                                                \
        __builtin_amdgcn_sched_barrier(0);                             \
    } while (0)
#else
#define SN_RT(k) do { } while (0)
#endif
[[maybe_unused]] constexpr int kMaxWaves = 8;         // physical waves per workgroup (16 virtual wavefronts, 7680 pixels)

// Access to the scratch pools of the chroma coupling (see Mode).  A thread has two 8-byte chunks in a pool row:
// chunk 0 = the 8 smoothed bytes of its low-half strip, chunk 1 = those of its high-half strip (all chunks 0 of the
// row first, then all chunks 1).  A lane fetches each
// half from the chunk of the thread that OWNS those columns (ghost lanes: the neighbouring wave's seam lanes; at the
// wrap seam between strip NW-1 and strip NW the owner's columns sit in the OTHER half), so nothing is shuffled after
// the load, and each half is stored / fetched only where the dependency cone (Args::cone_*) needs it.
struct PoolIO {
    __amdgpu_buffer_rsrc_t rin, rout;
    int v_lo, v_hi;          // voffsets of the chunks this lane's low / high half reads
    int v_out_lo, v_out_hi;  // ... writes (out of range for ghost and dead halves)
    int row_stride, buf_stride;

    struct RawPair {
        u32x2 lo, hi;
    };
    // issue the loads of buffer `b`, pool row `row`; vlo / vhi: v_lo / v_hi, or kOutOfRange where the row does not exist
    // or the half lies outside the cone (the load then returns zero)
    __device__ __forceinline__ RawPair issue(int b, int row, int vlo, int vhi) const
    {
        const int soff = b * buf_stride + row * row_stride;
        RawPair q;
        q.lo = __builtin_amdgcn_raw_buffer_load_b64(rin, vlo, soff, 0);
        q.hi = __builtin_amdgcn_raw_buffer_load_b64(rin, vhi, soff, 0);
        return q;
    }
    // ... and turn them into packed pairs
    __device__ __forceinline__ void finish(const RawPair& q, unsigned (&P)[PXL]) const
    {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            P[k] = pair_byte(q.hi.x, q.lo.x, k);
            P[4 + k] = pair_byte(q.hi.y, q.lo.y, k);
        }
    }
    __device__ __forceinline__ void load(int b, int row, int vlo, int vhi, unsigned (&P)[PXL]) const
    {
        finish(issue(b, row, vlo, vhi), P);
    }
    // vlo / vhi: v_out_lo / v_out_hi, or kOutOfRange for what is not kept (the store is then dropped: no branch)
    __device__ __forceinline__ void store(int b, int row, int vlo, int vhi, const unsigned (&O)[PXL]) const
    {
        // O[j] = low-strip byte | high-strip byte << 16  ->  four bytes per dword and strip
        const unsigned t01 = __builtin_amdgcn_perm(O[1], O[0], 0x06020400u), t23 = __builtin_amdgcn_perm(O[3], O[2], 0x06020400u);
        const unsigned t45 = __builtin_amdgcn_perm(O[5], O[4], 0x06020400u), t67 = __builtin_amdgcn_perm(O[7], O[6], 0x06020400u);
        u32x2 lo, hi;
        lo.x = __builtin_amdgcn_perm(t23, t01, 0x05040100u);
        lo.y = __builtin_amdgcn_perm(t67, t45, 0x05040100u);
        hi.x = __builtin_amdgcn_perm(t23, t01, 0x07060302u);
        hi.y = __builtin_amdgcn_perm(t67, t45, 0x07060302u);
        const int soff = b * buf_stride + row * row_stride;
        __builtin_amdgcn_raw_buffer_store_b64(lo, rout, vlo, soff, 0);
        __builtin_amdgcn_raw_buffer_store_b64(hi, rout, vhi, soff, 0);
    }
};

struct RowCtx {  // what a row needs besides the lines
    int r;          // pool row being smoothed
    int vin_lo, vin_hi;  // kChroma: voffsets of the loads of row r + 1 (out of range: row missing or outside the cone)
    int vout_hi;         // like vout (= the low half's chunk), for the high half's chunk
    bool any_out;        // wave-uniform: some lane of this wave stores in this row
    int vout;       // voffset for this row's O in pool_out (out of range: not kept)
    int slot_c, slot_n;  // byte-domain stage 3: LDS slots of the lines above / below the interpolated one
};

#ifndef SN_LUMA_WIDE
#define SN_LUMA_WIDE 0
#endif
#ifndef SN_CHROMA_WIDE
#define SN_CHROMA_WIDE 1
#endif
__host__ __device__ constexpr bool wide_lines(int mode) { return !has_pools(mode) || (mode == kLumaSpill && SN_LUMA_WIDE) || (chroma_mode(mode) && SN_CHROMA_WIDE); }
template <int MODE>
using LineOf = typename std::conditional<wide_lines(MODE), WideLine, Line>::type;

// S1: the costs of row r+1 come from the lines (n, nn); otherwise they are zero (kPlain /
// kLumaSpill: row bh is never written) or the previous pass's values (kChroma).
// STORE: the smoothed row goes to pool_out (lanes / rows that keep nothing carry an out-of-range voffset).
template <int BUF, int MODE, bool S1, bool STORE>
__device__ __forceinline__ void buffer_step(unsigned (&A)[PXL], unsigned (&kmin)[PXL], const LineOf<MODE>& n, const LineOf<MODE>& nn,
                                            const LaneRole& role, const PoolIO& io, const RowCtx& rc,
                                            PoolIO::RawPair& stale)
{
    unsigned D[PXL], S[PXL], Bx[PXL], O[PXL];
    if constexpr ((MODE == kPlain || MODE == kLumaSpill) && S1) {
        // A plane on its own never needs the cost itself: |x - y| = (x -sat y) + (y -sat x), so S = A + U + V and
        // A' = O + U + V are one three-operand add each -- two saturating subtracts and two v_add3 where maximum,
        // minimum, subtract and two adds took five instructions (every one costs the same issue slot here).
        unsigned U[PXL], V[PXL];
#pragma unroll
        for (int j = 0; j < PXL; ++j) {
            unsigned x, y;
            cost_operands<BUF>(n, nn, j, x, y);
            U[j] = pk_sub_sat(x, y);
            V[j] = pk_sub_sat(y, x);
            S[j] = add3(A[j], U[j], V[j]);
        }
        box7_any(S, Bx, role);
#pragma unroll
        for (int j = 0; j < PXL; ++j) {
            const unsigned key = and_or(Bx[j], role.key_mask, rank_of<BUF, MODE>());
            O[j] = pk_lshr4(key);
            A[j] = add3(O[j], U[j], V[j]);
            kmin[j] = pk_min(kmin[j], key);
        }
        if constexpr (MODE == kLumaSpill && STORE) io.store(BUF, rc.r, rc.vout, rc.vout_hi, O);
        return;
    }
    if constexpr (chroma_mode(MODE)) {
        io.finish(stale, D);
        // the stale row of the buffer after next is fetched into the registers this one has just left (two buffer steps
        // of lead, two loads in flight; the fence keeps the compiler from hoisting the load above the unpacking, which
        // would add registers in flight -- this mode lives at the register limit, and a spill reload waits for every
        // load before it)
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (BUF + 2 < kBuffers) stale = io.issue(BUF + 2, rc.r + 1, rc.vin_lo, rc.vin_hi);
        if constexpr (S1) {
            // No select: where the lines exist (inside the chroma region) the previous pass's row r + 1 is outside the
            // dependency cone as long as r + 1 <= cone_nr -- every row that has a line pair -- so its load was dropped and
            // reads zero; where they do not exist the lines read zero and so do their costs.  One of the two is zero
            // everywhere: S = A + cost + stale and A' = O + cost + stale are three-operand adds.
            unsigned C[PXL];
#pragma unroll
            for (int j = 0; j < PXL; ++j) {
                C[j] = cost<BUF>(n, nn, j);
                S[j] = add3(A[j], C[j], D[j]);
            }
            box7_any(S, Bx, role);
#pragma unroll
            for (int j = 0; j < PXL; ++j) {
                const unsigned key = and_or(Bx[j], role.key_mask, rank_of<BUF, MODE>());
                O[j] = pk_lshr4(key);
                A[j] = add3(O[j], C[j], D[j]);
                kmin[j] = pk_min(kmin[j], key);
            }
            if constexpr (MODE != kChromaLast && STORE) io.store(BUF, rc.r, rc.vout, rc.vout_hi, O);
            return;
        }
    } else if constexpr (MODE == kPadded) {
#pragma unroll
        for (int j = 0; j < PXL; ++j) D[j] = S1 ? (cost<BUF>(n, nn, j) & role.inside_mask) : 0u;
    } else {
#pragma unroll
        for (int j = 0; j < PXL; ++j) D[j] = S1 ? cost<BUF>(n, nn, j) : 0u;
    }
#pragma unroll
    for (int j = 0; j < PXL; ++j) S[j] = A[j] + D[j];
    box7_any(S, Bx, role);
#pragma unroll
    for (int j = 0; j < PXL; ++j) {
        // key = (sum / 16 mod 256) << 4 | rank in ONE v_and_or_b32; the rank (< 16) falls off the PACKED shift that
        // yields O (a 32-bit shift would push the high half's rank into the low half).  Every VALU instruction of this
        // stream costs the same issue slot (profiles/r3_ubench_valu_table.txt), so what counts is their NUMBER.
        const unsigned key = and_or(Bx[j], role.key_mask, rank_of<BUF, MODE>());
        O[j] = pk_lshr4(key);        // (sum / 16) wraps to uint8_t, SangNom2.cpp:152
        A[j] = O[j] + D[j];          // O + D[r+1]
        kmin[j] = pk_min(kmin[j], key);
    }
    // no branch around the packing (a branch inside the buffer steps costs more than it saves, see box7): rows past the
    // hand-off run a sweep without it (STORE), lanes that keep nothing let the range check drop their stores
    if constexpr (has_pools(MODE) && MODE != kChromaLast && STORE) io.store(BUF, rc.r, rc.vout, rc.vout_hi, O);
}

// Stage 3 needs the line above and the line below the interpolated one as RawLines; a line is cut into that form once,
// when it is unpacked, and parked in LDS for the two rows that use it (a ring of three lines, each lane reads back
// exactly what it wrote itself, so no barrier is involved).
// Buffers 0 .. reg_buffers(MODE)-1 keep their A state in VGPRs, the others (none since round 3, every mode holds all nine;
// the SN_*_RB knobs remain for A/B builds) in LDS between their steps.
#ifndef SN_PLAIN_RB
#define SN_PLAIN_RB 9
#endif
#ifndef SN_LUMA_RB
#define SN_LUMA_RB 9
#endif
#ifndef SN_CHROMA_RB
#define SN_CHROMA_RB 9
#endif
#ifndef SN_CHROMALAST_RB
#define SN_CHROMALAST_RB 9
#endif
__host__ __device__ constexpr int reg_buffers(int mode)
{
    return mode == kChromaLast ? SN_CHROMALAST_RB : mode == kChroma ? SN_CHROMA_RB : mode == kLumaSpill ? SN_LUMA_RB : mode == kPlain ? SN_PLAIN_RB : 9;
}
template <int NT, int RB>
struct Parked {  // views into the workgroup's dynamic LDS, sized by its thread count NT
    static constexpr int nthreads = NT;
    static constexpr int kRegBuffers = RB;
    uint4* v;    // [6][NT]: the parked line; byte-domain stage 3: [3 lines][5][NT], line k in slot k % 3 (RawLine)
    uint4* a;    // [kBuffers - RB][2][NT]: A of the LDS-resident buffers, thread-private slots
};
__host__ __device__ constexpr int parked_line_slots(int) { return 15; }  // three lines (RawLine) x five uint4 per thread

template <int NT, int RB>
__device__ __forceinline__ void load_A(const Parked<NT, RB>& pk, int tid, int b, unsigned (&A)[PXL])
{
    const uint4 x = pk.a[((b - RB) * 2 + 0) * pk.nthreads + tid], y = pk.a[((b - RB) * 2 + 1) * pk.nthreads + tid];
    A[0] = x.x; A[1] = x.y; A[2] = x.z; A[3] = x.w;
    A[4] = y.x; A[5] = y.y; A[6] = y.z; A[7] = y.w;
}
template <int NT, int RB>
__device__ __forceinline__ void store_A(const Parked<NT, RB>& pk, int tid, int b, const unsigned (&A)[PXL])
{
    pk.a[((b - RB) * 2 + 0) * pk.nthreads + tid] = make_uint4(A[0], A[1], A[2], A[3]);
    pk.a[((b - RB) * 2 + 1) * pk.nthreads + tid] = make_uint4(A[4], A[5], A[6], A[7]);
}

template <int NT, int RB>
__device__ __forceinline__ void park_raw(const Parked<NT, RB>& pk, int tid, int slot, const RawLine& R)
{
    park_raw_at(pk.v + slot * 5 * pk.nthreads, pk.nthreads, tid, R);
}
template <int NT, int RB>
__device__ __forceinline__ void unpark_raw(const Parked<NT, RB>& pk, int tid, int slot, RawLine& R)
{
    unpark_raw_at(pk.v + slot * 5 * pk.nthreads, pk.nthreads, tid, R);
}

// S3: the row has an interpolated line (stage 3); kChroma sweeps one extra row without one.
template <int MODE, bool S1, bool S3, bool STORE, int NT>
__device__ __forceinline__ Out row_step(unsigned (&A)[reg_buffers(MODE)][PXL], const Parked<NT, reg_buffers(MODE)>& pk, int tid, const LineOf<MODE>& n,
                                        const LineOf<MODE>& nn, const LaneRole& role, unsigned thr_key, const PoolIO& io,
                                        const RowCtx& rc)
{
    unsigned kmin[PXL];
#pragma unroll
    for (int j = 0; j < PXL; ++j) kmin[j] = thr_key;  // the `minBuf > aaf` arm: cost aaf + 1, rank 0
    // kChroma: the previous pass's row r+1 is fetched one buffer ahead of its use (HBM latency), and the
    // scheduler is kept from hoisting all nine fetches (their registers would spill).
    PoolIO::RawPair st0{}, st1{};  // even / odd buffers' stale rows in flight
    if constexpr (chroma_mode(MODE)) {
        st0 = io.issue(0, rc.r + 1, rc.vin_lo, rc.vin_hi);
        st1 = io.issue(1, rc.r + 1, rc.vin_lo, rc.vin_hi);
    }
    auto run = [&](auto buf) {
        constexpr int B = decltype(buf)::value;
        PoolIO::RawPair& st = (B & 1) ? st1 : st0;
        if constexpr (B < reg_buffers(MODE)) {
            buffer_step<B, MODE, S1, STORE>(A[B], kmin, n, nn, role, io, rc, st);
        } else {
            unsigned t[PXL];
            load_A(pk, tid, B, t);
            buffer_step<B, MODE, S1, STORE>(t, kmin, n, nn, role, io, rc, st);
            store_A(pk, tid, B, t);
        }
    };
    run(std::integral_constant<int, 0>{});
    if constexpr (MODE == kPlain) __builtin_amdgcn_sched_barrier(0);
    run(std::integral_constant<int, 1>{});
    if constexpr (MODE == kPlain) __builtin_amdgcn_sched_barrier(0);
    run(std::integral_constant<int, 2>{});
    if constexpr (MODE == kPlain) __builtin_amdgcn_sched_barrier(0);
    run(std::integral_constant<int, 3>{});
    if constexpr (MODE == kPlain) __builtin_amdgcn_sched_barrier(0);
    run(std::integral_constant<int, 4>{});
    if constexpr (MODE == kPlain) __builtin_amdgcn_sched_barrier(0);
    run(std::integral_constant<int, 5>{});
    if constexpr (MODE == kPlain) __builtin_amdgcn_sched_barrier(0);
    run(std::integral_constant<int, 6>{});
    if constexpr (MODE == kPlain) __builtin_amdgcn_sched_barrier(0);
    run(std::integral_constant<int, 7>{});
    if constexpr (MODE == kPlain) __builtin_amdgcn_sched_barrier(0);
    run(std::integral_constant<int, 8>{});
    if constexpr (MODE == kPlain) __builtin_amdgcn_sched_barrier(0);

    Out o{};
    if constexpr (!S3) return o;
    // rank codes of the eight winners of each strip, four to a dword in pixel order (RawLine has the rest)
    RawLine c{}, nr{};
#ifndef SN_X_NO_S3_LDS
    unpark_raw(pk, tid, rc.slot_c, c);
    unpark_raw(pk, tid, rc.slot_n, nr);
#else
    c.W[0][0] = kmin[0]; nr.W[1][1] = kmin[1];  // (knock-out: wrong results, no LDS reads in stage 3)
#endif
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const unsigned t01 = __builtin_amdgcn_perm(kmin[4 * g + 1], kmin[4 * g + 0], 0x06020400u);  // [lo0 lo1 hi0 hi1]
        const unsigned t23 = __builtin_amdgcn_perm(kmin[4 * g + 3], kmin[4 * g + 2], 0x06020400u);
        const unsigned lo = __builtin_amdgcn_perm(t23, t01, 0x05040100u) & 0x0f0f0f0fu;
        const unsigned hi = __builtin_amdgcn_perm(t23, t01, 0x07060302u) & 0x0f0f0f0fu;
        o.lo[g] = interpolate4(c, nr, 0, g, lo);
        o.hi[g] = interpolate4(c, nr, 1, g, hi);
    }
    return o;
}

// LDS mailbox, receiver-ready: word [refresh parity][wave W][side][slot][i] is what ghost lane
// `slot` on that side of wave W loads into packed A register i, so receiving costs no shuffling.  Inside the
// plane the publisher stores its registers whole; at the wrap seam a half is stored on its own (16-bit store):
//   left ghosts  (lanes 0, 1)   of wave W: both halves <- wave W-1 lanes 60, 61 (same half), except that the hi
//                                          half of wave 0 (strip NW) <- wave NW-1 lanes 60, 61 lo half (strip NW-1)
//   right ghosts (lanes 62, 63) of wave W: both halves <- wave W+1 lanes 2, 3 (same half), except that the lo
//                                          half of wave NW-1 (strip NW-1) <- wave 0 lanes 2, 3 hi half (strip NW)
// The pool-coupled modes park six buffers' state in LDS and have room for ONE copy of the mailbox only: there a second
// barrier (before publishing) makes sure every wave has taken the previous refresh out of it.  The other modes keep two
// copies, alternating, and meet once per refresh.
__host__ __device__ constexpr int mailbox_copies(int) { return 2; }  // (round 2: one copy and a second barrier where six buffers' state lived in LDS)
template <int NW, int COPIES>
struct Mailbox {  // [copy][wave 0..NW-1][side][slot][72][2 halves] 16-bit entries in dynamic LDS
    unsigned short* h;
    __device__ __forceinline__ unsigned short* at(int par, int wave, int side, int slot) const
    {
        return h + (((((COPIES > 1 ? par : 0) * NW + wave) * 2 + side) * GH + slot) * (kBuffers * PXL)) * 2;
    }
};

__host__ __device__ constexpr int lds_bytes(int nw, int mode)
{
    return (parked_line_slots(mode) + (kBuffers - reg_buffers(mode)) * 2) * 16 * nw * 64 + mailbox_copies(mode) * nw * 2 * GH * kBuffers * PXL * 4;
}

template <int NW, int MODE, bool BAND>
__global__ void __launch_bounds__(NW * group_of(NW) * 64, 2) k_fused_u8_v3(Args a)
{
    extern __shared__ __align__(16) unsigned char lds_raw[];
    // group_of(NW) frames per workgroup, each on its own NW waves and its own slice of the LDS; a frame index
    // past the end repeats the last frame (same inputs, same outputs: harmless, and the barrier counts agree)
    const int sub = __builtin_amdgcn_readfirstlane((int)threadIdx.x) / (NW * 64);
    const int f_raw = (int)blockIdx.x * group_of(NW) + sub;
    const int f = f_raw < a.nframes ? f_raw : a.nframes - 1;
    const int tid = (int)threadIdx.x - sub * (NW * 64);
    constexpr int kRegBuffers = reg_buffers(MODE);
    Parked<NW * 64, kRegBuffers> parked;
    parked.v = reinterpret_cast<uint4*>(lds_raw + sub * lds_bytes(NW, MODE));
    parked.a = parked.v + parked_line_slots(MODE) * NW * 64;
    Mailbox<NW, mailbox_copies(MODE)> mb;
    mb.h = reinterpret_cast<unsigned short*>(parked.a + (kBuffers - kRegBuffers) * 2 * NW * 64);
    const int wave = tid >> 6;
    const int lane = tid & 63;
    const int nvw = a.nvw;

    // per-half lane roles: virtual wavefront vw = wave + half * NW (strips in column order; the two
    // strips of a wave are NW strips apart, so both halves' neighbours sit in the adjacent wave)
    int x0[2];
    bool live[2], real[2], ghost[2];
    LaneRole role;
    role.first_mask = 0;
    role.last_mask = 0;
    role.line_last_mask = 0;
    role.inside_mask = 0;
    role.key_mask = 0x0ff00ff0u;
    const int line_w = has_region(MODE) ? a.region_w : a.w;  // width of the source / destination plane
    bool line_live[2], line_real[2];
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
        x0[h] = gl * PXL;
        line_live[h] = live[h] && x0[h] < line_w;
        line_real[h] = real[h] && x0[h] < line_w;
        if (live[h] && gl == 0) role.first_mask |= h ? kHi : kLo;
        if (live[h] && gl == a.nl - 1) role.last_mask |= h ? kHi : kLo;
        if (line_live[h] && x0[h] + PXL == line_w) role.line_last_mask |= h ? kHi : kLo;
        if (line_live[h]) role.inside_mask |= h ? kHi : kLo;
    }
    role.edge_wave =
        __builtin_amdgcn_readfirstlane(__any((int)(role.first_mask | role.last_mask | role.line_last_mask)) ? 1 : 0) != 0;

    // Buffer descriptors of this frame's source / destination plane; rows are addressed through the
    // scalar offset, columns through per-lane voffsets that never change.
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint8_t*>(a.src + (int64_t)f * a.src_frame_stride), 0, a.src_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rd =
        __builtin_amdgcn_make_buffer_rsrc(a.dst + (int64_t)f * a.dst_frame_stride, 0, a.dst_bytes, 0x00020000);
    int vload[2], vstore[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        vload[h] = line_live[h] ? (x0[h] > 0 ? x0[h] - 4 : 0) : kOutOfRange;
        vstore[h] = line_real[h] ? x0[h] : kOutOfRange;
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
    auto own = [&](const Raw& q, int h) {  // the 8 bytes the lane owns (the column-0 lane loaded them first)
        const bool first = (role.first_mask & (h ? kHi : kLo)) != 0;
        u32x2 v;
        v.x = first ? q.h[h].l : q.h[h].m0;
        v.y = first ? q.h[h].m0 : q.h[h].m1;
        return v;
    };
    auto keep = [&](int row_off, const Raw& q, bool on = true) {  // GetFrame's field copy, SangNom2.cpp:365 / :376
        __builtin_amdgcn_raw_buffer_store_b64(own(q, 0), rd, on ? vstore[0] : kOutOfRange, row_off, 0);
        __builtin_amdgcn_raw_buffer_store_b64(own(q, 1), rd, on ? vstore[1] : kOutOfRange, row_off, 0);
    };
    auto put = [&](int row_off, const Out& o) {
        u32x2 lo, hi;
        lo.x = o.lo[0]; lo.y = o.lo[1];
        hi.x = o.hi[0]; hi.y = o.hi[1];
        __builtin_amdgcn_raw_buffer_store_b64(lo, rd, vstore[0], row_off, 0);
        __builtin_amdgcn_raw_buffer_store_b64(hi, rd, vstore[1], row_off, 0);
    };

    // scratch pools of the chroma coupling
    PoolIO io{};
    if constexpr (has_pools(MODE)) {
        const bool linear = MODE == kLumaSpill && a.pool_row_bytes > 0;  // the pool path's layout
        io.row_stride = linear ? a.pool_row_bytes : NW * 64 * 16;
        const int pool_bytes = kBuffers * a.pool_rows * io.row_stride;
        io.buf_stride = a.pool_rows * io.row_stride;
        if (chroma_mode(MODE))
            io.rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(a.pool_in ? a.pool_in + (int64_t)f * a.pool_frame_stride : nullptr), 0,
                                                       a.pool_in ? pool_bytes : 0, 0x00020000);
        io.rout = __builtin_amdgcn_make_buffer_rsrc(a.pool_out ? a.pool_out + (int64_t)f * a.pool_frame_stride : nullptr, 0,
                                                    a.pool_out ? pool_bytes : 0, 0x00020000);
        // whose chunks a lane reads: its own, or -- for ghost lanes -- those of the thread that owns the columns (lanes
        // 60, 61 of the previous wave / lanes 2, 3 of the next one).  The two strips of a wave meet their neighbours in
        // the same thread, except at the seam between strip NW-1 and strip NW, where the partner's columns sit in
        // the other half of another wave.
        int t_lo = tid, t_hi = tid, c_lo = 0, c_hi = 1;
        if (lane < GH) {
            if (wave > 0) t_lo = t_hi = (wave - 1) * 64 + (64 - 2 * GH) + lane;
            else { t_hi = (NW - 1) * 64 + (64 - 2 * GH) + lane; c_hi = 0; }  // hi half <- strip NW-1 (a low half)
        } else if (lane >= 64 - GH) {
            // only where these lanes ARE ghosts: in the last strip they own columns (lanes 62, 63 of a strip that ends
            // there -- 64 + 60 k lanes, e.g. 512 or 992 columns -- read their own chunks; treating them as ghosts made
            // the sweep see zeros in the last 16 columns of the previous pass's rows)
            if (ghost[0]) {
                if (wave < NW - 1) t_lo = (wave + 1) * 64 + GH + (lane - (64 - GH));
                else { t_lo = GH + (lane - (64 - GH)); c_lo = 1; }           // lo half <- strip NW (a high half)
            }
            if (ghost[1]) t_hi = (wave + 1) * 64 + GH + (lane - (64 - GH));  // strip wave + NW + 1: the next wave's high half
        }
        // a pool row is [chunk kind][thread][8 bytes]: the chunks of one kind lie side by side, so the lanes inside the
        // cone write and read whole cache lines (interleaved with the other kind every line was half useful)
        io.v_lo = c_lo * (NW * 64 * 8) + t_lo * 8;
        io.v_hi = c_hi * (NW * 64 * 8) + t_hi * 8;
        io.v_out_lo = real[0] ? (linear ? x0[0] : tid * 8) : kOutOfRange;
        io.v_out_hi = real[1] ? (linear ? x0[1] : NW * 64 * 8 + tid * 8) : kOutOfRange;
    }

    // Does a half of this lane matter in pool row q (Args::cone_*)?  Column of the low half: 8 * lane + 480 * wave, the
    // high half lies 480 * NW further right (strips in column order); recomputed per row from the lane id, not kept.
    auto in_cone = [&](int q, int extra, int h) -> bool {
        const int lim = a.cone_w + 3 * (a.cone_nr - q + 2) + extra;
        const int cols = lim < a.w ? lim : a.w;
        const int x = (lane << 3) + (wave + h * NW) * (kInner * PXL);
        return x < cols && (q > a.cone_nr || x + PXL > a.cone_w);
    };

    const int nk = a.nk;
    const int nr = nk - 1;
    const int last = chroma_mode(MODE) ? a.sweep_rows : nr;  // pool rows 1 .. last are swept
    // BAND: own rows ra .. rb, swept from row r0 on; otherwise the whole plane
    int r0 = 1, ra = 1, rb = last;
    if constexpr (BAND) {
        ra = 1 + (int)blockIdx.y * a.band_rows;
        rb = ra + a.band_rows - 1 < last ? ra + a.band_rows - 1 : last;
        r0 = ra - a.band_warm > 1 ? ra - a.band_warm : 1;
        if (a.band_reset && blockIdx.y == 0 && tid == 0) a.band_flags[f] = 0;
    }
    const bool top = ra == 1;                    // the band with the first row of the plane
    const bool bottom = rb >= nr;                // ... with its last interpolated line
    const int sweep = rb;
    const unsigned thr_key = (unsigned)((a.thr + 1) << 4) * 0x00010001u;

    // a band copies the kept lines ra .. rb (the top band line 0 as well)
    LineOf<MODE> L0, L1;
    Raw q0 = load_raw(src_line + (r0 - 1) * src_step);
    Raw q1 = nk > 1 ? load_raw(src_line + r0 * src_step) : q0;
    // The third line is fetched HERE, before the first stores, so that on entering the row loop the loads in flight have
    // stores behind them, as they do when the loop comes round (a row's order is: kept-line stores, next line's loads, ...,
    // output stores).  vmcnt counts in order: with the loads as the newest operations on ONE way into the loop header the
    // compiler waits there with vmcnt(1) / vmcnt(0) in every other row -- i.e. for the output stores of the row just
    // finished, a trip to memory of which a wave of an 8-wave workgroup hides nothing (4320p: wait_any 0.29) -- instead of
    // vmcnt(3) / vmcnt(2), which only asks for loads issued a whole row ago.
    Raw qn = r0 + 1 <= nr ? load_raw(src_line + (r0 + 1) * src_step) : q1;
    keep(dst_line, q0, top);
    if (a.offset == 1) keep(0, q0, top);  // the line that cannot be interpolated, SangNom2.cpp:386-391
    if (nk > 1) keep(dst_line + r0 * dst_step, q1, r0 == ra && r0 <= nr);
    {
        const Raw f0 = clamp_edges(q0, role), f1 = clamp_edges(q1, role);
        unpack(L0, f0);
        unpack(L1, f1);
        RawLine R;
        make_raw(R, f0, L0);
        park_raw(parked, tid, (r0 - 1) % 3, R);  // K[r0 - 1]: c of row r0
        make_raw(R, f1, L1);
        park_raw(parked, tid, r0 % 3, R);        // K[r0]: n of row r0
    }

    // A[1] = O[0] + P[1] = P[1] (pool row 0 is never written: zero); P[1] = stage-1 costs of the first
    // line pair, and outside the chroma region (kChroma) what the previous pass left in row 1
    unsigned A[kRegBuffers][PXL];
    const bool first_lo = chroma_mode(MODE) && a.rows_in >= r0 && in_cone(r0, a.cone_in, 0);
    const bool first_hi = chroma_mode(MODE) && a.rows_in >= r0 && in_cone(r0, a.cone_in, 1);
    auto init_A = [&](auto buf, unsigned (&Ab)[PXL]) {
        constexpr int B = decltype(buf)::value;
        if constexpr (chroma_mode(MODE)) {
            io.load(B, r0, first_lo ? io.v_lo : kOutOfRange, first_hi ? io.v_hi : kOutOfRange, Ab);
            if (r0 <= nr) {
#pragma unroll
                for (int j = 0; j < PXL; ++j) Ab[j] = bfi(role.inside_mask, cost<B>(L0, L1, j), Ab[j]);
            }
        } else if constexpr (MODE == kPadded) {
#pragma unroll
            for (int j = 0; j < PXL; ++j) Ab[j] = r0 <= nr ? (cost<B>(L0, L1, j) & role.inside_mask) : 0u;
        } else {
#pragma unroll
            for (int j = 0; j < PXL; ++j) Ab[j] = r0 <= nr ? cost<B>(L0, L1, j) : 0u;
        }
    };
    auto init_buf = [&](auto buf) {
        constexpr int B = decltype(buf)::value;
        if constexpr (B < kRegBuffers) {
            init_A(buf, A[B]);
        } else {
            unsigned t[PXL];
            init_A(buf, t);
            store_A(parked, tid, B, t);
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
    int out_row = dst_line + a.dst_pitch + (r0 - 1) * dst_step;
    src_next += src_step;

    // Seam exchange roles.  Lanes 60, 61 are the right seam lanes and lanes 2, 3 the left seam
    // lanes of BOTH virtual wavefronts of this wave, so they publish whole packed registers; the
    // ghost lanes (0, 1 and 62, 63) take one half from each of two published registers.
    const bool pub_right = lane >= 64 - 2 * GH && lane < 64 - GH;
    const bool pub_left = lane >= GH && lane < 2 * GH;
    const bool recv_left = lane < GH;         // left ghosts of both strips
    const bool recv_right = lane >= 64 - GH;  // right ghosts of both strips
    const unsigned ghost_mask = (ghost[0] && live[0] ? kLo : 0u) | (ghost[1] && live[1] ? kHi : 0u);
    const int slot = recv_left ? lane : recv_right ? lane - (64 - GH) : pub_right ? lane - (64 - 2 * GH) : lane - GH;

    // One pool row r: n = K[r], nn = K[r+1] (S1: the pair exists), c = K[r-1] parked (S3: the row has an
    // interpolated line).
    TurnTaking turns;
    turns.init(a.turn_shift);
#ifdef SN_ROW_TIMING
    unsigned long long rt_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, rt_last = __builtin_amdgcn_s_memtime();
#endif
    auto step = [&](int r, LineOf<MODE>& n, LineOf<MODE>& nn, auto s1_tag, auto s3_tag, auto store_tag) __attribute__((always_inline)) {
        constexpr bool HAS_NEXT = decltype(s1_tag)::value;
        constexpr bool S3 = decltype(s3_tag)::value;
        constexpr bool STORE = decltype(store_tag)::value;
        turns.update((r - 1) % K);
        SN_RT(6);
        Raw qnext = qn;
        if constexpr (HAS_NEXT) {
            const Raw fq = clamp_edges(qn, role);  // waits for the line prefetched one row ago
#ifdef SN_ROW_TIMING
            asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            SN_RT(0);  // [0] the wait for the prefetched line
#endif
            unpack(nn, fq);
            RawLine R;
            make_raw(R, fq, nn);
            park_raw(parked, tid, (r + 1) % 3, R);  // K[r + 1]: n of the next row, c of the one after
            keep(dst_keep, qn, !BAND || (r + 1 >= ra && r < rb));
            dst_keep += dst_step;
        }
        if constexpr (HAS_NEXT) {
            if (r + 2 <= nr) qnext = load_raw(src_next);  // prefetch K[r+2]
            src_next += src_step;
        }
        const int par = (r / K) & 1;
        SN_RT(1);  // [1] unpack, windows, park, kept-line store, prefetch issue
        if (r > r0 && (r - 1) % K == 0) {
#ifndef SN_X_NO_SEAM_BARRIER
            __syncthreads();
#endif
            SN_RT(2);  // [2] the seam barrier
            if (recv_left || recv_right) {
                const unsigned* from = reinterpret_cast<const unsigned*>(mb.at(par, wave, recv_left ? 0 : 1, slot));
                auto merge = [&](int b, unsigned (&Ab)[PXL]) {
#pragma unroll
                    for (int j = 0; j < PXL; ++j) Ab[j] = bfi(ghost_mask, from[b * PXL + j], Ab[j]);
                };
#pragma unroll
                for (int b = 0; b < kRegBuffers; ++b) merge(b, A[b]);
#pragma unroll
                for (int b = kRegBuffers; b < kBuffers; ++b) {
                    unsigned t[PXL];
                    load_A(parked, tid, b, t);
                    merge(b, t);
                    store_A(parked, tid, b, t);
                }
            }
        }
        RowCtx rc;
        rc.r = r;
        rc.vin_lo = rc.vin_hi = rc.vout = rc.vout_hi = kOutOfRange;
        rc.any_out = false;
        rc.slot_c = (r - 1) % 3;
        rc.slot_n = r % 3;
        if constexpr (chroma_mode(MODE)) {
            const bool row_in = r + 1 <= a.rows_in;
            rc.vin_lo = (row_in && in_cone(r + 1, a.cone_in, 0)) ? io.v_lo : kOutOfRange;
            rc.vin_hi = (row_in && in_cone(r + 1, a.cone_in, 1)) ? io.v_hi : kOutOfRange;
        }
        if constexpr (has_pools(MODE)) {
            const bool row_out = r <= a.rows_out && (!BAND || r >= ra);
            rc.vout = (row_out && in_cone(r, a.cone_out, 0)) ? io.v_out_lo : kOutOfRange;
            rc.vout_hi = (row_out && in_cone(r, a.cone_out, 1)) ? io.v_out_hi : kOutOfRange;
            rc.any_out = __builtin_amdgcn_readfirstlane(__any((rc.vout != kOutOfRange) | (rc.vout_hi != kOutOfRange)) ? 1 : 0) != 0;
        }
        SN_RT(3);  // [3] ghost refresh (every fifth row) and row set-up
        const Out o = row_step<MODE, HAS_NEXT, S3, STORE>(A, parked, tid, n, nn, role, thr_key, io, rc);
        if constexpr (S3) put(out_row, o);  // stored at once: nothing is carried into the next row
        SN_RT(4);  // [4] nine buffer steps + stage 3 + output store
        out_row += dst_step;
        if (r < sweep) {
            if (r % K == 0) {
                if constexpr (mailbox_copies(MODE) == 1) __syncthreads();  // the previous refresh has been taken out
                const int wpar = ((r + 1) / K) & 1;
                if (pub_right || pub_left) {
                    // Inside the plane both halves of a seam register go to the same ghost lane of the neighbouring
                    // wave (see Mailbox), i.e. the register is published whole, 16 bytes per store.  Only at the
                    // wrap seam (strip NW-1 | strip NW) a half crosses over into the other half of its receiver
                    // and is written on its own; halves at the image edge have no receiver at all.
                    const bool whole = pub_right ? wave < NW - 1 : wave > 0;
                    if (whole) {
                        uint4* to = reinterpret_cast<uint4*>(pub_right ? mb.at(wpar, wave + 1, 0, slot) : mb.at(wpar, wave - 1, 1, slot));
                        auto send = [&](int b, const unsigned (&Ab)[PXL]) {
                            to[b * 2 + 0] = make_uint4(Ab[0], Ab[1], Ab[2], Ab[3]);
                            to[b * 2 + 1] = make_uint4(Ab[4], Ab[5], Ab[6], Ab[7]);
                        };
#pragma unroll
                        for (int b = 0; b < kRegBuffers; ++b) send(b, A[b]);
#pragma unroll
                        for (int b = kRegBuffers; b < kBuffers; ++b) {
                            unsigned t[PXL];
                            load_A(parked, tid, b, t);
                            send(b, t);
                        }
                    } else {
                        // wave NW-1, right seam lanes: low half (strip NW-1) -> high half of wave 0's left ghosts;
                        // wave 0, left seam lanes: high half (strip NW) -> low half of wave NW-1's right ghosts
                        // (two branches with the half fixed at compile time: the high half goes out with ds_write_b16_d16_hi, no
                        // shift -- a per-lane shift amount cost 72 v_lshrrev per publish in the two waves the others wait for)
                        auto send_half = [&](unsigned short* to, auto hi_tag) {
                            constexpr bool HI = decltype(hi_tag)::value;
                            auto send = [&](int b, const unsigned (&Ab)[PXL]) {
#pragma unroll
                                for (int j = 0; j < PXL; ++j) to[(b * PXL + j) * 2] = HI ? (unsigned short)(Ab[j] >> 16) : (unsigned short)Ab[j];
                            };
#pragma unroll
                            for (int b = 0; b < kRegBuffers; ++b) send(b, A[b]);
#pragma unroll
                            for (int b = kRegBuffers; b < kBuffers; ++b) {
                                unsigned t[PXL];
                                load_A(parked, tid, b, t);
                                send(b, t);
                            }
                        };
                        if (pub_right) send_half(mb.at(wpar, 0, 0, slot) + 1, std::integral_constant<bool, false>{});
                        else send_half(mb.at(wpar, NW - 1, 1, slot), std::integral_constant<bool, true>{});
                    }
                }
            }
        }
        qn = qnext;
        SN_RT(5);  // [5] publish (every fifth row)
#ifdef SN_ROW_TIMING
        rt_acc[7] += 1;
#endif
    };
    using T = std::integral_constant<bool, true>;
    using F = std::integral_constant<bool, false>;

    // L1 = K[r] (n), L0 is reused for K[r+1] (nn); c lives in LDS.  Rows 1 .. nr-1 have a following line
    // pair, row nr does not (its next costs are zero or stale), rows beyond nr (kChroma only) have no
    // interpolated line either.  (Specialising the whole sweep on role.edge_wave -- two copies of the loop, no
    // branch per buffer step -- was tried: the register allocator then spills in both copies.)
    if constexpr (BAND) {
        // the state the band holds on entering a row, as the next row's sweep sees it: real columns only
        auto leave_state = [&](int which) {
            const unsigned real_mask = (real[0] ? kLo : 0u) | (real[1] ? kHi : 0u);
            uint32_t* to = a.band_state + ((int64_t)(f * a.nbands + (int)blockIdx.y) * 2 + which) * (kBuffers * PXL * NW * 64) + tid;
#pragma unroll
            for (int b = 0; b < kRegBuffers; ++b)
#pragma unroll
                for (int j = 0; j < PXL; ++j) to[(b * PXL + j) * (NW * 64)] = A[b][j] & real_mask;
#pragma unroll
            for (int b = kRegBuffers; b < kBuffers; ++b) {
                unsigned t[PXL];
                load_A(parked, tid, b, t);
#pragma unroll
                for (int j = 0; j < PXL; ++j) to[(b * PXL + j) * (NW * 64)] = t[j] & real_mask;
            }
        };
        // run-up rows (nothing interpolated, nothing handed on), then the band's own rows; row nr has no following line pair
        static_assert(!chroma_mode(MODE), "the chroma sweeps of the coupling are not cut (sn_fused_v3_common.h)");
        using ST = std::integral_constant<bool, MODE == kLumaSpill>;
        int r = r0;
        for (; r < ra; ++r) {
            step(r, L1, L0, T{}, F{}, F{});
            L1 = L0;
        }
        leave_state(0);
        const int own_next = rb < nr ? rb + 1 : nr;
        for (; r < own_next; ++r) {
            step(r, L1, L0, T{}, T{}, ST{});
            L1 = L0;
        }
        if (rb == nr) step(nr, L1, L0, F{}, T{}, ST{});
        if (rb < last) leave_state(1);
    } else {
        // Rows [from, to) with a following line pair: two rows per trip with the roles of the two line registers swapped,
        // so that no row ends in a copy of a line; afterwards L1 is K[to] again.
        auto rows = [&](int from, int to, auto store_tag) __attribute__((always_inline)) {
            int r = from;
            for (; r + 1 < to; r += 2) {
                step(r, L1, L0, T{}, T{}, store_tag);
                step(r + 1, L0, L1, T{}, T{}, store_tag);
            }
            if (r < to) {
                step(r, L1, L0, T{}, T{}, store_tag);
                L1 = L0;
            }
        };
        if constexpr (MODE == kLumaSpill) {
            // the rows whose smoothed values a chroma pass can see first, with the hand-off; the rest of the plane without
            const int split = a.rows_out + 1 < nr ? a.rows_out + 1 : nr;
            rows(1, split, T{});
            rows(split, nr, F{});
            if (nr >= 1) step(nr, L1, L0, F{}, T{}, T{});
        } else {
            using ST = std::integral_constant<bool, MODE == kChroma>;
            rows(1, nr, ST{});
            if (nr >= 1) step(nr, L1, L0, F{}, T{}, ST{});
            if constexpr (chroma_mode(MODE)) {
                for (int r = nr + 1; r <= last; ++r) step(r, L1, L0, F{}, F{}, ST{});
            }
        }
    }

#ifdef SN_ROW_TIMING
    if (lane == 0)
        for (int k = 0; k < 8; ++k) atomicAdd(&sn_row_cycles[k], rt_acc[k]);
#endif
    // dst row h-1 := K[nk-1] when the top field is kept, SangNom2.cpp:380-385
    if (a.offset == 0 && bottom) {
        const Raw q = load_raw(src_line + (nk - 1) * src_step);
        keep((2 * nk - 1) * a.dst_pitch, q);
    }
}

static int virtual_waves_for(int nl) { return strips_for(nl); }

}  // namespace v3

// This file is compiled twice (csrc/Makefile).  The sweeps of planes on their own (kPlain, kPadded) go into an object
// of their own, built with -mllvm -amdgpu-sched-strategy=max-ilp: +1.7 % on them, but that scheduler makes the
// pool-coupled modes (which live at the register limit) spill, so those keep the default one.
template <int MODE, bool BAND = false>
static hipError_t launch_mode(hipStream_t st, const v3::Args& a, int nframes)
{
    const int g = v3c::group_of(a.nw);
    const int lds = v3::lds_bytes(a.nw, MODE) * g;
    hipError_t e = hipSuccess;
#define SN_LAUNCH(NW)                                                                                              \
    case NW:                                                                                                       \
        if (lds > 64 * 1024)                                                                                       \
            e = hipFuncSetAttribute((const void*)v3::k_fused_u8_v3<NW, MODE, BAND>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
        if (e == hipSuccess)                                                                                       \
            hipLaunchKernelGGL((v3::k_fused_u8_v3<NW, MODE, BAND>), dim3((nframes + g - 1) / g, BAND ? a.nbands : 1), dim3(NW * g * 64), lds, st, a); \
        break;
    switch (a.nw) {
        SN_LAUNCH(1) SN_LAUNCH(2) SN_LAUNCH(3) SN_LAUNCH(4) SN_LAUNCH(5) SN_LAUNCH(6) SN_LAUNCH(7) SN_LAUNCH(8)
    default: return hipErrorInvalidValue;
    }
#undef SN_LAUNCH
    if (e != hipSuccess) return e;
    return hipGetLastError();
}

#if defined(SN_ROW_TIMING) && (defined(SN_TU_PLAIN) || defined(SN_ROW_TIMING_COUPLED))  // (one of the two objects of this file carries the counters)
extern "C" __attribute__((visibility("default"))) int sn_debug_row_cycles(unsigned long long out[8], int reset)
{
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(v3::sn_row_cycles), 8 * sizeof(unsigned long long)) != hipSuccess) return 1;
    if (reset) {
        const unsigned long long zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(v3::sn_row_cycles), zero, sizeof zero) != hipSuccess) return 1;
    }
    return 0;
}
#endif
#ifdef SN_TU_PLAIN
hipError_t launch_fused_u8_v3_plain(hipStream_t st, const v3c::Args& a, int nframes, int mode)
{
    if (a.nbands > 1) return mode == v3::kPlain ? launch_mode<v3::kPlain, true>(st, a, nframes) : hipErrorInvalidValue;  // only planes on their own are cut
    return mode == v3::kPadded ? launch_mode<v3::kPadded>(st, a, nframes) : launch_mode<v3::kPlain>(st, a, nframes);
}
#else

bool fused_v3_plane_ok(int w)
{
    if (w % 32 != 0) return false;
    return v3::virtual_waves_for(w / v3::PXL) <= 2 * v3::kMaxWaves;
}

int fused_v3_waves(int sweep_w) { return (v3::virtual_waves_for(sweep_w / v3::PXL) + 1) / 2; }

// bytes of one scratch pool of one frame: [9][rows][threads][16]
int64_t fused_v3_pool_bytes(int sweep_w, int rows) { return (int64_t)kBuffers * rows * fused_v3_waves(sweep_w) * 64 * 16; }

// A pool row is [2 chunk kinds][threads][8 bytes]: kind 0 = the eight columns of the thread's strip in the low halves (strip
// `wave`), kind 1 = those of its strip in the high halves (strip `wave + nw`), as PoolIO::store packs them.  Cells outside the dependency cone are
// never written; `out` keeps what the caller put there.
void fused_v3_pool_unpack(const uint32_t* raw, int sweep_w, int rows, uint8_t* out)
{
    using namespace v3c;
    const int nl = sweep_w / PXL, nvw = v3::virtual_waves_for(nl), nw = (nvw + 1) / 2, nt = nw * 64;
    const uint8_t* bytes = reinterpret_cast<const uint8_t*>(raw);
    for (int64_t br = 0; br < (int64_t)kBuffers * rows; ++br)
        for (int t = 0; t < nt; ++t)
            for (int h = 0; h < 2; ++h) {
                const int wave = t / 64, lane = t % 64, vw = wave + h * nw;
                const int gl = vw == 0 ? lane : kFirst + kInner * (vw - 1) + (lane - GH);
                const bool ghost = vw == 0 ? (nvw > 1 && lane >= 64 - GH) : (lane < GH || (lane >= 64 - GH && vw < nvw - 1));
                if (vw >= nvw || ghost || gl >= nl) continue;
                const uint8_t* d = bytes + br * nt * 16 + (int64_t)h * nt * 8 + t * 8;
                uint8_t* o = out + br * sweep_w + gl * PXL;
                for (int k = 0; k < PXL; ++k) o[k] = d[k];
            }
}

// pool == nullptr: a plane on its own (kPlain).  Otherwise pool->mode selects kLumaSpill / kChroma and
// pool->sweep_w is the luma width the sweep covers (p describes the plane being interpolated).
hipError_t launch_fused_u8_v3(hipStream_t st, const PlaneArgs& p, double threshold, int nframes, const FusedPool* pool)
{
#ifdef SN_EXPERIMENT_V4  // A/B builds of tools/experiments only
    if (!pool && fused_v4_plane_ok(p.w)) return launch_fused_u8_v4(st, p, threshold, nframes);
#endif
    v3::Args a{};
    a.src = p.src;
    a.dst = p.dst;
    a.src_frame_stride = p.src_frame_stride;
    a.dst_frame_stride = p.dst_frame_stride;
    a.src_pitch = p.src_pitch;
    a.dst_pitch = p.dst_pitch;
    a.w = pool && pool->mode != v3::kPlain ? pool->sweep_w : p.w;
    a.nk = p.h_out / 2;
    a.offset = p.offset;
    a.dh = p.dh;
    a.thr = (int)threshold;
    a.nl = a.w / v3::PXL;
    a.nvw = v3::virtual_waves_for(a.nl);
    a.nw = (a.nvw + 1) / 2;
    a.src_bytes = (int)((int64_t)p.src_pitch * p.h_in);
    a.dst_bytes = (int)((int64_t)p.dst_pitch * p.h_out);

    a.turn_shift = v3c::turn_shift_for(a.nk, a.nw * v3c::group_of(a.nw), 1);
    a.nframes = nframes;
    if (!pool) return launch_fused_u8_v3_plain(st, a, nframes, v3::kPlain);
    if (pool->nbands > 1) {
        a.band_rows = pool->band_rows;
        a.band_warm = pool->band_warm;
        a.nbands = pool->nbands;
        a.band_state = pool->band_state;
        a.band_flags = pool->band_flags;
        a.band_reset = pool->band_reset;
    }
    if (pool->mode == v3::kPlain) return launch_fused_u8_v3_plain(st, a, nframes, v3::kPlain);
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
    if (pool->mode == v3::kPadded) return launch_fused_u8_v3_plain(st, a, nframes, v3::kPadded);
    a.pool_row_bytes = pool->mode == v3::kLumaSpill ? pool->pool_row_bytes : 0;
    if (a.nbands > 1) {  // of the pool-coupled sweeps only the luma one is cut (sn_fused_v3_common.h)
        if (pool->mode != v3::kLumaSpill) return hipErrorInvalidValue;
        return launch_mode<v3::kLumaSpill, true>(st, a, nframes);
    }
    if (pool->mode == v3::kLumaSpill) return launch_mode<v3::kLumaSpill>(st, a, nframes);
    if (!pool->pool_out) return launch_mode<v3::kChromaLast>(st, a, nframes);
    return launch_mode<v3::kChroma>(st, a, nframes);
}

#endif  // SN_TU_PLAIN

}  // namespace sn
