// sn_fused_f32_v3.hip -- the fused sweep for 32-bit float samples (Y32, YUV444PS; 4:2:0 / 4:2:2 float clips through
// the same hand-off modes as the integer sweeps, see Mode in sn_fused_v3_common.h and the stale-wave path below).
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

#include <cstring>

namespace sn {
namespace f32 {

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
    float inside;    // kPadded: 1 where the lane's columns belong to the plane, 0 in the padding
    bool inside_b;   // chroma modes: the lane's columns lie inside the chroma region (else: stale values instead of costs)
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

// Scratch pools of the chroma coupling (Mode): [buffer][row][thread][8 floats] -- a thread re-reads what the thread with the
// same columns wrote (ghost lanes read their owner's slot); only slots inside the dependency cone (Args::cone_*) move.
struct PoolIO {
    __amdgpu_buffer_rsrc_t rin, rout;
    int v_a;    // slot this lane reads: its own, or its owner's for ghost lanes
    int v_out;  // slot this lane writes
    int row_stride, buf_stride;
    struct RawPair {
        u32x4 a, b;
    };
    __device__ __forceinline__ RawPair issue(int b, int row, int va) const  // va: v_a or kOutOfRange (-> zeros)
    {
        const int soff = b * buf_stride + row * row_stride;
        RawPair q;
        q.a = __builtin_amdgcn_raw_buffer_load_b128(rin, va, soff, 0);
        q.b = __builtin_amdgcn_raw_buffer_load_b128(rin, va, soff + 16, 0);
        return q;
    }
    // kPacked: va already holds the fetching lane's buffer (b * buf_stride) and the slot of the lane it fetches for
    __device__ __forceinline__ RawPair issue_packed(int row, int va) const
    {
        RawPair q;
        q.a = __builtin_amdgcn_raw_buffer_load_b128(rin, va, row * row_stride, 0);
        q.b = __builtin_amdgcn_raw_buffer_load_b128(rin, va, row * row_stride + 16, 0);
        return q;
    }
    __device__ __forceinline__ void finish(const RawPair& q, float (&P)[PXL]) const
    {
        P[0] = flt(q.a.x); P[1] = flt(q.a.y); P[2] = flt(q.a.z); P[3] = flt(q.a.w);
        P[4] = flt(q.b.x); P[5] = flt(q.b.y); P[6] = flt(q.b.z); P[7] = flt(q.b.w);
    }
    __device__ __forceinline__ void store(int b, int row, int vout, const float (&O)[PXL]) const
    {
        u32x4 lo, hi;
        lo.x = bits(O[0]); lo.y = bits(O[1]); lo.z = bits(O[2]); lo.w = bits(O[3]);
        hi.x = bits(O[4]); hi.y = bits(O[5]); hi.z = bits(O[6]); hi.w = bits(O[7]);
        const int soff = b * buf_stride + row * row_stride;
        store_b128(lo, rout, vout, soff);
        store_b128(hi, rout, vout, soff + 16);
    }
};

struct RowCtx {
    int r;
    int vin;       // chroma modes: voffset of the loads of row r + 1 (out of range: row missing or outside the cone)
    int vin_next;  // ... and of row r + 2 (kPacked: of the packed fetch)
    const uint4* packed_lds;  // kPacked: where the lane that fetched buffer 0 for this lane left the first half of its values
    int packed_half;          // ... and how many uint4 further the second half
    uint4* packed_out;        // kPacked, kChroma: where this lane leaves buffer 0's O for the row's one store (null: hands nothing on)
    int vout;      // voffset of this row's stores (out of range: not kept)
    bool any_out;  // wave-uniform: some lane of this wave stores in this row
};

// the order in which a row visits the buffers: lowest rung of the ladder first (a later buffer with an equal cost takes over)
__host__ __device__ constexpr int visit(int i)
{
    constexpr int o[9] = {0, 8, 1, 7, 2, 6, 3, 5, 4};
    return o[i];
}

// kPadded: a plane narrower than its pool stride on a zero-filled pool (sn_config.fresh_pool): costs are zero in the padding
// columns -- a multiplication by 1 or 0 (costs are finite and not negative, so that is exact).  Chroma modes: outside the
// chroma region the cost of the next row is what the previous pass left there (`stale`, fetched one buffer ahead).
// How a wave of the region gets the previous pass's values in a row of a chroma mode (a wave runs whole loops of one kind):
//   kQuiet   no lane takes one (a wave inside the region: three of the four of a 4:2:0 pass): no fetch -- even one whose every
//            lane is out of range is a trip to the texture unit the step would wait for -- and no select;
//   kPacked  only the last lanes of the wave do (the wave the region ends in: two real lanes and the two ghosts).  A fetch per
//            buffer and step, one step ahead, is all the registers allow and less than the trip to memory, and the other region
//            waves wait for this one at every seam barrier.  Instead lane 9 g + b fetches buffer b of the g-th such lane, ONE
//            fetch for the whole row, issued a row ahead; when a row begins the fetched values go to LDS (the parking slots of
//            the workgroup's last wave, which lies right of the region and parks nothing) and the fetch of the row after next
//            starts BEFORE this row's stores -- a wait for it later waits for nothing younger; a step reads its buffer's eight
//            values as two ds_read_b128;
//   kFetch   any lane may (rows below the region, planes whose region ends inside a wave's real lanes): a fetch per buffer.
enum Fetch { kQuiet = 0, kFetch = 1, kPacked = 2 };
constexpr int kMaxPackedLanes = 7;  // 7 x 9 buffers <= 64 lanes

template <int I, bool S1, int MODE, int FETCH>
__device__ __forceinline__ void buffer_step(float (&A)[PXL], float (&vmin)[PXL], unsigned (&rank)[PXL], const Line& n,
                                            const Line& nn, const LaneRole& role, const PoolIO& io, const RowCtx& rc, PoolIO::RawPair& stale)
{
    constexpr int BUF = visit(I);
    float D[PXL], S[PXL], Bx[PXL], O[PXL];
    if constexpr (chroma_mode(MODE) && FETCH == kQuiet) {
#pragma unroll
        for (int j = 0; j < PXL; ++j) D[j] = S1 ? cost<BUF>(n, nn, j) : 0.0f;
    } else if constexpr (chroma_mode(MODE) && FETCH == kPacked) {
        __builtin_amdgcn_sched_barrier(0);  // one step's reads at a time: hoisted over earlier steps they spill
        const uint4 lo = rc.packed_lds[BUF], hi = rc.packed_lds[rc.packed_half + BUF];
        D[0] = flt(lo.x); D[1] = flt(lo.y); D[2] = flt(lo.z); D[3] = flt(lo.w);
        D[4] = flt(hi.x); D[5] = flt(hi.y); D[6] = flt(hi.z); D[7] = flt(hi.w);
        if constexpr (S1) {
#pragma unroll
            for (int j = 0; j < PXL; ++j) D[j] = role.inside_b ? cost<BUF>(n, nn, j) : D[j];
        }
    } else if constexpr (chroma_mode(MODE)) {
        io.finish(stale, D);
        __builtin_amdgcn_sched_barrier(0);  // the next fetch goes into the registers this one has just left
        if constexpr (I + 1 < kBuffers) stale = io.issue(visit(I + 1), rc.r + 1, rc.vin);
        else stale = io.issue(visit(0), rc.r + 2, rc.vin_next);  // the next row's first buffer: the fetch never starts a row late
        if constexpr (S1) {
#pragma unroll
            for (int j = 0; j < PXL; ++j) D[j] = role.inside_b ? cost<BUF>(n, nn, j) : D[j];
        }
    } else {
#pragma unroll
        for (int j = 0; j < PXL; ++j) D[j] = S1 ? (MODE == kPadded ? cost<BUF>(n, nn, j) * role.inside : cost<BUF>(n, nn, j)) : 0.0f;
    }
#pragma unroll
    for (int j = 0; j < PXL; ++j) S[j] = A[j] + D[j];  // (O[r-1] + D[r]) + D[r+1]
    if (role.edge_wave) box7<true>(S, Bx, role);
    else box7<false>(S, Bx, role);
#pragma unroll
    for (int j = 0; j < PXL; ++j) {
        O[j] = Bx[j] * 0.0625f;
        A[j] = O[j] + D[j];
        const bool le = O[j] <= vmin[j];
        vmin[j] = le ? O[j] : vmin[j];
        rank[j] = le ? rank_of<BUF>() : rank[j];
    }
    if constexpr (MODE == kLumaSpill) {
        if (rc.any_out) io.store(BUF, rc.r, rc.vout, O);  // packed only where a lane of the wave stores
    } else if constexpr (MODE == kChroma) {
        if constexpr (FETCH == kPacked) {
            // the few lanes that hand a value on leave it in LDS; the row's end stores all nine buffers' with one store (the
            // way they were fetched): nine stores of two live lanes each queued behind the stale waves' full ones
            if (rc.packed_out) {
                rc.packed_out[BUF] = make_uint4(bits(O[0]), bits(O[1]), bits(O[2]), bits(O[3]));
                rc.packed_out[rc.packed_half + BUF] = make_uint4(bits(O[4]), bits(O[5]), bits(O[6]), bits(O[7]));
            }
        } else {
            if (FETCH != kQuiet || rc.any_out) io.store(BUF, rc.r, rc.vout, O);
        }
    }
}

// A wave of a chroma sweep whose columns all lie outside the chroma region: no lines, no stage 1, no stage 3 (as in
// sn_fused_u16_v3.hip: stale_wave_sweep).
template <int BUF>
__device__ __forceinline__ void stale_buffer_step(float (&A)[PXL], const PoolIO::RawPair& stale, const LaneRole& role, const PoolIO& io, int r,
                                                  int vout, bool vout_any)
{
    float D[PXL], S[PXL], Bx[PXL], O[PXL];
    io.finish(stale, D);
#pragma unroll
    for (int j = 0; j < PXL; ++j) S[j] = A[j] + D[j];
    if (role.edge_wave) box7<true>(S, Bx, role);
    else box7<false>(S, Bx, role);
#pragma unroll
    for (int j = 0; j < PXL; ++j) {
        O[j] = Bx[j] * 0.0625f;
        A[j] = O[j] + D[j];
    }
    if (vout_any) io.store(BUF, r, vout, O);
}

// buffers 0 .. reg_buffers-1 keep A in VGPRs, the others in LDS between their steps (no scratch spills).  Built without the
// SLP vectoriser the plain sweep needs ~212 VGPRs with two: six fit without scratch, seven touch the 256 limit; the
// pool-coupled modes carry the stale row in flight and the row to store, so they keep fewer.  Five in the chroma modes since
// round 4: the kPacked rows hold the row's packed fetch AND a step's eight values (six: 44 - 64 bytes of scratch per lane and
// a V pass 8 % slower instead of 9 % faster; what is left, 16 bytes in kChromaLast, sits in the two rows of kFetch).
__host__ __device__ constexpr int reg_buffers(int mode) { return mode == kPadded ? 4 : chroma_mode(mode) ? 5 : 6; }

template <int NT, int kRegBuffers>
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

// S3: the row has an interpolated line (stage 3); a chroma sweep covers rows below its plane without one.
template <bool S1, bool S3, int MODE, int FETCH, int NT>
__device__ __forceinline__ Out row_step(float (&A)[reg_buffers(MODE)][PXL], const Parked<NT, reg_buffers(MODE)>& pk, int tid, const Line& n,
                                        const Line& nn, const LaneRole& role, float aaf, const PoolIO& io, const RowCtx& rc, PoolIO::RawPair& stale)
{
    // stale (chroma modes): the previous pass's row r + 1 of the first buffer, in flight since the last step of the row before
    constexpr int kRegBuffers = reg_buffers(MODE);
    float vmin[PXL];
    unsigned rank[PXL];
#pragma unroll
    for (int j = 0; j < PXL; ++j) {
        vmin[j] = __builtin_inff();
        rank[j] = 0u;
    }
    auto run = [&](auto idx) {
        constexpr int I = decltype(idx)::value;
        constexpr int B = visit(I);
        if constexpr (B < kRegBuffers) {
            buffer_step<I, S1, MODE, FETCH>(A[B], vmin, rank, n, nn, role, io, rc, stale);
        } else {
            float t[PXL];
            pk.load_A(tid, B, t);
            buffer_step<I, S1, MODE, FETCH>(t, vmin, rank, n, nn, role, io, rc, stale);
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

__host__ __device__ constexpr int lds_bytes(int nw, int mode)
{
    return (4 + (kBuffers - reg_buffers(mode)) * 2) * 16 * nw * 64 + 2 * (2 * nw - 2) * GH * kBuffers * PXL * 4;
}

template <int NW, int MODE, bool BAND>
__global__ void __launch_bounds__(NW * group_of(NW) * 64, 2) k_fused_f32_v3(Args a, float aaf)
{
    extern __shared__ __align__(16) unsigned char lds_raw[];
    constexpr int NT = NW * 64;
    constexpr int kRegBuffers = reg_buffers(MODE);
    constexpr bool PADDED = MODE == kPadded;
    const int sub = __builtin_amdgcn_readfirstlane((int)threadIdx.x) / (NW * 64);
    const int f_raw = (int)blockIdx.x * group_of(NW) + sub;
    const int f = f_raw < a.nframes ? f_raw : a.nframes - 1;  // see sn_fused_u16_v3.hip
    const int tid = (int)threadIdx.x - sub * (NW * 64);
    Parked<NT, kRegBuffers> parked;
    parked.v = reinterpret_cast<uint4*>(lds_raw + sub * lds_bytes(NW, MODE));
    parked.a = parked.v + 4 * NT;
    Mailbox<NW> mb;
    mb.h = reinterpret_cast<unsigned*>(parked.a + (kBuffers - kRegBuffers) * 2 * NT);
    const int wave = tid >> 6;
    const int lane = tid & 63;
#ifdef SN_WAVE_TIMING
    const unsigned long long wt_start = __builtin_amdgcn_s_memtime();
    unsigned long long wt_barrier = 0;
#endif

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
    const int line_w = has_region(MODE) ? a.region_w : a.w;  // width of the source / destination plane
    const bool line_live = live && x0 < line_w;
    const bool line_real = real && x0 < line_w;
    LaneRole role;
    role.first = live && gl == 0;
    role.last = live && gl == a.nl - 1;
    role.line_last = line_live && x0 + PXL == line_w;
    role.inside = line_live ? 1.0f : 0.0f;
    role.inside_b = line_live;
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
    auto keep = [&](int row_off, const Raw& q, bool on = true) {  // GetFrame's field copy: the lane's own 8 pixels
        const bool early = role.first;                             // loaded from column 0: own pixels come first
        store_b128(early ? q.q[0] : q.q[1], rd, on ? vstore : kOutOfRange, row_off);
        store_b128(early ? q.q[1] : q.q[2], rd, on ? vstore : kOutOfRange, row_off + 16);
    };
    auto put = [&](int row_off, const Out& o) {
        store_b128(o.lo, rd, vstore, row_off);
        store_b128(o.hi, rd, vstore, row_off + 16);
    };

    // scratch pools of the chroma coupling
    PoolIO io{};
    if constexpr (has_pools(MODE)) {
        const bool linear = MODE == kLumaSpill && a.pool_row_bytes > 0;  // the pool path's layout
        io.row_stride = linear ? a.pool_row_bytes : NT * 32;
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
        io.v_a = ta * 32;
        io.v_out = real ? (linear ? x0 * 4 : tid * 32) : kOutOfRange;
    }
    // Does this lane's slot of pool row q matter (Args::cone_*)?  Its first column is 8 * lane + 480 * wave.
    auto in_cone_at = [&](int q, int extra, int xa) -> bool {
        const int lim = a.cone_w + 3 * (a.cone_nr - q + 2) + extra;
        const int cols = lim < a.w ? lim : a.w;
        return xa < cols && (q > a.cone_nr || xa + PXL > a.cone_w);
    };
    auto in_cone = [&](int q, int extra) -> bool { return in_cone_at(q, extra, (lane << 3) + wave * (kInner * PXL)); };

    // seam exchange: lanes 60, 61 feed the next wave's left ghosts, lanes 2, 3 the previous wave's right ghosts
    const bool pub_right = lane >= 64 - 2 * GH && lane < 64 - GH && wave < NW - 1;
    const bool pub_left = lane >= GH && lane < 2 * GH && wave > 0;
    const bool recv = ghost && live;
    const int slot = lane < GH ? lane : lane >= 64 - GH ? lane - (64 - GH) : pub_right ? lane - (64 - 2 * GH) : lane - GH;

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

    if constexpr (chroma_mode(MODE)) {
        // Waves entirely to the right of the chroma region only re-smooth what the previous pass left (all nine buffers'
        // state in registers, the stale row of the next buffer in flight) and leave once every column from their first
        // one on lies outside the dependency cone; their SIMD partners are waves of the region (wave i and i + 4).
        if (wave * (kInner * PXL) >= a.region_w) {
            const int x_wave = wave * (kInner * PXL);
            float As[kBuffers][PXL];
            const int v1 = (a.rows_in >= 1 && in_cone(1, a.cone_in)) ? io.v_a : kOutOfRange;
#pragma unroll
            for (int b = 0; b < kBuffers; ++b) io.finish(io.issue(b, 1, v1), As[b]);
            // the stale rows are fetched a whole row ahead (one row of nine buffers is 72 registers; a buffer's slot is refilled as soon
            // as its step has taken the values out): two steps ahead, as until round 4, was less than the trip to memory
            PoolIO::RawPair ahead[kBuffers];
            {
                const int v2 = (a.rows_in >= 2 && in_cone(2, a.cone_in)) ? io.v_a : kOutOfRange;
#pragma unroll
                for (int b = 0; b < kBuffers; ++b) ahead[b] = io.issue(b, 2, v2);
            }
            for (int r = 1; r <= sweep; ++r) {
#ifdef SN_X_BALANCED  // DIAGNOSTIC (wrong results): every stale wave lives 5/9 of the sweep, the load of a balanced hand-over
                if (r > sweep * 5 / 9) { SN_WT_FLUSH(); return; }
#else
                if (x_wave >= a.cone_w + 3 * (a.cone_nr - (r - 1) + 2) + a.cone_in) {  // outside for good
                    SN_WT_FLUSH();
                    return;
                }
#endif
                const int vin = (r + 2 <= a.rows_in && in_cone(r + 2, a.cone_in)) ? io.v_a : kOutOfRange;
                if (r > 1 && (r - 1) % K == 0) {
                    SN_SYNC();
                    if (recv) {
                        const unsigned* from = mb.at((r / K) & 1, wave, lane < GH ? 0 : 1, slot);
#pragma unroll
                        for (int b = 0; b < kBuffers; ++b)
#pragma unroll
                            for (int j = 0; j < PXL; ++j) As[b][j] = flt(from[b * PXL + j]);
                    }
                }
                const int vout = (r <= a.rows_out && in_cone(r, a.cone_out)) ? io.v_out : kOutOfRange;
                const bool vout_any = __builtin_amdgcn_readfirstlane(__any(vout != kOutOfRange) ? 1 : 0) != 0;
                auto run = [&](auto buf) {
                    constexpr int B = decltype(buf)::value;
                    const PoolIO::RawPair cur = ahead[B];
                    ahead[B] = io.issue(B, r + 2, vin);
                    stale_buffer_step<B>(As[B], cur, role, io, r, vout, vout_any);
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
                if (r < sweep && r % K == 0) {
                    if (pub_right || pub_left) {
                        unsigned* to = pub_right ? mb.at(((r + 1) / K) & 1, wave + 1, 0, slot) : mb.at(((r + 1) / K) & 1, wave - 1, 1, slot);
#pragma unroll
                        for (int b = 0; b < kBuffers; ++b)
#pragma unroll
                            for (int j = 0; j < PXL; ++j) to[b * PXL + j] = bits(As[b][j]);
                    }
                }
            }
            SN_WT_FLUSH();
            return;
        }
    }

    // a band copies the kept lines ra .. rb (the top band line 0 as well)
    Line L0, L1;
    Raw q0 = load_raw(src_line + (r0 - 1) * src_step);
    Raw q1 = nk > 1 ? load_raw(src_line + r0 * src_step) : q0;
    keep(dst_line, q0, top);
    if (a.offset == 1) keep(0, q0, top);  // the line that cannot be interpolated, SangNom2.cpp:386-391
    if (nk > 1) keep(dst_line + r0 * dst_step, q1, r0 == ra);
    unpack(L0, q0, role);
    unpack(L1, q1, role);
    parked.park(tid, L0);

    // A[1] = O[0] + P[1] = P[1] (pool row 0 is never written: zero, and 0 + x = x)
    float A[kRegBuffers][PXL];
    const bool first_in = chroma_mode(MODE) && a.rows_in >= 1 && in_cone(1, a.cone_in);
    auto init_buf = [&](auto buf) {
        constexpr int B = decltype(buf)::value;
        float t[PXL];
        if constexpr (chroma_mode(MODE)) {
            io.finish(io.issue(B, 1, first_in ? io.v_a : kOutOfRange), t);
            if (nr > 0) {
#pragma unroll
                for (int j = 0; j < PXL; ++j) t[j] = role.inside_b ? cost<B>(L0, L1, j) : t[j];
            }
        } else {
#pragma unroll
            for (int j = 0; j < PXL; ++j) t[j] = r0 <= nr ? (PADDED ? cost<B>(L0, L1, j) * role.inside : cost<B>(L0, L1, j)) : 0.0f;
        }
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

    int src_next = src_line + (r0 + 1) * src_step;
    int dst_keep = dst_line + (r0 + 1) * dst_step;
    int out_row = dst_line + a.dst_pitch + (r0 - 1) * dst_step;
    Raw qn = r0 + 1 <= nr ? load_raw(src_next) : q1;
    src_next += src_step;

    PoolIO::RawPair stale_next{};
    int packed_x = 0, packed_va = kOutOfRange;  // kPacked (set before its loop)
    const uint4* packed_lds = parked.v;
    uint4* packed_out = nullptr;
    int packed_vout = kOutOfRange;
    TurnTaking turns;
    turns.init(a.turn_shift);
    // fetch_tag (chroma modes): false for the rows in which no lane of this wave takes a stale value (buffer_step)
    auto step = [&](int r, Line& n, Line& nn, auto s1_tag, auto s3_tag, auto fetch_tag) {
        constexpr bool S1 = decltype(s1_tag)::value;
        constexpr bool S3 = decltype(s3_tag)::value;
        turns.update((r - 1) % K);
        Raw qnext = qn;
        if constexpr (S1) {
            unpack(nn, qn, role);  // waits for the line prefetched one row ago
            keep(dst_keep, qn, !BAND || (r + 1 >= ra && r < rb));
            dst_keep += dst_step;
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
        RowCtx rc;
        rc.r = r;
        rc.vin = rc.vout = kOutOfRange;
        rc.any_out = false;
        rc.vin_next = kOutOfRange;
        rc.packed_lds = packed_lds;
        rc.packed_half = NT;
        rc.packed_out = packed_out;
        constexpr int kKind = chroma_mode(MODE) ? decltype(fetch_tag)::value : (int)kQuiet;
        if constexpr (chroma_mode(MODE) && kKind == kPacked) rc.vin_next = (r + 2 <= a.rows_in && in_cone_at(r + 2, a.cone_in, packed_x)) ? packed_va : kOutOfRange;
        else if constexpr (chroma_mode(MODE)) rc.vin_next = (r + 2 <= a.rows_in && in_cone(r + 2, a.cone_in)) ? io.v_a : kOutOfRange;
        if constexpr (chroma_mode(MODE)) rc.vin = (r + 1 <= a.rows_in && in_cone(r + 1, a.cone_in)) ? io.v_a : kOutOfRange;
#ifdef SN_X_REGION_NOSTALE  // DIAGNOSTIC (wrong results): what do the stale loads of the region's ghost lanes cost?
        rc.vin = kOutOfRange;
#endif
        if constexpr (has_pools(MODE)) {
            rc.vout = (r <= a.rows_out && (!BAND || r >= ra) && in_cone(r, a.cone_out)) ? io.v_out : kOutOfRange;
            rc.any_out = __builtin_amdgcn_readfirstlane(__any(rc.vout != kOutOfRange) ? 1 : 0) != 0;
        }
        if constexpr (kKind == kPacked) {  // this row's values to LDS, the next row's fetch on its way
            parked.v[(NW - 1) * 64 + lane] = make_uint4(stale_next.a.x, stale_next.a.y, stale_next.a.z, stale_next.a.w);
            parked.v[NT + (NW - 1) * 64 + lane] = make_uint4(stale_next.b.x, stale_next.b.y, stale_next.b.z, stale_next.b.w);
            __builtin_amdgcn_sched_barrier(0);
            stale_next = io.issue_packed(r + 2, rc.vin_next);
        }
        const Out o = row_step<S1, S3, MODE, kKind>(A, parked, tid, n, nn, role, aaf, io, rc, stale_next);
        if constexpr (kKind == kPacked && MODE == kChroma) {  // the row's one store: lane 9 g + b stores buffer b of lane first + g
            const uint4 lo = parked.v[2 * NT + (NW - 1) * 64 + lane], hi = parked.v[3 * NT + (NW - 1) * 64 + lane];
            const int vo = (r <= a.rows_out && in_cone_at(r, a.cone_out, packed_x)) ? packed_vout : kOutOfRange;
            u32x4 d0, d1;
            d0.x = lo.x; d0.y = lo.y; d0.z = lo.z; d0.w = lo.w;
            d1.x = hi.x; d1.y = hi.y; d1.z = hi.z; d1.w = hi.w;
            store_b128(d0, io.rout, vo, r * io.row_stride);
            store_b128(d1, io.rout, vo, r * io.row_stride + 16);
        }
        if constexpr (S3) put(out_row, o);
        out_row += dst_step;
        if constexpr (S1) parked.park(tid, n);  // n is the next row's c
        if (r < sweep && r % K == 0) {
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

    if constexpr (BAND) {
        // the state the band holds on entering a row: real columns only, as bit patterns
        auto leave_state = [&](int which) {
            uint32_t* to = a.band_state + ((int64_t)(f * a.nbands + (int)blockIdx.y) * 2 + which) * (kBuffers * PXL * NT) + tid;
#pragma unroll
            for (int b = 0; b < kRegBuffers; ++b)
#pragma unroll
                for (int j = 0; j < PXL; ++j) to[(b * PXL + j) * NT] = real ? bits(A[b][j]) : 0u;
#pragma unroll
            for (int b = kRegBuffers; b < kBuffers; ++b) {
                float t[PXL];
                parked.load_A(tid, b, t);
#pragma unroll
                for (int j = 0; j < PXL; ++j) to[(b * PXL + j) * NT] = real ? bits(t[j]) : 0u;
            }
        };
        int r = r0;
        for (; r < ra; ++r) {  // the run-up: nothing interpolated
            step(r, L1, L0, T{}, F{}, T{});
            L1 = L0;
        }
        leave_state(0);
        const int own_next = rb < nr ? rb + 1 : nr;
        for (; r < own_next; ++r) {
            step(r, L1, L0, T{}, T{}, T{});
            L1 = L0;
        }
        if (rb == nr) step(nr, L1, L0, F{}, T{}, T{});
        else leave_state(1);
    } else {
        using Quiet = std::integral_constant<int, kQuiet>;
        using Packed = std::integral_constant<int, kPacked>;
        int from = 1;  // rows [1, from) have been swept
        if constexpr (chroma_mode(MODE)) {
            // rows 1 .. calm - 1 fetch pool rows 2 .. calm <= cone_nr, where only lanes right of the region take stale values
            const int calm = nr < a.cone_nr ? nr : a.cone_nr;
            const int xa = (lane << 3) + wave * (kInner * PXL);
            const int lim = a.cone_w + 3 * (a.cone_nr - 2 + 2) + a.cone_in;  // in_cone(2, cone_in): the widest of those rows
            const bool beyond = xa < (lim < a.w ? lim : a.w) && xa + PXL > a.cone_w;
            const unsigned long long takers = __ballot(beyond);  // wave-uniform
            // (three loops one after the other, two of them empty: as arms of a branch their register allocations collide)
            const bool packed = takers != 0 && wave < NW - 1 && (takers >> (64 - kMaxPackedLanes)) << (64 - kMaxPackedLanes) == takers;
            const int quiet_end = takers == 0 ? calm : 1, packed_end = packed ? calm : 1;
            for (int r = 1; r < quiet_end; ++r) {
                step(r, L1, L0, T{}, T{}, Quiet{});
                L1 = L0;
            }
            {
                // kPacked: the takers are among the wave's last kMaxPackedLanes lanes (the others of those fetch too; the select
                // drops their values)
                const int first = 64 - kMaxPackedLanes;
                const int g = lane / kBuffers, b = lane - g * kBuffers;  // this lane fetches buffer b for lane first + g
                const int t = first + g;
                const int owner = t < 64 - GH ? wave * 64 + t : (wave + 1) * 64 + GH + (t - (64 - GH));  // ghosts read their owner's slot
                packed_x = (t << 3) + wave * (kInner * PXL);
                packed_va = packed && g < kMaxPackedLanes ? owner * 32 + b * io.buf_stride : kOutOfRange;
                packed_lds = parked.v + (NW - 1) * 64 + (lane >= first ? kBuffers * (lane - first) : 0);
                packed_out = lane >= first ? parked.v + 2 * NT + (NW - 1) * 64 + kBuffers * (lane - first) : nullptr;
                packed_vout = packed && g < kMaxPackedLanes && t < 64 - GH ? (wave * 64 + t) * 32 + b * io.buf_stride : kOutOfRange;  // ghosts store nothing
                stale_next = io.issue_packed(2, (2 <= a.rows_in && in_cone_at(2, a.cone_in, packed_x)) ? packed_va : kOutOfRange);
            }
            for (int r = 1; r < packed_end; ++r) {
                step(r, L1, L0, T{}, T{}, Packed{});
                L1 = L0;
            }
            from = takers == 0 || packed ? (calm > 1 ? calm : 1) : 1;
            // the first buffer of the first row that fetches buffer by buffer
            stale_next = io.issue(visit(0), from + 1, (from + 1 <= a.rows_in && in_cone(from + 1, a.cone_in)) ? io.v_a : kOutOfRange);
        }
        for (int r = from; r < nr; ++r) {
            step(r, L1, L0, T{}, T{}, T{});
            L1 = L0;
        }
        if (nr >= 1) step(nr, L1, L0, F{}, T{}, T{});
        if constexpr (chroma_mode(MODE)) {
            for (int r = nr + 1; r <= sweep; ++r) step(r, L1, L0, F{}, F{}, T{});
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
}  // namespace f32
}  // namespace sn
extern "C" __attribute__((visibility("default"))) int sn_debug_wave_ticks_f32(unsigned long long out[5 * 8 * 2], int reset)
{
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(sn::f32::sn_wave_ticks), 5 * 8 * 2 * sizeof(unsigned long long)) != hipSuccess) return 1;
    if (reset) {
        static const unsigned long long zero[5 * 8 * 2] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(sn::f32::sn_wave_ticks), zero, sizeof zero) != hipSuccess) return 1;
    }
    return 0;
}
namespace sn {
namespace f32 {
#endif

template <int MODE, bool BAND = false>
static hipError_t launch_mode(hipStream_t st, const Args& a, float aaf, int nframes)
{
    const int g = v3c::group_of(a.nw);
    const int lds = lds_bytes(a.nw, MODE) * g;
    hipError_t e = hipSuccess;
#define SN_LAUNCH(NW)                                                                                              \
    case NW:                                                                                                       \
        if (lds > 64 * 1024)                                                                                       \
            e = hipFuncSetAttribute((const void*)k_fused_f32_v3<NW, MODE, BAND>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
        if (e == hipSuccess)                                                                                       \
            hipLaunchKernelGGL((k_fused_f32_v3<NW, MODE, BAND>), dim3((nframes + g - 1) / g, BAND ? a.nbands : 1), dim3(NW * g * 64), lds, st, a, aaf); \
        break;
    switch (a.nw) {
        SN_LAUNCH(1) SN_LAUNCH(2) SN_LAUNCH(3) SN_LAUNCH(4) SN_LAUNCH(5) SN_LAUNCH(6) SN_LAUNCH(7) SN_LAUNCH(8)
    default: return hipErrorInvalidValue;
    }
#undef SN_LAUNCH
    if (e != hipSuccess) return e;
    return hipGetLastError();
}

}  // namespace f32

bool fused_f32_plane_ok(int w)
{
    if (w % 32 != 0) return false;
    return v3c::strips_for(w / v3c::PXL) <= f32::kMaxWaves;
}

int fused_f32_waves(int sweep_w) { return v3c::strips_for(sweep_w / v3c::PXL); }

int64_t fused_f32_pool_bytes(int sweep_w, int rows) { return (int64_t)kBuffers * rows * fused_f32_waves(sweep_w) * 64 * 32; }

// Slot of thread t = the O values of the 8 columns the lane owns (PoolIO::store); ghost lanes and lanes past the
// sweep width own nothing.
void fused_f32_pool_unpack(const uint32_t* raw, int sweep_w, int rows, float* out)
{
    using namespace v3c;
    const int nl = sweep_w / PXL, nw = strips_for(nl), nt = nw * 64;
    for (int64_t br = 0; br < (int64_t)kBuffers * rows; ++br)
        for (int t = 0; t < nt; ++t) {
            const int wave = t / 64, lane = t % 64;
            const int gl = wave == 0 ? lane : kFirst + kInner * (wave - 1) + (lane - GH);
            const bool ghost = wave == 0 ? (nw > 1 && lane >= 64 - GH) : (lane < GH || (lane >= 64 - GH && wave < nw - 1));
            if (ghost || gl >= nl) continue;
            std::memcpy(out + br * sweep_w + gl * PXL, raw + (br * nt + t) * 8, 8 * sizeof(float));
        }
}

hipError_t launch_fused_f32_v3(hipStream_t st, const PlaneArgs& p, double threshold, int nframes, const FusedPool* pool)
{
    v3c::Args a{};
    a.src = p.src;
    a.dst = p.dst;
    a.src_frame_stride = p.src_frame_stride;
    a.dst_frame_stride = p.dst_frame_stride;
    a.src_pitch = p.src_pitch;
    a.dst_pitch = p.dst_pitch;
    a.w = pool && pool->mode != v3c::kPlain ? pool->sweep_w : p.w;  // sweep_w: the pool stride a narrower plane is swept over
    a.region_w = p.w;
    a.nk = p.h_out / 2;
    a.offset = p.offset;
    a.dh = p.dh;
    a.nl = a.w / v3c::PXL;
    a.nvw = v3c::strips_for(a.nl);
    a.nw = a.nvw;
    a.turn_shift = v3c::turn_shift_for(a.nk, a.nw * v3c::group_of(a.nw), 4);
    a.nframes = nframes;
    a.src_bytes = (int)((int64_t)p.src_pitch * p.h_in);
    a.dst_bytes = (int)((int64_t)p.dst_pitch * p.h_out);
    const float aaf = (float)threshold;
    if (!pool) return f32::launch_mode<v3c::kPlain>(st, a, aaf, nframes);
    if (pool->nbands > 1) {
        a.band_rows = pool->band_rows;
        a.band_warm = pool->band_warm;
        a.nbands = pool->nbands;
        a.band_state = pool->band_state;
        a.band_flags = pool->band_flags;
        a.band_reset = pool->band_reset;
        if (pool->mode == v3c::kPlain) return f32::launch_mode<v3c::kPlain, true>(st, a, aaf, nframes);
        if (pool->mode != v3c::kLumaSpill) return hipErrorInvalidValue;  // of the pool-coupled sweeps only the luma one is cut
    }
    if (pool->mode == v3c::kPlain) return f32::launch_mode<v3c::kPlain>(st, a, aaf, nframes);
    if (pool->mode == v3c::kPadded) return f32::launch_mode<v3c::kPadded>(st, a, aaf, nframes);
    a.pool_in = pool->pool_in;
    a.pool_out = pool->pool_out;
    a.pool_frame_stride = pool->frame_stride;
    a.pool_rows = pool->pool_rows;
    a.rows_in = pool->rows_in;
    a.rows_out = pool->pool_out ? pool->rows_out : 0;
    a.sweep_rows = pool->sweep_rows;
    a.cone_w = pool->cone_w;
    a.cone_nr = pool->cone_nr;
    a.cone_in = pool->cone_in;
    a.cone_out = pool->cone_out;
    a.pool_row_bytes = pool->mode == v3c::kLumaSpill ? pool->pool_row_bytes : 0;
    if (pool->mode == v3c::kLumaSpill && a.nbands > 1) return f32::launch_mode<v3c::kLumaSpill, true>(st, a, aaf, nframes);
    if (pool->mode == v3c::kLumaSpill) return f32::launch_mode<v3c::kLumaSpill>(st, a, aaf, nframes);
    if (!pool->pool_out) return f32::launch_mode<v3c::kChromaLast>(st, a, aaf, nframes);
    return f32::launch_mode<v3c::kChroma>(st, a, aaf, nframes);
}

}  // namespace sn
