// sn_fused_u8_uv.hip -- the two chroma passes of an 8-bit 4:2:0 frame as ONE sweep: U in the low halves of the packed
// registers, V in the high halves, two steps behind.
//
// Reference semantics: the nine buffers are sized for luma and shared by all planes (/root/reference/src/SangNom2.cpp:
// 287-288, 305-310), every pass smooths the WHOLE pool in place (:126-159, :267-272), and prepareBuffers_c rewrites only
// the plane's own columns and rows (:74-124).  So the U pass re-smooths, right of the chroma region, what the luma pass
// left, and the V pass what the U pass left; both results depend on it inside the dependency cone of the region
// (sn_fused_v3_common.h, Args::cone_*).  Rounds 1-3 ran that as three sweeps (luma, U, V: sn_fused_u8_v3.hip, kLumaSpill /
// kChroma / kChromaLast) with two hand-off pools in HBM and every chroma wave pairing a region strip with a stale strip.
//
// This sweep keeps the luma -> U hand-off (it reads the luma sweep's pool) and nothing else:
//   * wave w owns strip w of the luma-wide pool in BOTH halves of its registers: bits 0..15 are the U pass at pool row s,
//     bits 16..31 the V pass at row s - 2 (kSkew).  Right of the chroma region the V pass needs, at its row r, exactly what
//     the U pass produced for row r + 1 in the same columns -- the O of the SAME lane one step earlier.  The U -> V hand-off
//     never leaves the registers: no second pool, no LDS ring, no wait between the passes;
//   * waves whose columns all lie right of the region (class S) have no lines, no stage 1, no stage 3: per buffer they
//     fetch the luma pass's row (U half; a whole step ahead), merge last step's O (V half) with one v_perm per register,
//     and smooth -- 66-69 instructions per buffer, 623 per step where a region wave needs 1 063 -- and they LEAVE the
//     workgroup when their first column falls out of the cone (it only shrinks: three columns per row; both passes' cones
//     coincide at a step because the skew is two rows and the luma hand-off's cone is six columns wider);
//   * waves inside the region (class R) run the plain sweep's row body on both passes at once; the one wave that holds
//     the region's right edge (class RS; two when the region ends on a seam) does both: lines and costs where its lanes
//     are chroma, stale values where they are not (per-lane v_perm selectors, last step's O parked in LDS).  Each class is
//     a function of its own (sweep_entry, not inlined): no branch inside a buffer step, no spill in any row loop;
//   * 4:2:0 and 4:2:2 alike (4:2:2: the pool has no row below the chroma planes' last one, so U has no extra row and V's
//     last row takes nothing from it);
//   * the first kSkew steps (V not started) and the last kSkew + 1 (U finished; the rows below the region, where stale
//     values enter every column) run a masked variant of the step; everything between is branch-free per buffer.
// Seams between strips, ghost lanes, the mailbox every K rows: as in sn_fused_u8_v3.hip -- one strip per wave here, so a
// seam register is always published whole; waves that re-smooth also exchange last step's O of their seam lanes.
#include <stdlib.h>

#include <type_traits>

#include "sn_fused_u8_parts.h"
#include "sn_fused_v3_common.h"

namespace sn {
namespace uv {

using namespace v3c;
using namespace v3;

constexpr int kSkew = 2;       // steps the V pass runs behind the U pass
constexpr int kMaxWaves = 8;   // strips of the luma-wide pool: 3840 columns
constexpr unsigned kAll = 0xffffffffu;
enum WaveClass { kR = 0, kRS = 1, kS = 2 };

struct Args {
    const uint8_t* src[2];  // U, V
    uint8_t* dst[2];
    int64_t src_frame_stride[2], dst_frame_stride[2];
    int32_t src_pitch[2], dst_pitch[2];
    int32_t src_bytes[2], dst_bytes[2];
    int32_t w;         // luma width = the pool's width (the sweep covers it)
    int32_t region_w;  // chroma width
    int32_t nk;        // kept lines of a chroma plane
    int32_t offset, dh;
    int32_t thr[2];
    int32_t nl;        // real lanes of the sweep = w / 8
    int32_t nreg;      // waves 0 .. nreg-1 hold chroma columns (classes R, RS): they have a line park
    int32_t nrs;       // the last nrs of those also hold stale columns (class RS): they have an O park
    const uint8_t* pool_in;  // the luma sweep's hand-off pool (sn_fused_u8_v3.hip, PoolIO layout)
    int64_t pool_frame_stride;
    int32_t pool_rows, rows_in;
    int32_t pool_threads;    // threads of the luma sweep's workgroup (its pool row is [2 kinds][threads][8 bytes])
    int32_t sweep_u;         // last pool row of the U pass (nr_c + 1)
    int32_t nframes;
};

// LDS: [line parks: nreg waves x 15 uint4 x 64][O parks: nrs waves x 18 uint4 x 64][mailbox A][mailbox O]
__host__ __device__ constexpr int mailbox_words(int nw) { return 2 * nw * 2 * GH * kBuffers * PXL; }
__host__ __device__ inline int lds_bytes(int nw, int nreg, int nrs)
{
    return nreg * 15 * 16 * 64 + nrs * 2 * kBuffers * 16 * 64 + 2 * mailbox_words(nw) * 4;
}

struct Mail {
    unsigned* a;  // [parity][wave][side][slot][72]: A state for the ghost lane `slot` on that side of `wave`
    unsigned* o;  // ... last step's O (stale columns only matter)
    int nw;
    __device__ __forceinline__ int at(int par, int wave, int side, int slot) const
    {
        return ((((par * nw + wave) * 2 + side) * GH + slot) * (kBuffers * PXL));
    }
};

// what a step needs besides the lines (wave-uniform unless noted)
struct Step {
    int s;                    // step: U row s, V row s - kSkew
    int vin2;                 // per lane: voffset of the luma pool's row s + 2 for this lane (fetched a step ahead), or out of range
    unsigned sel[4];          // per lane: v_perm selectors that put byte k of the luma row into bits 0..7 and -- where the lane
                              // takes last step's O for the V half -- its bits 0..7 into bits 16..23
    unsigned cmask;           // MASKED: halves whose next line pair exists (costs count)
    unsigned omask;           // MASKED: halves whose O counts (the V half before its first row gives zero)
    unsigned amask;           // MASKED: halves whose D enters the next state (the U half of its last step keeps O alone)
    bool from_a;              // MASKED: the V half's stale value is the U half of A (left there by amask one step earlier)
    int slot_c, slot_n;       // line park slots of the lines above / below the interpolated one
};

struct Ctx {
    int tid, wave, lane;
    LaneRole role;
    __amdgpu_buffer_rsrc_t rs[2], rd[2], rin;
    int vload;
    int row_stride, buf_stride;  // of the luma pool
    uint4* lines;                // this wave's line park (region classes)
    uint4* opark;                // this wave's O park (class RS), else null
    Mail mb;
    bool recv_left, recv_right, pub_left, pub_right;
    int slot;
    unsigned ghost_mask;  // all ones in live ghost lanes
};

__device__ __forceinline__ u32x2 own_bytes(const RawHalf& h, bool first)
{
    u32x2 v;
    v.x = first ? h.l : h.m0;
    v.y = first ? h.m0 : h.m1;
    return v;
}

// One cost buffer of one step in a wave that holds chroma columns.
// STALE: some lanes re-smooth stale values (class RS, and the masked steps of both region classes).
// MASKED: the first kSkew and the last kSkew + 1 steps (see Step).
template <int BUF, bool STALE, bool MASKED, bool PARK, bool RC>
__device__ __forceinline__ void region_buffer_step(unsigned (&A)[PXL], unsigned (&kmin)[PXL], const WideLine& n, const WideLine& nn, const Ctx& cx,
                                                   const Step& st, const u32x2& ld)
{
    unsigned U[PXL], V[PXL], S[PXL], Bx[PXL];
#pragma unroll
    for (int j = 0; j < PXL; ++j) {
        unsigned x, y;
        cost_operands<BUF>(n, nn, j, x, y);
        U[j] = pk_sub_sat(x, y);
        V[j] = pk_sub_sat(y, x);
        if constexpr (MASKED) {
            U[j] &= st.cmask;
            V[j] &= st.cmask;
        }
    }
    if constexpr (STALE) {
        // the stale value of a lane: the luma pass's row (U half; the load was dropped where the lane is chroma, inside the
        // cone's shadow or outside the cone) and last step's O of this lane (V half; selector zero where the lane is
        // chroma).  Where it is not zero the costs are (the lines read zero right of the region): one of the two.
        // (V's last row, MASKED: a chroma lane finds last step's O in the U half of its own A -- st.amask left it there, in
        // the lanes that published to its ghosts as well; a lane that re-smooths has it in the park, refreshed through the mailbox)
        unsigned pw[PXL];
        if constexpr (PARK) {
            const uint4 p0 = cx.opark[(BUF * 2 + 0) * 64 + cx.lane], p1 = cx.opark[(BUF * 2 + 1) * 64 + cx.lane];
            pw[0] = p0.x; pw[1] = p0.y; pw[2] = p0.z; pw[3] = p0.w;
            pw[4] = p1.x; pw[5] = p1.y; pw[6] = p1.z; pw[7] = p1.w;
            if constexpr (MASKED) {
                const bool use_a = st.from_a && cx.role.inside_mask != 0;
#pragma unroll
                for (int j = 0; j < PXL; ++j) pw[j] = use_a ? A[j] : pw[j];
            }
        } else {
#pragma unroll
            for (int j = 0; j < PXL; ++j) pw[j] = A[j];
        }
#pragma unroll
        for (int j = 0; j < PXL; ++j) U[j] += __builtin_amdgcn_perm(pw[j], j < 4 ? ld.x : ld.y, st.sel[j & 3]);
    }
#pragma unroll
    for (int j = 0; j < PXL; ++j) S[j] = add3(A[j], U[j], V[j]);
    box7<RC>(S, Bx, cx.role);  // RC: this wave holds the pool's last column (the select of the right-hand clamp)
    unsigned O[PXL];
#pragma unroll
    for (int j = 0; j < PXL; ++j) {
        const unsigned key = and_or(Bx[j], cx.role.key_mask, rank_of<BUF, 0>());
        O[j] = pk_lshr4(key);
        if constexpr (MASKED) {
            O[j] &= st.omask;
            A[j] = add3(O[j], U[j] & st.amask, V[j]);
        } else {
            A[j] = add3(O[j], U[j], V[j]);
        }
        kmin[j] = pk_min(kmin[j], key);
    }
    if constexpr (STALE) {
        if constexpr (PARK) {  // (class R runs this code only in its masked steps, without a park)
            cx.opark[(BUF * 2 + 0) * 64 + cx.lane] = make_uint4(O[0], O[1], O[2], O[3]);
            cx.opark[(BUF * 2 + 1) * 64 + cx.lane] = make_uint4(O[4], O[5], O[6], O[7]);
        }
    }
}

__device__ __forceinline__ u32x2 issue_stale(const Ctx& cx, int b, int row, int voff)
{
    return __builtin_amdgcn_raw_buffer_load_b64(cx.rin, voff, b * cx.buf_stride + row * cx.row_stride, 0);
}

// The nine buffers and stage 3 of one step of a region wave; returns the interpolated bytes of both passes.
template <bool STALE, bool MASKED, bool PARK, bool RC>
__device__ __forceinline__ Out region_row(unsigned (&A)[kBuffers][PXL], const WideLine& n, const WideLine& nn, const Ctx& cx, const Step& st, unsigned thr_key,
                                          u32x2 (&ahead)[kBuffers])
{
    unsigned kmin[PXL];
#pragma unroll
    for (int j = 0; j < PXL; ++j) kmin[j] = thr_key;
    // STALE: `ahead` holds the luma pass's row s + 1, fetched a whole step ago (a trip to HBM is longer than two buffer steps);
    // each buffer's registers are refilled with row s + 2 as soon as the buffer step has read them
    auto run = [&](auto buf) {
        constexpr int B = decltype(buf)::value;
        const u32x2 ld = ahead[B];
        if constexpr (STALE) {
            __builtin_amdgcn_sched_barrier(0);  // (the refill must not be hoisted above the read of the same registers' previous content)
            ahead[B] = issue_stale(cx, B, st.s + 2, st.vin2);
        }
        region_buffer_step<B, STALE, MASKED, PARK, RC>(A[B], kmin, n, nn, cx, st, ld);
        if constexpr (!STALE) __builtin_amdgcn_sched_barrier(0);
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
    RawLine c, nr;
    unpark_raw_at(cx.lines + st.slot_c * 5 * 64, 64, cx.lane, c);
    unpark_raw_at(cx.lines + st.slot_n * 5 * 64, 64, cx.lane, nr);
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const unsigned t01 = __builtin_amdgcn_perm(kmin[4 * g + 1], kmin[4 * g + 0], 0x06020400u);
        const unsigned t23 = __builtin_amdgcn_perm(kmin[4 * g + 3], kmin[4 * g + 2], 0x06020400u);
        const unsigned lo = __builtin_amdgcn_perm(t23, t01, 0x05040100u) & 0x0f0f0f0fu;
        const unsigned hi = __builtin_amdgcn_perm(t23, t01, 0x07060302u) & 0x0f0f0f0fu;
        o.lo[g] = interpolate4(c, nr, 0, g, lo);
        o.hi[g] = interpolate4(c, nr, 1, g, hi);
    }
    return o;
}

// One cost buffer of one step in a wave right of the region (class S): D = the luma pass's row | last step's O << 16.
template <int BUF, bool RC>
__device__ __forceinline__ void stale_buffer_step(unsigned (&A)[PXL], unsigned (&Oprev)[PXL], const LaneRole& role, const u32x2& ld, unsigned omask,
                                                  unsigned sel)
{
    unsigned D[PXL], S[PXL], Bx[PXL];
#pragma unroll
    for (int j = 0; j < PXL; ++j) {
        D[j] = __builtin_amdgcn_perm(Oprev[j], j < 4 ? ld.x : ld.y, sel + (unsigned)(j & 3));  // sel: 0x0c040c00, or 0x0c0c0c00 (no U row behind)
        S[j] = A[j] + D[j];
    }
    box7<RC>(S, Bx, role);
#pragma unroll
    for (int j = 0; j < PXL; ++j) {
        const unsigned O = pk_lshr4(Bx[j]) & omask;  // (sum / 16) wraps to uint8_t, SangNom2.cpp:152
        A[j] = O + D[j];
        Oprev[j] = O;
    }
}

// The sweep of one wave of class CLS.  Each class is a function of its own (sweep_entry, not inlined into the kernel): one
// function holding all three made the register allocator spill a line of the plain steps to scratch, and a scratch reload
// waits for every load issued before it -- the prefetched lines, the luma pass's rows -- i.e. for HBM, in every row.
template <int NW, int CLS, bool RC>
__device__ __forceinline__ void sweep(const Args& a)
{
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int f = (int)blockIdx.x;
    Ctx cx;
    cx.tid = (int)threadIdx.x;
    cx.wave = __builtin_amdgcn_readfirstlane(cx.tid >> 6);
    cx.lane = cx.tid & 63;
    const int wave = cx.wave, lane = cx.lane;

    // the strip of this wave and the role of this lane in it (both halves alike)
    int gl;
    bool ghost;
    if (wave == 0) {
        gl = lane;
        ghost = NW > 1 && lane >= 64 - GH;
    } else {
        gl = kFirst + kInner * (wave - 1) + (lane - GH);
        ghost = lane < GH || (lane >= 64 - GH && wave < NW - 1);
    }
    const bool live = gl < a.nl;
    const bool real = live && !ghost;
    const int x0 = gl * PXL;
    const bool chroma = live && x0 < a.region_w;   // the lane's columns belong to the chroma planes
    const bool stale = live && !chroma;            // ... hold what the previous pass left
    cx.role.first_mask = live && gl == 0 ? kAll : 0u;
    cx.role.last_mask = live && gl == a.nl - 1 ? kAll : 0u;
    cx.role.line_last_mask = chroma && x0 + PXL == a.region_w ? kAll : 0u;
    cx.role.inside_mask = chroma ? kAll : 0u;
    cx.role.key_mask = 0x0ff00ff0u;
    cx.role.edge_wave = __builtin_amdgcn_readfirstlane(__any((int)(cx.role.first_mask | cx.role.last_mask | cx.role.line_last_mask)) ? 1 : 0) != 0;

    for (int p = 0; p < 2; ++p) {
        cx.rs[p] = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(a.src[p] + (int64_t)f * a.src_frame_stride[p]), 0, a.src_bytes[p], 0x00020000);
        cx.rd[p] = __builtin_amdgcn_make_buffer_rsrc(a.dst[p] + (int64_t)f * a.dst_frame_stride[p], 0, a.dst_bytes[p], 0x00020000);
    }
    cx.vload = chroma ? (x0 > 0 ? x0 - 4 : 0) : kOutOfRange;
    const int vstore = real && chroma ? x0 : kOutOfRange;

    // the luma sweep's pool: a row is [2 chunk kinds][pool_threads][8 bytes], strip v of the luma sweep's NWl waves lies in
    // kind v / NWl, wave v % NWl (sn_fused_u8_v3.hip, PoolIO); ghost lanes read the chunk of the lane that owns the columns
    cx.row_stride = a.pool_threads * 16;
    cx.buf_stride = a.pool_rows * cx.row_stride;
    cx.rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(a.pool_in + (int64_t)f * a.pool_frame_stride), 0, kBuffers * cx.buf_stride, 0x00020000);
    int v_pool;
    {
        int os = wave, ol = lane;
        if (wave > 0 && lane < GH) {
            os = wave - 1;
            ol = 64 - 2 * GH + lane;
        } else if (ghost && lane >= 64 - GH) {
            os = wave + 1;
            ol = GH + lane - (64 - GH);
        }
        const int nwl = a.pool_threads / 64, kind = os >= nwl ? 1 : 0;
        v_pool = live ? kind * (a.pool_threads * 8) + ((os - kind * nwl) * 64 + ol) * 8 : kOutOfRange;
    }

    // LDS
    uint4* const lds4 = reinterpret_cast<uint4*>(lds_raw);
    cx.lines = lds4 + (wave < a.nreg ? wave : 0) * 15 * 64;
    cx.opark = CLS == kRS ? lds4 + a.nreg * 15 * 64 + (wave - (a.nreg - a.nrs)) * 2 * kBuffers * 64 : nullptr;
    cx.mb.a = reinterpret_cast<unsigned*>(lds4 + a.nreg * 15 * 64 + a.nrs * 2 * kBuffers * 64);
    cx.mb.o = cx.mb.a + mailbox_words(NW);
    cx.mb.nw = NW;
    cx.pub_right = lane >= 64 - 2 * GH && lane < 64 - GH && wave < NW - 1;
    cx.pub_left = lane >= GH && lane < 2 * GH && wave > 0;
    cx.recv_left = lane < GH && wave > 0;
    cx.recv_right = lane >= 64 - GH && wave < NW - 1;
    cx.ghost_mask = ghost && live ? kAll : 0u;
    cx.slot = lane < GH ? lane : lane >= 64 - GH ? lane - (64 - GH) : lane >= 64 - 2 * GH ? lane - (64 - 2 * GH) : lane - GH;

    const int nk = a.nk, nr = nk - 1;          // nr = nr_c: the chroma planes' interpolated lines
    const int last_step = nr + kSkew;          // U rows 1 .. nr + 1 = sweep_u, V rows 1 .. nr
    const int cone_w = a.region_w;
    // Does this lane's stale value of pool row q matter (Args::cone_* of sn_fused_v3_common.h; `extra` = 6 for what acts
    // through the U pass)?  Right of the region inside the cone; inside the region only below it.
    auto in_cone = [&](int q, int extra) -> bool {
        const int lim = cone_w + 3 * (nr - q + 2) + extra;
        const int cols = lim < a.w ? lim : a.w;
        return x0 < cols && (q > nr || x0 + PXL > cone_w);
    };
    auto luma_row = [&](int q) -> int { return (q <= a.rows_in && in_cone(q, 3 * kSkew)) ? v_pool : kOutOfRange; };

    // Seam refresh: before step s, the ghost lanes take the state their owners published after step s - 1.
    auto refresh_due = [](int s) { return s > 1 && (s - 1) % K == 0; };
    auto publish_due = [&](int s) { return s % K == 0 && s < last_step; };

    if constexpr (CLS == kS) {
        // ---------------- right of the region: re-smooth, hand O from the U half to the V half, leave when out of the cone
        unsigned A[kBuffers][PXL], Oprev[kBuffers][PXL];
        u32x2 ahead[kBuffers];  // the luma pass's row for the next step, one load per buffer in flight
        {
            const int v1 = luma_row(1), v2 = luma_row(2);
            auto init = [&](auto buf) {
                constexpr int B = decltype(buf)::value;
                const u32x2 q = issue_stale(cx, B, 1, v1);
                ahead[B] = issue_stale(cx, B, 2, v2);
#pragma unroll
                for (int j = 0; j < PXL; ++j) {
                    A[B][j] = __builtin_amdgcn_perm(0u, j < 4 ? q.x : q.y, 0x0c0c0c00u + (unsigned)(j & 3));  // A[1] = O[0] + D[1], O[0] = 0
                    Oprev[B][j] = 0u;
                }
            };
            init(std::integral_constant<int, 0>{});
            init(std::integral_constant<int, 1>{});
            init(std::integral_constant<int, 2>{});
            init(std::integral_constant<int, 3>{});
            init(std::integral_constant<int, 4>{});
            init(std::integral_constant<int, 5>{});
            init(std::integral_constant<int, 6>{});
            init(std::integral_constant<int, 7>{});
            init(std::integral_constant<int, 8>{});
        }
        const int first_col = (wave == 0 ? 0 : kFirst + kInner * (wave - 1) - GH) * PXL;  // of lane 0 (a ghost lane)
        for (int s = 1; s <= last_step; ++s) {
            // out of both passes' cones for good (they coincide at a step: row s + 1 with six extra columns, row s - 1 without)
            if (first_col >= cone_w + 3 * (nr - s) + 9) return;  // (a finished wave no longer counts at the barriers)
            if (refresh_due(s)) {
                __syncthreads();
                if (cx.recv_left || cx.recv_right) {
                    const int at = cx.mb.at(((s - 1) / K) & 1, wave, cx.recv_left ? 0 : 1, cx.slot);
                    const unsigned *fa = cx.mb.a + at, *fo = cx.mb.o + at;
#pragma unroll
                    for (int b = 0; b < kBuffers; ++b)
#pragma unroll
                        for (int j = 0; j < PXL; ++j) {
                            A[b][j] = bfi(cx.ghost_mask, fa[b * PXL + j], A[b][j]);
                            Oprev[b][j] = bfi(cx.ghost_mask, fo[b * PXL + j], Oprev[b][j]);
                        }
                }
            }
            const int vin2 = luma_row(s + 2);
            const unsigned omask = s <= kSkew ? 0x000000ffu : 0x00ff00ffu;  // the V half starts from zero: A'[hi] = D
            // V's row s - 2 takes U's row s - 1 -- unless that row does not exist (4:2:2: the pool ends with the chroma planes' last
            // row, so V's last row adds nothing, as its pass of the reference finds nothing below it: SangNom2.cpp:126-159)
            const unsigned sel = s - 1 <= a.sweep_u ? 0x0c040c00u : 0x0c0c0c00u;
            auto run = [&](auto buf) {
                constexpr int B = decltype(buf)::value;
                const u32x2 ld = ahead[B];  // row s + 1, fetched a whole step ago
                __builtin_amdgcn_sched_barrier(0);
                ahead[B] = issue_stale(cx, B, s + 2, vin2);
                stale_buffer_step<B, RC>(A[B], Oprev[B], cx.role, ld, omask, sel);
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
            if (publish_due(s) && (cx.pub_right || cx.pub_left)) {
                const int at = cx.pub_right ? cx.mb.at(((s + 1) / K) & 1, wave + 1, 0, cx.slot) : cx.mb.at(((s + 1) / K) & 1, wave - 1, 1, cx.slot);
                uint4* ta = reinterpret_cast<uint4*>(cx.mb.a + at);
                uint4* to = reinterpret_cast<uint4*>(cx.mb.o + at);
#pragma unroll
                for (int b = 0; b < kBuffers; ++b) {
                    ta[b * 2 + 0] = make_uint4(A[b][0], A[b][1], A[b][2], A[b][3]);
                    ta[b * 2 + 1] = make_uint4(A[b][4], A[b][5], A[b][6], A[b][7]);
                    to[b * 2 + 0] = make_uint4(Oprev[b][0], Oprev[b][1], Oprev[b][2], Oprev[b][3]);
                    to[b * 2 + 1] = make_uint4(Oprev[b][4], Oprev[b][5], Oprev[b][6], Oprev[b][7]);
                }
            }
        }
        return;
    } else {
    // ---------------- waves that hold chroma columns (classes R and RS)
    int src_step[2], src_line[2], dst_step[2], dst_line[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        src_step[p] = (a.dh ? 1 : 2) * a.src_pitch[p];       // kept line k -> k + 1
        src_line[p] = (a.dh ? 0 : a.offset) * a.src_pitch[p]; // kept line 0
        dst_step[p] = 2 * a.dst_pitch[p];
        dst_line[p] = a.offset * a.dst_pitch[p];              // kept line 0 in dst
    }
    auto clampk = [&](int k) { return k < 0 ? 0 : k > nk - 1 ? nk - 1 : k; };
    // kept lines ku of U and kv of V (clamped to the plane: a line that does not exist is loaded from one that does and never used)
    auto load_lines = [&](int ku, int kv) {
        Raw q;
        q.h[0] = load_half(cx.rs[0], cx.vload, src_line[0] + clampk(ku) * src_step[0]);
        q.h[1] = load_half(cx.rs[1], cx.vload, src_line[1] + clampk(kv) * src_step[1]);
        return q;
    };
    const bool first_lane = cx.role.first_mask != 0;
    auto keep = [&](int p, int row_off, const Raw& q, int voff) {  // GetFrame's field copy, SangNom2.cpp:365 / :376
        __builtin_amdgcn_raw_buffer_store_b64(own_bytes(q.h[p], first_lane), cx.rd[p], voff, row_off, 0);
    };

    WideLine L0, L1;
    unsigned A[kBuffers][PXL];
    u32x2 ahead[kBuffers] = {};  // stale steps: the luma pass's row for the next step (see region_row)
    auto fetch_ahead = [&](int q) {
        const int v = luma_row(q);
#pragma unroll
        for (int b = 0; b < kBuffers; ++b) ahead[b] = issue_stale(cx, b, q, v);
    };
    const unsigned thr_key = ((unsigned)((a.thr[0] + 1) << 4) & 0xffffu) | ((unsigned)((a.thr[1] + 1) << 4) << 16);
    Raw qn;
    {
        const Raw q0 = load_lines(0, 0), q1 = load_lines(1, 0);
        keep(0, dst_line[0], q0, vstore);
        if (a.offset == 1) keep(0, 0, q0, vstore);  // the line that cannot be interpolated, SangNom2.cpp:386-391
        keep(0, dst_line[0] + dst_step[0], q1, vstore);
        const Raw f0 = clamp_edges(q0, cx.role), f1 = clamp_edges(q1, cx.role);
        unpack(L0, f0);
        unpack(L1, f1);
        RawLine R;
        make_raw(R, f0, L0);
        park_raw_at(cx.lines + 0 * 5 * 64, 64, lane, R);  // K_U[0]: c of U's row 1
        make_raw(R, f1, L1);
        park_raw_at(cx.lines + 1 * 5 * 64, 64, lane, R);  // K_U[1]: n of U's row 1
        // A[1] = O[0] + D[1] = D[1]: the costs of the first line pair where the lane is chroma, what the luma pass left in row 1
        // where it is not (the V half is set by the first kSkew steps)
        const int v1 = CLS == kRS ? luma_row(1) : kOutOfRange;
        auto init = [&](auto buf) {
            constexpr int B = decltype(buf)::value;
            const u32x2 q = issue_stale(cx, B, 1, v1);
#pragma unroll
            for (int j = 0; j < PXL; ++j)
                A[B][j] = cost<B>(L0, L1, j) + __builtin_amdgcn_perm(0u, j < 4 ? q.x : q.y, 0x0c0c0c00u + (unsigned)(j & 3));
        };
        init(std::integral_constant<int, 0>{});
        init(std::integral_constant<int, 1>{});
        init(std::integral_constant<int, 2>{});
        init(std::integral_constant<int, 3>{});
        init(std::integral_constant<int, 4>{});
        init(std::integral_constant<int, 5>{});
        init(std::integral_constant<int, 6>{});
        init(std::integral_constant<int, 7>{});
        init(std::integral_constant<int, 8>{});
        qn = load_lines(2, 0);  // step 1 unpacks K_U[2] and K_V[0]
        fetch_ahead(2);         // ... and reads the luma pass's row 2 where it re-smooths
    }

    // One step.  n = (K_U[s], K_V[s - 2]), nn receives (K_U[s + 1], K_V[s - 1]) from the prefetched qn.
    auto step = [&](int s, WideLine& n, WideLine& nn, auto stale_tag, auto masked_tag) __attribute__((always_inline)) {
        constexpr bool STALE = decltype(stale_tag)::value;
        constexpr bool MASKED = decltype(masked_tag)::value;
        Step st;
        st.s = s;
        const int ku = s + 1, kv = s + 1 - kSkew;  // the kept lines this step takes in
        {
            const Raw fq = clamp_edges(qn, cx.role);
            unpack(nn, fq);
            RawLine R;
            make_raw(R, fq, nn);
            park_raw_at(cx.lines + ((s + 1) % 3) * 5 * 64, 64, lane, R);
            if constexpr (MASKED) {
                keep(0, dst_line[0] + ku * dst_step[0], qn, ku <= nk - 1 ? vstore : kOutOfRange);
                keep(1, dst_line[1] + kv * dst_step[1], qn, kv >= 0 && kv <= nk - 1 ? vstore : kOutOfRange);
                if (a.offset == 1 && kv == 0) keep(1, 0, qn, vstore);
            } else {
                keep(0, dst_line[0] + ku * dst_step[0], qn, vstore);
                keep(1, dst_line[1] + kv * dst_step[1], qn, vstore);
            }
        }
        const Raw qnext = load_lines(ku + 1, kv + 1);
        if (refresh_due(s)) {
            __syncthreads();
            if (cx.recv_left || cx.recv_right) {
                const int at = cx.mb.at(((s - 1) / K) & 1, wave, cx.recv_left ? 0 : 1, cx.slot);
                const unsigned* fa = cx.mb.a + at;
#pragma unroll
                for (int b = 0; b < kBuffers; ++b)
#pragma unroll
                    for (int j = 0; j < PXL; ++j) A[b][j] = bfi(cx.ghost_mask, fa[b * PXL + j], A[b][j]);
                if constexpr (CLS == kRS) {  // last step's O of the ghost lanes, for the lanes among them that re-smooth
                    const uint4* fo = reinterpret_cast<const uint4*>(cx.mb.o + at);
#pragma unroll
                    for (int k = 0; k < 2 * kBuffers; ++k) cx.opark[k * 64 + lane] = fo[k];
                }
            }
        }
        st.slot_c = (s - 1) % 3;
        st.slot_n = s % 3;
        st.vin2 = kOutOfRange;
        st.cmask = st.omask = st.amask = kAll;
        st.from_a = false;
        if constexpr (STALE) {
            st.vin2 = luma_row(s + 2);
            // the V half takes last step's O where the lane re-smooths; in the V pass's last row (row nr: its next costs do not
            // exist, SangNom2.cpp:74-124 writes rows 1 .. nr only) every lane does
            // (... if U has that row: in a 4:2:2 clip the pool ends with the chroma planes' last row and V's last row adds nothing)
            const bool prev = MASKED ? (s - kSkew == nr ? a.sweep_u > nr : stale) : stale;
            const unsigned base = prev ? 0x0c040c00u : 0x0c0c0c00u;
#pragma unroll
            for (int k = 0; k < 4; ++k) st.sel[k] = base + (unsigned)k;
        }
        if constexpr (MASKED) {
            const int ru = s, rv = s - kSkew;
            st.cmask = (ru <= nr - 1 ? kLo : 0u) | (rv >= 0 && rv <= nr - 1 ? kHi : 0u);
            st.omask = kLo | (rv >= 1 ? kHi : 0u);
            st.amask = ru == nr + 1 ? kHi : kAll;  // U's last row: its half of A keeps O alone, which is what V's last row needs
            st.from_a = rv == nr;
        }
        const Out o = region_row<STALE, MASKED, CLS == kRS, RC>(A, n, nn, cx, st, thr_key, ahead);
        {
            const int ru = s, rv = s - kSkew;
            u32x2 lo, hi;
            lo.x = o.lo[0]; lo.y = o.lo[1];
            hi.x = o.hi[0]; hi.y = o.hi[1];
            const int vlo = !MASKED || (ru >= 1 && ru <= nr) ? vstore : kOutOfRange;
            const int vhi = !MASKED || (rv >= 1 && rv <= nr) ? vstore : kOutOfRange;
            __builtin_amdgcn_raw_buffer_store_b64(lo, cx.rd[0], vlo, dst_line[0] + a.dst_pitch[0] + (ru - 1) * dst_step[0], 0);
            __builtin_amdgcn_raw_buffer_store_b64(hi, cx.rd[1], vhi, dst_line[1] + a.dst_pitch[1] + (rv - 1) * dst_step[1], 0);
        }
        if (publish_due(s) && (cx.pub_right || cx.pub_left)) {
            const int at = cx.pub_right ? cx.mb.at(((s + 1) / K) & 1, wave + 1, 0, cx.slot) : cx.mb.at(((s + 1) / K) & 1, wave - 1, 1, cx.slot);
            uint4* ta = reinterpret_cast<uint4*>(cx.mb.a + at);
#pragma unroll
            for (int b = 0; b < kBuffers; ++b) {
                ta[b * 2 + 0] = make_uint4(A[b][0], A[b][1], A[b][2], A[b][3]);
                ta[b * 2 + 1] = make_uint4(A[b][4], A[b][5], A[b][6], A[b][7]);
            }
            if constexpr (CLS == kRS) {
                uint4* to = reinterpret_cast<uint4*>(cx.mb.o + at);
#pragma unroll
                for (int k = 0; k < 2 * kBuffers; ++k) to[k] = cx.opark[k * 64 + lane];
            }
        }
        qn = qnext;
    };
    using T = std::integral_constant<bool, true>;
    using F = std::integral_constant<bool, false>;

    // steps 1 .. kSkew (V not started) and nr .. nr + kSkew (the rows without a following line pair) masked, the rest plain;
    // ONE call site per variant of the step (each is a few thousand instructions)
    const int plain_from = kSkew + 1, plain_to = nr;  // [plain_from, plain_to)
    for (int s = 1; s <= last_step; ++s) {
        if (s == plain_from && plain_from < plain_to) {
            if constexpr (CLS == kR) {
                for (; s + 1 < plain_to; s += 2) {
                    step(s, L1, L0, F{}, F{});
                    step(s + 1, L0, L1, F{}, F{});
                }
                if (s < plain_to) {
                    step(s, L1, L0, F{}, F{});
                    L1 = L0;
                    ++s;
                }
                fetch_ahead(plain_to + 1);  // (the plain steps fetch nothing; the masked steps that follow read a row ahead)
            } else {
                for (; s < plain_to; ++s) {
                    step(s, L1, L0, T{}, F{});
                    L1 = L0;
                }
            }
        }
        step(s, L1, L0, T{}, T{});
        L1 = L0;
    }

    // dst row h-1 := K[nk-1] when the top field is kept, SangNom2.cpp:380-385
    if (a.offset == 0) {
        const Raw q = load_lines(nk - 1, nk - 1);
        keep(0, (2 * nk - 1) * a.dst_pitch[0], q, vstore);
        keep(1, (2 * nk - 1) * a.dst_pitch[1], q, vstore);
    }
    }  // region classes
}

// (an argument of a device function travels in vector registers and the compiler must take it for divergent: every word of
// the argument block goes through v_readfirstlane once, so that in the sweep rows, pitches and pointers are scalars again --
// as they are in a kernel that reads its own argument segment)
template <int NW, int CLS, bool RC>
__device__ __attribute__((noinline)) void sweep_entry(const Args* from)
{
    static_assert(sizeof(Args) % 4 == 0, "Args is copied word by word");
    constexpr int kWords = (int)(sizeof(Args) / 4);
    const uint32_t* in = reinterpret_cast<const uint32_t*>(from);
    uint32_t words[kWords];
#pragma unroll
    for (int i = 0; i < kWords; ++i) words[i] = (uint32_t)__builtin_amdgcn_readfirstlane((int)in[i]);
    Args a;
    __builtin_memcpy(&a, words, sizeof a);
    sweep<NW, CLS, RC>(a);
}

template <int NW>
__global__ void __launch_bounds__(NW * 64, 2) k_fused_u8_uv(Args a)
{
    // the class of this wave, from its lanes' columns (ghost lanes included); the rest of the geometry is derived in sweep()
    const int tid = (int)threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int gl = wave == 0 ? lane : kFirst + kInner * (wave - 1) + (lane - GH);
    const bool live = gl < a.nl, chroma = live && gl * PXL < a.region_w, stale = live && !chroma;
    const int cls = __builtin_amdgcn_readfirstlane(__any((int)chroma) ? (__any((int)stale) ? (int)kRS : (int)kR) : (int)kS);
    // ... and whether it holds the pool's last column: only there the box needs its right-hand clamp (a wave-uniform choice of
    // the whole function, never a branch inside a buffer step)
    const bool rc = __builtin_amdgcn_readfirstlane(__any((int)(live && gl == a.nl - 1)) ? 1 : 0) != 0;
    if (cls == kS) {
        if (rc) sweep_entry<NW, kS, true>(&a);
        else sweep_entry<NW, kS, false>(&a);
    } else if (cls == kR) {
        sweep_entry<NW, kR, false>(&a);  // (the last column is never chroma: region_w < w)
    } else {
        if (rc) sweep_entry<NW, kRS, true>(&a);
        else sweep_entry<NW, kRS, false>(&a);
    }
}

}  // namespace uv

// Which subsampled geometries the one-sweep chroma passes take; everything else keeps the two chroma sweeps of
// sn_fused_u8_v3.hip.  Both chroma planes processed and alike, the region a whole number of lanes, enough rows for the skew.
// 4:2:0: the U pass has an extra row (nr_c + 1) and the V pass's last row finds it; 4:2:2: the pool ends with the chroma
// planes' last row, the U pass has no extra row and V's last row takes nothing from it.
bool fused_uv_ok(int sweep_w, int region_w, int nk_c, int bh)
{
    if (sweep_w % 32 != 0 || region_w % 8 != 0 || region_w <= 0 || region_w >= sweep_w) return false;
    if (v3c::strips_for(sweep_w / v3c::PXL) > uv::kMaxWaves) return false;
    const int nr = nk_c - 1;
    return nr >= 2 * uv::kSkew + 2 && nr <= bh - 1;
}

hipError_t launch_fused_u8_uv(hipStream_t st, const PlaneArgs& pu, const PlaneArgs& pv, double thr_u, double thr_v, int nframes, const FusedPool& pool)
{
    using namespace v3c;
    uv::Args a{};
    const PlaneArgs* pp[2] = {&pu, &pv};
    for (int p = 0; p < 2; ++p) {
        a.src[p] = pp[p]->src;
        a.dst[p] = pp[p]->dst;
        a.src_frame_stride[p] = pp[p]->src_frame_stride;
        a.dst_frame_stride[p] = pp[p]->dst_frame_stride;
        a.src_pitch[p] = pp[p]->src_pitch;
        a.dst_pitch[p] = pp[p]->dst_pitch;
        a.src_bytes[p] = (int)((int64_t)pp[p]->src_pitch * pp[p]->h_in);
        a.dst_bytes[p] = (int)((int64_t)pp[p]->dst_pitch * pp[p]->h_out);
    }
    if (pu.w != pv.w || pu.h_out != pv.h_out || pu.h_in != pv.h_in || pu.offset != pv.offset || pu.dh != pv.dh) return hipErrorInvalidValue;
    a.w = pool.sweep_w;
    a.region_w = pu.w;
    a.nk = pu.h_out / 2;
    a.offset = pu.offset;
    a.dh = pu.dh;
    a.thr[0] = (int)thr_u;
    a.thr[1] = (int)thr_v;
    a.nl = a.w / PXL;
    const int nw = strips_for(a.nl);
    // classes of the waves, as the kernel derives them from its lanes (ghost lanes included)
    int nreg = 0, nrs = 0;
    for (int w = 0; w < nw; ++w) {
        bool chroma = false, stale = false;
        for (int lane = 0; lane < 64; ++lane) {
            const int gl = w == 0 ? lane : kFirst + kInner * (w - 1) + (lane - GH);
            if (gl >= a.nl) continue;
            (gl * PXL < a.region_w ? chroma : stale) = true;
        }
        if (chroma) {
            if (w != nreg) return hipErrorInvalidValue;  // (the region is a prefix of the strips)
            ++nreg;
            if (stale) ++nrs;
        }
    }
    if (nreg < 1 || nrs > 2) return hipErrorInvalidValue;
    a.nreg = nreg;
    a.nrs = nrs;
    a.pool_in = pool.pool_in;
    a.pool_frame_stride = pool.frame_stride;
    a.pool_rows = pool.pool_rows;
    a.rows_in = pool.rows_in;
    a.pool_threads = fused_v3_waves(pool.sweep_w) * 64;
    a.sweep_u = pool.sweep_rows;
    a.nframes = nframes;
    if (a.sweep_u != a.nk && a.sweep_u != a.nk - 1) return hipErrorInvalidValue;  // nr_c + 1, or nr_c where the pool has no row below (4:2:2)
    const int lds = uv::lds_bytes(nw, nreg, nrs);
    hipError_t e = hipSuccess;
#define SN_LAUNCH(NW)                                                                                                  \
    case NW:                                                                                                           \
        if (lds > 64 * 1024) e = hipFuncSetAttribute((const void*)uv::k_fused_u8_uv<NW>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
        if (e == hipSuccess) hipLaunchKernelGGL((uv::k_fused_u8_uv<NW>), dim3(nframes), dim3(NW * 64), lds, st, a);    \
        break;
    switch (nw) {
        SN_LAUNCH(1) SN_LAUNCH(2) SN_LAUNCH(3) SN_LAUNCH(4) SN_LAUNCH(5) SN_LAUNCH(6) SN_LAUNCH(7) SN_LAUNCH(8)
    default: return hipErrorInvalidValue;
    }
#undef SN_LAUNCH
    if (e != hipSuccess) return e;
    return hipGetLastError();
}

}  // namespace sn
