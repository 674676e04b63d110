// sn_fused_v3_common.h -- definitions shared by the fused sweep kernels (sn_fused_u8_v3.hip: two
// 8-bit strips per register; sn_fused_u16_v3.hip: one 16-bit strip per register).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "sn_internal.h"

namespace sn {
namespace v3c {

constexpr int PXL = 8;           // pixels per lane and per strip
constexpr int GH = 2;            // ghost lanes on each inner side of a strip
constexpr int K = GH * PXL / 3;  // rows between two seam refreshes (5)
constexpr int kFirst = 64 - GH;      // real lanes of strip 0
constexpr int kInner = 64 - 2 * GH;  // real lanes of every later strip
constexpr int kOutOfRange = 0x7fffffff;  // voffset that the buffer range check always rejects

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

struct Args {
    const uint8_t* src;
    uint8_t* dst;
    int64_t src_frame_stride;
    int64_t dst_frame_stride;
    int32_t src_pitch;
    int32_t dst_pitch;
    int32_t w;
    int32_t nk;      // kept lines
    int32_t offset;  // first kept line in dst
    int32_t dh;
    int32_t thr;
    int32_t nl;      // real lanes = w / 8
    int32_t nvw;     // virtual wavefronts
    int32_t nw;      // physical waves
    int32_t src_bytes;  // bytes of one source plane (buffer descriptor range)
    int32_t dst_bytes;  // bytes of one destination plane
    // Dependency cone of the pool coupling.  A hand-off cell (pool row q, column x) can reach a chroma output only if
    // x < cone_w + 3 * (cone_nr - q + 2) (+6 for the luma -> U hand-off, whose cells act through U's sweep): stage 2
    // spreads 3 columns per row and the chroma region ends at column cone_w, row cone_nr.  Inside the region
    // (x < cone_w) a cell matters only below it (q > cone_nr).  Lanes whose columns lie outside neither store nor
    // load their slot (out-of-range voffset: no branch, no HBM traffic).
    int32_t cone_w, cone_nr;   // chroma width, chroma nr = interpolated lines
    int32_t cone_in, cone_out; // extra columns of the loads (U: 6, V: 0) / of the stores (luma: 6, U: 0)
    // pool coupling for subsampled chroma (modes kLumaSpill / kChroma, see below)
    const uint8_t* pool_in;   // smoothed buffers left by the previous pass (kChroma)
    uint8_t* pool_out;        // where this pass leaves its smoothed buffers (kLumaSpill, first kChroma pass)
    int64_t pool_frame_stride;
    int32_t pool_rows;        // rows a pool buffer holds (row index 1 .. pool_rows - 1 used)
    int32_t pool_row_bytes;   // 0: pool_out has the sweeps' own layout (a slot per thread); > 0 (kLumaSpill): it is a pool
                              // of the pool path, [buffer][row][column] samples with this row pitch (sn_pool_kernels.hip)
    int32_t rows_in;          // rows 1 .. rows_in of pool_in are valid, later rows read as zero
    int32_t rows_out;         // rows 1 .. rows_out are written to pool_out (0 = none)
    int32_t region_w;         // kChroma: columns < region_w belong to the chroma plane
    int32_t sweep_rows;       // kChroma: pool rows to sweep (>= nk - 1)
    int32_t turn_shift;       // log2 of the priority time slice in 100 MHz ticks (TurnTaking)
    int32_t nframes;          // frames of this launch (the last workgroup may hold fewer than group_of(nw))
    // row bands (kernel template parameter BAND): blockIdx.y = band; see below
    int32_t band_rows;        // pool rows per band
    int32_t band_warm;        // rows a band sweeps before its first own row, starting from a zero guess of the state
    int32_t nbands;
    uint32_t* band_state;     // [frame][band][warm, end][kBuffers * PXL][threads of the plane] state words, ghosts zeroed
    int32_t* band_flags;      // [frame]: set by the verification that follows (sn_band.hip)
    int32_t band_reset;       // this sweep clears band_flags first (the first plane of a frame)
};

// The reference's nine buffers are sized for the luma plane and shared by all planes, so a
// subsampled chroma pass smooths a pool that still holds the previous pass's results outside the
// chroma region.  Exact emulation in the fused kernel:
//   kLumaSpill  the luma sweep also leaves its smoothed values O of the rows the chroma passes can
//               reach in a scratch pool;
//   kChroma     the sweep runs over the whole luma-wide pool: inside the chroma region the cost of
//               the next row comes from the chroma lines (stage 1), elsewhere it is the previous
//               pass's O read back from the pool; stage 3 and the output exist only inside the region.
// Pool layout: [buffer][row][thread][4 dwords], dword k = O[2k] | O[2k+1] << 8 (packed pairs), i.e. every
// thread re-reads what the thread with the same columns wrote; ghost lanes read their owner's slot.
//   kPadded     a plane narrower than its pool stride on a zero-filled pool (sn_config.fresh_pool): the sweep covers
//               the whole stride, costs are zero in the padding columns, nothing is read back or left behind.
//   kChromaLast the last chroma sweep of a frame: kChroma that hands nothing on (8-bit sweep only: no packing of stores
//               that would all be dropped)
// Row bands (BAND, every mode): the plane is cut into bands of rows, one workgroup per band, so that ONE frame fills the
// device (the latency path: a synchronous GetFrame, a short look-ahead).  Stage 2 is a recurrence from the top of the
// plane, so a band cannot know its starting state; but the recurrence forgets (each row keeps 7/16 of the previous
// one), so the band starts `band_warm` rows early from a zero state and has, on ordinary content, the exact state when
// it reaches its own rows.  "Ordinary" is not "always" (a rounding difference of one can live on for ever on flat or
// periodic content), so every band leaves the state it reached its first row with and the state it ends with;
// sn_band.hip compares each band's end with the next band's start -- equal everywhere means, by induction from band 0,
// that every band computed what the top-to-bottom sweep computes -- and a frame that fails is redone by the pool path
// (guarded launches that otherwise exit at once). 
// Planes on their own are cut (kPlain), and the luma sweep that leaves its rows in a pool of the POOL PATH
// (kLumaSpill with pool_row_bytes): a single 4:2:0 frame takes its luma plane through the bands and its chroma planes
// through the pool kernels, which find in the pool what the reference's luma pass would have left there.  The pool-coupled sweeps were tried and work mechanically -- a band
// starts from the hand-off row of its first row and hands on its own rows only -- but the last chroma sweep never passes
// the check: outside the chroma region it re-smooths what two passes have smoothed already, data so even that the
// rounding difference between the run-up and the true history does not die out (256 x 400 noise needs a run-up of 128
// rows, 3840 x 2160 is still wrong after 128), so every such frame would be done twice.
enum Mode { kPlain = 0, kLumaSpill = 1, kChroma = 2, kPadded = 3, kChromaLast = 4 };
__host__ __device__ constexpr bool chroma_mode(int mode) { return mode == kChroma || mode == kChromaLast; }
__host__ __device__ constexpr bool has_region(int mode) { return chroma_mode(mode) || mode == kPadded; }  // lines narrower than the sweep
__host__ __device__ constexpr bool has_pools(int mode) { return mode == kLumaSpill || chroma_mode(mode); }


typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned dpp_from_left(unsigned v)
{
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x138 /* wave_shr:1 */, 0xf, 0xf, true);
}
// ... where lane 0, which has no left neighbour, gets `edge` instead
__device__ __forceinline__ unsigned dpp_from_left_or(unsigned edge, unsigned v)
{
    return (unsigned)__builtin_amdgcn_update_dpp((int)edge, (int)v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}
__device__ __forceinline__ unsigned dpp_from_right(unsigned v)
{
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x130 /* wave_shl:1 */, 0xf, 0xf, true);
}
__device__ __forceinline__ unsigned pk_max(unsigned a, unsigned b)
{
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
}
__device__ __forceinline__ unsigned pk_min(unsigned a, unsigned b)
{
    return __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
}
// Instruction selection the compiler cannot be talked into (it splits vector shifts into per-half selects, and its
// demanded-bits analysis undoes a shared and-or): spelled out.  Plain `asm` (not volatile), so they are scheduled and
// CSE'd like any other pure operation.
// bit BIT of each 16-bit half spread over that half
template <int BIT>
__device__ __forceinline__ unsigned pk_bit_mask(unsigned v)
{
    unsigned m;
    asm("v_pk_lshlrev_b16 %0, %2, %1 op_sel_hi:[0,1]\n\tv_pk_ashrrev_i16 %0, 15, %0 op_sel_hi:[0,1]" : "=v"(m) : "v"(v), "n"(15 - BIT));
    return m;
}
// (s + 1) >> 1 in both halves (s < 65535)
__device__ __forceinline__ unsigned pk_avg_from_sum(unsigned s)
{
    unsigned r;
    asm("v_pk_add_u16 %0, %1, 1 op_sel_hi:[1,0]\n\tv_pk_lshrrev_b16 %0, 1, %0 op_sel_hi:[0,1]" : "=v"(r) : "v"(s));
    return r;
}
// both halves shifted right by 4 (nothing crosses from the high half into the low one)
__device__ __forceinline__ unsigned pk_lshr4(unsigned v)
{
    const u16x2 four = {4, 4};
    return __builtin_bit_cast(unsigned, (u16x2)(__builtin_bit_cast(u16x2, v) >> four));  // v_pk_lshrrev_b16
}
// both halves shifted right by 8
__device__ __forceinline__ unsigned pk_lshr8(unsigned v)
{
    const u16x2 eight = {8, 8};
    return __builtin_bit_cast(unsigned, (u16x2)(__builtin_bit_cast(u16x2, v) >> eight));  // v_pk_lshrrev_b16
}
// max(a - b, 0) in both halves: v_pk_sub_u16 with clamp
__device__ __forceinline__ unsigned pk_sub_sat(unsigned a, unsigned b)
{
    unsigned r;
    asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// a + b + c as ONE instruction (left to itself the compiler forms b + c first where two such sums share it)
__device__ __forceinline__ unsigned add3(unsigned a, unsigned b, unsigned c)
{
    unsigned r;
    asm("v_add3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// (a & m) | c with a wave-uniform c (the one scalar operand the encoding allows)
__device__ __forceinline__ unsigned and_or(unsigned a, unsigned m, unsigned c)
{
    unsigned r;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(m), "s"(c));
    return r;
}
// ... with a per-lane c
__device__ __forceinline__ unsigned and_or_v(unsigned a, unsigned m, unsigned c)
{
    unsigned r;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(m), "v"(c));
    return r;
}
// |a - b| in both halves (values < 32768 per half)
__device__ __forceinline__ unsigned pk_absdiff(unsigned a, unsigned b) { return pk_max(a, b) - pk_min(a, b); }
// (m & x) | (~m & y): v_bfi_b32
__device__ __forceinline__ unsigned bfi(unsigned m, unsigned x, unsigned y) { return (m & x) | (~m & y); }

// Two workgroups share every SIMD (one wave each), and the issue arbiter serves equal priorities oldest first:
// left alone, the wave dispatched first runs at nearly its solo speed and the other one on the leftovers, so the
// first finishes a 2160p sweep after 3.6 ms and the second only after 5.7 ms -- the last 2.1 ms with one wave per
// SIMD (measured with per-wave timestamps; MI355X_MICROARCH.md, "Two waves per SIMD", item 2).  Taking turns at
// s_setprio 1 in slices of the free-running 100 MHz clock, the wave in the odd hardware slot during odd slices
// and vice versa, makes both advance at the same average rate and finish together.  Slices of about a quarter
// of a sweep measured best (short ones cost throughput); the slice length comes from the host.
struct TurnTaking {
    unsigned slot_parity;
    int shift;  // > 0: time slices of 2^shift ticks of the 100 MHz clock; kLadder: by rows since the last seam barrier; 0: off
    static constexpr int kLadder = -1;
    __device__ __forceinline__ void init(int turn_shift)
    {
        slot_parity = __builtin_amdgcn_s_getreg((3 << 11) | 4) & 1u;  // HW_REG_HW_ID, wave_id bit 0
        shift = turn_shift;
    }
    // once per row: a few scalar instructions.  `block_row`: rows this wave has done since the last seam barrier (0 .. K - 1).
    __device__ __forceinline__ void update(int block_row) const
    {
        if (shift == 0) return;
        if (shift == kLadder) {
            // Workgroups of eight waves: the two waves a SIMD holds belong to the SAME workgroup and meet at its seam barrier.
            // The priority FALLS with the rows done since that barrier -- 3, 3, 2, 1, 0 -- so whichever of the two is behind
            // has the higher one and they reach the next barrier together.  No communication: the barrier itself is the
            // common clock (see turn_shift_for).
            switch (block_row) {
            case 0:
            case 1: __builtin_amdgcn_s_setprio(3); break;
            case 2: __builtin_amdgcn_s_setprio(2); break;
            case 3: __builtin_amdgcn_s_setprio(1); break;
            default: __builtin_amdgcn_s_setprio(0); break;
            }
            return;
        }
        const unsigned turn = (unsigned)(__builtin_amdgcn_s_memrealtime() >> shift) & 1u;
        if (turn != slot_parity) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
    }
};

// Frames per workgroup.  A plane that needs only one or two waves shares its workgroup with other frames' planes so
// that every workgroup has four waves, one per SIMD (TurnTaking below relies on that shape).  The frames of a
// workgroup are independent: each has its own slice of the dynamic LDS and they only meet at the barriers.
__host__ __device__ constexpr int group_of(int nw) { return nw == 1 ? 4 : nw == 2 ? 2 : 1; }

// slice = about a quarter of the time a sweep of nk kept lines takes (a row costs roughly 4.5 us) for workgroups of four
// waves: those put one wave on each SIMD, two workgroups fill a CU, and the partners on all four SIMDs are the same two
// workgroups in opposite slots.  (2-wave workgroups lost 8 % with turns: the waves of a workgroup -- tied to each other by
// the seam refresh -- would hold different priorities at the same time.)
// Workgroups of EIGHT waves (4320p 8-bit planes, 16-bit and float planes from 2160p on) hold both waves of every SIMD
// themselves, waves k and k + 4.  Round 3 left them without turns ("they cannot drift apart"); they do, inside every block
// of five rows: tools/row_timing.py (s_memtime around the phases of a row) finds a 4320p wave waiting at the seam barrier
// for 22 % of its cycles (2160p, two workgroups per CU: 4 %) -- equal priorities are served oldest first, so wave k issues
// whenever it can, reaches the barrier early and waits, and wave k + 4 then finishes the block alone at a single wave's
// issue rate.  Removing the barrier gains nothing (the kernel is as slow as its slowest wave); keeping the pair in step
// does.  Measured at 4320p Y8 (profiles/r4_ab_experiments.md 3., 7., 8.): time slices of 10 us +3.3 % (5 us +2 %, 20 us +1 %,
// 40 us and more -1 %); row numbers exchanged through LDS, the wave behind takes the priority: +2.9 %; the same by buffer
// steps: -19 % (nine LDS round trips per row); and what ships -- TurnTaking::kLadder, a priority that falls with the rows done
// since the last barrier, which needs no exchange at all because the barrier is the pair's common clock: **+8.3 %** (8-bit
// only: 2160p Y16 +0.8 % over the slices, YUV420P16 -1 %, Y32 and YUV444PS -7 %: those keep the slices of 10 us).
inline int turn_shift_for(int nk, int waves, int bytes_per_sample)
{
    // (the ladder is for the 8-bit sweep, whose rows are all but pure vector arithmetic; the float sweep, which moves three
    // buffers' state through LDS in every row, loses 7 % with it and the 16-bit sweep gains nothing over the slices)
    if (waves == 8) return bytes_per_sample == 1 ? TurnTaking::kLadder : 10;
    if (waves != 4) return 0;
    int s = 10;
    while ((128ll * nk) >> (s + 1)) ++s;
    return s;
}

// 16-byte buffer store with the whole offset in voffset and soffset = 0.  With an SGPR soffset hipcc pads no wait
// state after a >8-byte store, and gfx950 was seen still reading the last data register while the next VALU
// instruction overwrote it (stale words in the spilled rows of 3840-wide 16-bit sweeps); in this form the
// compiler's hazard recogniser adds the s_nop itself where one is needed.  `voff` may be kOutOfRange: adding
// an offset below 2^31 keeps it out of range.
__device__ __forceinline__ void store_b128(const u32x4& d, __amdgpu_buffer_rsrc_t r, int voff, int off)
{
    __builtin_amdgcn_raw_buffer_store_b128(d, r, (int)((unsigned)voff + (unsigned)off), 0, 0);
}


inline int strips_for(int nl) { return nl <= 64 ? 1 : 1 + (nl - kFirst + kInner - 1) / kInner; }

}  // namespace v3c

// sn_fused_u8_v3.hip compiled with -DSN_TU_PLAIN: the 8-bit sweeps of planes on their own (mode kPlain / kPadded)
hipError_t launch_fused_u8_v3_plain(hipStream_t st, const v3c::Args& a, int nframes, int mode);

}  // namespace sn
