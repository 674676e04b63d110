// sn_internal.h -- shared declarations of libsangnom_hip.so (not part of the public ABI).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

#include "sangnom_hip.h"

namespace sn {

constexpr int kBuffers = 9;  // TOTAL_BUFFERS, /root/reference/src/SangNom2.h:22

// One plane of one launch (all frames of a batch share it; frame f adds f * *_frame_stride).
struct PlaneArgs {
    const uint8_t* src;
    uint8_t* dst;
    int64_t src_frame_stride;  // bytes
    int64_t dst_frame_stride;  // bytes
    int32_t src_pitch;         // bytes
    int32_t dst_pitch;         // bytes
    int32_t w;                 // pixels
    int32_t h_in;              // source rows
    int32_t h_out;             // destination rows
    int32_t offset;            // first kept line (0 or 1)
    int32_t dh;
    int32_t enabled;           // processPlane[i] || dh
    const int32_t* guard;      // pool-path kernels: when set, frame f is worked on only if guard[f] != 0 (sn_band.hip)
    int32_t guard_single;      // ... 1: ONE word decides for every frame of the launch (guard[0]; the redo of a chain that timed out)
};

// Scratch pool geometry (src/SangNom2.cpp:287-288,305-310), in elements of T.
struct PoolArgs {
    uint8_t* base;             // slot 0
    int64_t slot_bytes;        // 9 * rows * stride_e * sizeof(T)
    int32_t stride_e;          // roundup(luma width, 32)
    int32_t bh;                // bufferHeight; a buffer has bh + 1 rows
    const int32_t* guard;      // as PlaneArgs::guard
    int32_t guard_single;      // as PlaneArgs::guard_single
    int32_t rows;              // stage 2 stops before this pool row (0 = bh: the whole pool, as the reference does)
    int32_t slot_step = 0;     // k_prepare / k_finalize: frame f uses slot (slot0 + f * slot_step) % slot_mod
    int32_t slot_mod = 0;      // (0 / 0: slot0 + f)
};

// Stage 2 of a history-carrying clip as a chain of passes (one per processed plane and frame, in the reference's
// order): pass j smooths slot (origin + 1 + j) % slot_mod and takes every cell its own k_prepare did not write --
// columns from w[j % pn] on, rows above nr[j % pn], row 0 -- from the slot of the pass before it.
constexpr int kChainMaxGroups = 8;  // workgroups per buffer of an 8-bit chain (k_smooth_u8_chain<true>)
struct ChainArgs {
    int32_t npass, pn;
    int32_t w[3], nr[3];
    int32_t origin;
    int32_t groups, slack;  // 8-bit: workgroups per buffer (<= 1: one) and the rounds of slack between two of them
    uint32_t* flags;        // groups > 1: kBuffers * kChainMaxGroups * 32 words, zero at launch (rounds completed per workgroup)
    uint32_t* status;       // groups > 1: device word, set when a workgroup stopped waiting for the one before it (zero at launch
                            // unless a test forces the fault: then every wait is skipped, as after a real time-out)
    const uint32_t* only_if;  // != nullptr: the launch runs only if this word is not zero (the guarded redo on one workgroup per buffer)
    int32_t rows;           // > 0: stage 2 stops there (rows 1 .. rows - 1 are smoothed), as PoolArgs::rows
    int32_t nchains, chain_step;  // nchains > 1: that many independent chains of npass passes (blockIdx.y), chain y starting at
                                  // slot origin + y * chain_step (> npass; no wrap around the ring; one workgroup per buffer)
};

struct Context;

// sn_turn.hip: quarter turn of a plane (right = clockwise), for device-resident anti-aliasing pipelines
hipError_t launch_turn(hipStream_t s, int bytes, int right, int nframes, const uint8_t* src, int64_t src_frame_stride, int src_pitch, int w,
                       int h, uint8_t* dst, int64_t dst_frame_stride, int dst_pitch);

// sn_pool_kernels.hip: the three-kernel path over the HBM-resident pool (every format).
hipError_t launch_assemble(hipStream_t s, const PlaneArgs& p, int bytes, int nframes);
hipError_t launch_pool_plane(hipStream_t s, const PlaneArgs& p, const PoolArgs& pool, int bytes,
                             double threshold, int nframes, int slot0);
int pool_chain_lanes(int bytes, int stride_e);  // passes one workgroup keeps in flight; 0: no chain for this pool
int pool_chain_groups(int bytes, int stride_e, int want);  // workgroups per buffer the chain kernel can run for `want`
hipError_t launch_pool_prepare(hipStream_t s, const PlaneArgs& p, const PoolArgs& pool, int bytes, int nframes, int slot0);
hipError_t launch_pool_finalize(hipStream_t s, const PlaneArgs& p, const PoolArgs& pool, int bytes, double threshold, int nframes,
                                int slot0);
hipError_t launch_pool_chain(hipStream_t s, const PoolArgs& pool, const ChainArgs& chain, int bytes);
// after the guarded redo of a chain launch: *redone += (*fault != 0), and the running count into *host_mirror (host memory the device writes)
hipError_t launch_chain_redo_count(hipStream_t s, const uint32_t* fault, uint32_t* redone, uint32_t* host_mirror);

// sn_fused_select.hip: which configurations the fused sweeps serve.  A fused launch also does the plane's frame
// assembly, so launch_assemble must not be called for a plane it serves.
bool fused_eligible(const sn_config& c);
// one plane of width w on its own (plain sweep): sample size and width within what the fused kernels take
bool fused_plane_eligible(int bytes_per_sample, int w);
// ... swept over its pool stride roundup(w, 32) with zero costs in the padding (fresh_pool)
bool fused_padded_plane_eligible(int bytes_per_sample, int w);
bool fused_needs_pools(const sn_config& c);  // subsampled chroma: luma / chroma sweeps coupled through scratch pools
bool fused_layout_ok(const PlaneArgs& p);
// sn_fused_u8_v3.hip: the 8-bit sweep, two virtual wavefronts packed into every register.
bool fused_v3_plane_ok(int w);
// Scratch-pool coupling between the luma sweep and the subsampled-chroma sweeps (sn_fused_u8_v3.hip, Mode).
struct FusedPool {
    int mode;                // 0 = plane on its own, 1 = luma sweep that leaves its smoothed rows, 2 = chroma sweep, 3 = padded plane (no pools)
    int sweep_w;             // luma width (the pool's width)
    const uint8_t* pool_in;  // chroma: what the previous pass left
    uint8_t* pool_out;       // luma / first chroma pass: where this pass leaves its rows (may be null)
    int64_t frame_stride;    // bytes between the pools of consecutive frames
    int pool_rows;           // rows per pool buffer
    int pool_row_bytes;      // > 0 (mode 1): pool_out is a pool of the pool path with this row pitch (Args::pool_row_bytes)
    int rows_in, rows_out;   // valid rows in pool_in / rows to write to pool_out
    int sweep_rows;          // chroma: pool rows to sweep
    int cone_w, cone_nr;     // the chroma plane's width and interpolated lines (dependency cone of the hand-off, Args)
    int cone_in, cone_out;   // extra columns: what this pass loads / stores beyond the final pass's cone
    // nbands > 1: the sweep is cut into bands of rows (sn_fused_v3_common.h); to be verified by launch_band_verify
    int band_rows, band_warm, nbands, band_reset;
    uint32_t* band_state;    // nframes * band_state_words(threads, nbands) words
    int32_t* band_flags;     // one per frame
};
// sn_band.hip: words of band state per frame for a sweep of `threads` threads; the check of a band launch -- flags[f] != 0
// afterwards means frame f has to be redone by the pool path (its guarded launches look at the same flags);
// *fallbacks (device memory, may be null) counts such frames, *host_mirror (host memory the device can write, may be
// null) receives the count as well.
inline int64_t band_state_words(int threads, int nbands) { return (int64_t)nbands * 2 * kBuffers * 8 * threads; }
hipError_t launch_band_verify(hipStream_t s, const uint32_t* state, int threads, int nbands, int nframes, int32_t* flags, int64_t* fallbacks,
                              int64_t* host_mirror);
int fused_v3_waves(int sweep_w);
int64_t fused_v3_pool_bytes(int sweep_w, int rows);
// host: scatter one scratch pool (thread-slot layout) into [9][rows][sweep_w] samples (test hook)
void fused_v3_pool_unpack(const uint32_t* raw, int sweep_w, int rows, uint8_t* out);
hipError_t launch_fused_u8_v3(hipStream_t s, const PlaneArgs& p, double threshold, int nframes, const FusedPool* pool);
// sn_fused_u8_uv.hip: the U and V passes of an 8-bit 4:2:0 frame as one sweep (U in the low halves of the registers, V in the
// high halves two rows behind); `pool`: sweep_w, pool_in / frame_stride / pool_rows / rows_in = the luma sweep's hand-off
// pool, sweep_rows = the U pass's last row.
bool fused_uv_ok(int sweep_w, int region_w, int nk_c, int bh);
hipError_t launch_fused_u8_uv(hipStream_t s, const PlaneArgs& pu, const PlaneArgs& pv, double thr_u, double thr_v, int nframes, const FusedPool& pool);
#ifdef SN_EXPERIMENT_V4  // tools/experiments/sn_fused_u8_v4.hip (four waves per SIMD; slower: profiles/r2_v4_experiment.md)
bool fused_v4_plane_ok(int w);
int fused_v4_waves(int w);
hipError_t launch_fused_u8_v4(hipStream_t s, const PlaneArgs& p, double threshold, int nframes);
#endif
// sn_fused_u16_v3.hip: the same sweep for 9..16-bit samples (one pixel per register, up to 3840 wide).
bool fused_u16_plane_ok(int w);
// sn_fused_f32_v3.hip: the sweep for float samples.
bool fused_f32_plane_ok(int w);
int fused_f32_waves(int sweep_w);
int64_t fused_f32_pool_bytes(int sweep_w, int rows);
void fused_f32_pool_unpack(const uint32_t* raw, int sweep_w, int rows, float* out);
hipError_t launch_fused_f32_v3(hipStream_t s, const PlaneArgs& p, double threshold, int nframes, const FusedPool* pool);
int fused_u16_waves(int sweep_w);
int64_t fused_u16_pool_bytes(int sweep_w, int rows);
void fused_u16_pool_unpack(const uint32_t* raw, int sweep_w, int rows, uint16_t* out);
hipError_t launch_fused_u16_v3(hipStream_t s, const PlaneArgs& p, double threshold, int nframes, const FusedPool* pool);

}  // namespace sn
