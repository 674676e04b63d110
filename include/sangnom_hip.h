/*
 * sangnom_hip.h -- C ABI of libsangnom_hip.so, the MI355X (gfx950) implementation of the
 * SangNom2 edge-directed interpolation hot path.
 *
 * What this boundary replaces in the reference (Asd-g/AviSynth-SangNom2 v0.6.1):
 *   - the per-plane kernel entry   void (SangNom2::*process)(dstp, dstStride, w, h, offset, plane)
 *                                  src/SangNom2.h:58, called at src/SangNom2.cpp:393
 *   - its drivers                  SangNom2::sangnom_c  src/SangNom2.cpp:259-273
 *                                  SangNom2::sangnom_sse src/SangNom2_SSE2.cpp:1258-1272
 *   - the frame assembly around it SangNom2::GetFrame   src/SangNom2.cpp:346-394
 *   - instance state               ctor: thresholds, pool geometry  src/SangNom2.cpp:275-310
 *   - argument validation          Create_SangNom2      src/SangNom2.cpp:407-422
 *
 * The reference has no FFI of its own (it is one C++ plugin); the functions below are what a
 * plugin adapter (host/sangnom2_avs_plugin.cpp) binds instead of calling (this->*process)().
 * Plain C: pointers, sizes and integer return codes only; no exceptions cross this boundary and
 * no torch / HIP types appear in the signatures (a hipStream_t travels as void*).
 *
 * Results are bit-exact to the reference's opt=0 path for 8..16-bit integer formats and for
 * 32-bit float, under the conventions that make the reference's output defined
 * (zero-filled scratch pool, one context == one filter instance, frames in call order).
 *
 * There is NO CPU fallback: every entry point that computes needs a HIP device and fails with
 * SN_ERR_NO_DEVICE / SN_ERR_HIP otherwise.
 */
#ifndef SANGNOM_HIP_H
#define SANGNOM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SN_ABI_VERSION 4

typedef struct sn_context sn_context;

/* Return codes. */
enum {
    SN_OK = 0,
    SN_ERR_INVALID_ARG = 1, /* null pointer, bad struct_size, bad pitch ...                   */
    SN_ERR_CONFIG = 2,      /* rejected by the reference's own checks; message = its text     */
    SN_ERR_HIP = 3,         /* a HIP runtime call failed; message names it                    */
    SN_ERR_NO_DEVICE = 4,   /* no usable HIP device                                           */
    SN_ERR_UNSUPPORTED = 5, /* geometry outside what the kernels handle (see sn_create)       */
    SN_ERR_BUSY = 6         /* sn_submit_host: every slot of the host ring holds a frame      */
};

/* Execution path selection (sn_config.mode). */
enum {
    SN_MODE_AUTO = 0,   /* fused tile kernel where eligible, else the pool path               */
    SN_MODE_POOL = 1,   /* always the three-kernel path over the HBM-resident pool            */
    SN_MODE_FUSED = 2   /* require the fused kernel; sn_create fails if not eligible          */
};

/* Clip format + the script arguments of SangNom2(clip, order, aa, aac, threads, dh, luma,
 * chroma, opt)  (src/SangNom2.cpp:399-435, README.md:20-57).  `threads` is a dummy in the
 * reference and `opt` selects its CPU code path; neither exists here. */
typedef struct sn_config {
    int32_t struct_size;      /* = sizeof(sn_config)                                         */
    int32_t width;            /* luma width, pixels                                          */
    int32_t height;           /* luma height of the INPUT clip (before dh doubles it)        */
    int32_t bytes_per_sample; /* 1, 2 or 4 (float)              vi.ComponentSize()           */
    int32_t bits_per_sample;  /* 8..16 or 32                    vi.BitsPerComponent()        */
    int32_t num_planes;       /* 1 (Y) or 3 (YUV); alpha is never touched (SangNom2.cpp:347) */
    int32_t sub_w;            /* log2 horizontal chroma subsampling                          */
    int32_t sub_h;            /* log2 vertical chroma subsampling                            */
    int32_t order;            /* 0 = field by parity, 1 = keep top, 2 = keep bottom          */
    int32_t aa;               /* luma anti-aliasing strength 0..128 (default 48)             */
    int32_t aac;              /* chroma anti-aliasing strength 0..128 (default 0)            */
    int32_t dh;               /* double the height                                           */
    int32_t luma;             /* process luma (default 1)                                    */
    int32_t chroma;           /* process chroma (default 1)                                  */
    int32_t device;           /* HIP device ordinal                                          */
    int32_t max_batch;        /* frames one sn_process_device_strided call may carry (>= 1)  */
    int32_t mode;             /* SN_MODE_*                                                   */
    int32_t host_depth;       /* frames sn_submit_host may keep in flight (0 = 4)            */
    int32_t isolated_planes;  /* EXTENSION, default 0.  1: every plane is filtered as a Y clip *
                               * of its own (own scratch geometry, nothing shared) -- what    *
                               * ExtractY/U/V -> SangNom2 -> CombinePlanes gives with the      *
                               * reference -- instead of the reference's luma-sized pool that *
                               * subsampled chroma shares with luma (SangNom2.cpp:287-310)    */
    int32_t fresh_pool;       /* EXTENSION, default 0.  1: every plane of every frame is      *
                               * filtered as by a newly created reference instance (scratch   *
                               * zero-filled, planes isolated).  Removes the dependence on     *
                               * earlier frames that the reference has when the width is not   *
                               * a multiple of 32, and with it the need to     *
                               * run such clips frame by frame                                 */
    void*   stream;           /* hipStream_t to run on; NULL = the context creates its own   */
} sn_config;

/* Scheduling policy of a context (sn_create_with_policy / sn_set_policy).  Nothing here changes a result -- every path
 * is exact -- only which kernels run and how much memory they may take.  A zero in any field means "the default".
 * (Rounds 1-2 read these from environment variables; the library reads none now.  A build with -DSN_TEST_HOOKS still
 * honours SN_PREFER_POOL, SN_CHAIN, SN_COPY_THREADS and SN_SCRATCH_BUDGET_MB as overrides, for bisecting in the field.) */
enum {
    SN_SMALL_AUTO = 0,   /* launches of a few frames (a synchronous GetFrame, a short look-ahead) are cut into row   *
                          * bands, or go to the pool kernels where the cost model says so        */
    SN_SMALL_SWEEP = 1   /* whole-plane sweeps whenever the configuration is eligible, whatever the launch size     */
};
typedef struct sn_policy {
    int32_t struct_size;        /* = sizeof(sn_policy)                                                             */
    int32_t small_launches;     /* SN_SMALL_*                                                                      */
    int32_t chain;              /* history-carrying clips as chains of passes: 0 = on, -1 = off;  *
                                 * 1, 2, 4, 8 = on, with at most that many workgroups (CUs) per cost buffer (0 picks *
                                 * 8; wide pool rows allow fewer; 1 is the single-workgroup chain of round 2)       */
    int32_t copy_threads;       /* host threads that stage / copy lines, the caller included; 0 = by core count,   *
                                 * at most 16.  Takes effect when the context first needs them                      */
    int32_t scratch_budget_mb;  /* device scratch per kind (pool slots; hand-off pools of the coupled sweeps; the    *
                                 * chain's ring takes an eighth); 0 = 24576.  Read at creation only                  */
    int32_t chroma_sweeps;      /* 8-bit 4:2:0 clips in the sweeps: 0 = U and V as ONE sweep where the geometry allows *
                                 * (sn_fused_u8_uv.hip: the U -> V hand-off stays in registers), 1 = one sweep per      *
                                 * chroma plane with a hand-off pool between them (rounds 1-3).  May change any time   */
    int32_t reserved[2];        /* zero                                                                            */
} sn_policy;

/* Geometry and counters of a live context. */
typedef struct sn_info {
    int32_t struct_size;
    int32_t out_height;       /* luma output height                                          */
    int32_t pool_stride;      /* elements per pool row  = roundup(width, 32)                 */
    int32_t pool_rows;        /* bufferHeight + 1                                            */
    int32_t fused_eligible;   /* 1 if the fused kernel serves this configuration             */
    int32_t history_free;     /* 1 if a frame's result cannot depend on earlier frames       */
    int64_t frames;           /* frames processed so far                                     */
    int64_t fused_frames;     /* ... of which by the fused sweeps (a single 4:2:0 frame whose luma *
                               * ran in row bands and whose chroma ran on the pool kernels counts) */
    int32_t coupled_rows;     /* rows per buffer the fused 4:2:0 sweeps hand from plane to   *
                               * plane (0: this configuration has no such hand-off)          */
    int32_t uv_sweeps;        /* 1 once U and V of a 4:2:0 frame have run as ONE sweep on this *
                               * context (8-bit; sn_policy.chroma_sweeps), else 0              */
    double  threshold[3];     /* aaf[plane] after conversion to the sample type              */
    int64_t banded_frames;    /* ... of fused_frames, swept in row bands (small launches)    */
    int64_t band_fallbacks;   /* ... of which the check sent to the pool path (up to the     *
                               * last synchronisation)                                       */
    int64_t chained_frames;   /* frames of a history-carrying clip whose passes ran as one   *
                               * chain (several frames per launch, or one frame with two or  *
                               * three processed planes) instead of one pass at a time       */
    int64_t chain_redone;     /* launches of such chains over several workgroups per buffer  *
                               * in which a workgroup gave up waiting for its neighbour (it   *
                               * was not scheduled next to it in time) and which were redone  *
                               * on one workgroup per buffer before their frames were handed  *
                               * on (up to the last synchronisation); the frames are right    *
                               * either way                                                   */
} sn_info;

/* Create_SangNom2's argument checks, same order, same message text (src/SangNom2.cpp:407-422).
 * Returns SN_OK or SN_ERR_CONFIG (+ message). */
int sn_validate(const sn_config* cfg, char* msg, size_t msg_len);

/* Constructor of the filter instance (src/SangNom2.cpp:275-330): thresholds, pool geometry,
 * device pool (zero-filled), stream.  Runs sn_validate first. */
int sn_create(const sn_config* cfg, sn_context** out);
int sn_create_with_policy(const sn_config* cfg, const sn_policy* policy /* NULL = defaults */, sn_context** out);
void sn_destroy(sn_context* ctx);

/* The policy in force / a new one (small_launches, chain, copy_threads and chroma_sweeps may change during a context's life; the
 * scratch budget is fixed at creation and ignored here). */
int sn_get_policy(sn_context* ctx, sn_policy* policy);
int sn_set_policy(sn_context* ctx, const sn_policy* policy);

/* Text of the last error on this context; with ctx == NULL, of the last failed sn_create /
 * sn_validate on the calling thread.  Never NULL. */
const char* sn_last_error(const sn_context* ctx);

/* One GetFrame (src/SangNom2.cpp:332-397) with HOST planes: H2D, kernels, D2H, synchronous.
 * src planes have the input geometry, dst planes the output geometry; pitches in bytes, any
 * value >= row size.  parity = child->GetParity(n), used only when order == 0.
 * Only the lines the filter reads (the kept field of a processed plane) are sent to the device, and only the
 * interpolated lines come back: the kept lines of the output -- copies of source lines, src/SangNom2.cpp:361-391 --
 * and planes that are merely copied are written from src to dst on the host meanwhile.  src and dst planes must
 * not overlap. */
int sn_process_host(sn_context* ctx, const void* const src[3], const int32_t src_pitch[3],
                    void* const dst[3], const int32_t dst_pitch[3], int32_t parity);

/* The same with DEVICE planes, asynchronous on the context's stream. */
int sn_process_device(sn_context* ctx, const void* const src[3], const int32_t src_pitch[3],
                      void* const dst[3], const int32_t dst_pitch[3], int32_t parity);

/* nframes (<= max_batch) device-resident frames in one call: frame f's plane p starts at
 * src[p] + f * src_frame_stride[p] bytes (likewise dst).  Equivalent to nframes calls of
 * sn_process_device in order; when the configuration is history-free the frames are processed
 * concurrently.  parity may be NULL (= all 1). */
int sn_process_device_strided(sn_context* ctx, int32_t nframes,
                              const void* const src[3], const int64_t src_frame_stride[3],
                              const int32_t src_pitch[3],
                              void* const dst[3], const int64_t dst_frame_stride[3],
                              const int32_t dst_pitch[3], const int32_t* parity);

/* Pipelined host path (what a plugin's GetFrame with look-ahead binds).  The context owns
 * sn_host_slots() frame slots (about cfg.host_depth; fewer if scratch is short), each with pinned staging and
 * device buffers for one source and one output frame.  sn_submit_host copies the source planes into the next
 * slot (planes in pinned memory are not copied: see "Pinned host frames" below) and queues its H2D without waiting; the slots form up to four groups, and when a group is full (or one
 * of its frames is collected early) one launch sweeps its frames on the group's stream, D2H behind it.
 * sn_collect_host waits for the slot's group and copies the output planes out.  Transfers and sweeps of
 * different groups overlap.  Slots are handed out round-robin, so collecting in submission order never blocks
 * on a later frame; SN_ERR_BUSY means the next slot has not been collected.  History-carrying configurations
 * (sn_info.history_free == 0) still sweep their frames in submission order.
 * Not thread-safe per context; do not interleave with the batch entry points without sn_synchronize. */
int sn_host_slots(sn_context* ctx);
int sn_submit_host(sn_context* ctx, const void* const src[3], const int32_t src_pitch[3], int32_t parity,
                   int32_t* slot);
int sn_collect_host(sn_context* ctx, int32_t slot, void* const dst[3], const int32_t dst_pitch[3]);

/* Pinned host frames (optional).  A host that owns its frame memory can pin it once (hipHostRegister under the
 * hood; the caller keeps the memory alive and unpins it before freeing it).  Planes that lie inside pinned memory move
 * over PCIe directly: sn_process_host and sn_submit_host skip the staging copy of the source, and sn_submit_host_to
 * -- sn_submit_host with the destination planes named at submission, as a plugin's GetFrame has them
 * (env->NewVideoFrame before the kernel runs, src/SangNom2.cpp:344) -- has the output written straight into them;
 * sn_collect_host (dst = NULL allowed then) only waits.  Unpinned planes take the staged route as before, plane by
 * plane.  When every destination plane of a submission is pinned, the kept lines of the output are copied from src
 * to dst on the host during sn_submit_host_to and only the interpolated lines cross PCIe afterwards: the caller must
 * leave the announced planes alone until the slot is collected (and src, dst must not overlap).  Process-wide
 * registry, thread-safe.
 * LIFETIME OF A PINNED SOURCE: sn_submit_host / sn_submit_host_to do not capture a source plane that lies in pinned
 * memory -- the transfer reads the caller's memory asynchronously -- so such a plane must stay valid and UNCHANGED
 * until its slot has been collected (a host that recycles frame buffers keeps the frame referenced until then, as
 * host/sangnom2_filter.hpp does).  Unpinned source planes are copied into the slot's staging before the call
 * returns, as ever.  sn_unpin_host_buffer waits for the outstanding work of every device this library has a context on
 * before it unregisters; when the runtime refuses to unregister, the range stays pinned and known (SN_ERR_HIP).
 * EXPERIMENTAL on ROCm 7.2 / gfx950: a process that has registered and later UNREGISTERED host memory was seen to abort,
 * about once in twenty young processes, inside a LATER asynchronous copy from or to PAGEABLE memory -- the runtime's
 * own pin-on-the-fly path, torch's transfers included; never inside this library, which stages every pageable plane
 * through its own pinned buffers (profiles/r3_page_fault.md, tools/repro_pageable_after_unpin.py).  A host that uses
 * these two entry points should (a) pin its frame arenas once and unpin them only at shutdown, and (b) keep its other
 * device transfers on pinned memory as well.  Without them nothing is registered and the hazard does not arise. */
int sn_pin_host_buffer(void* ptr, size_t bytes);
int sn_unpin_host_buffer(void* ptr);
int sn_submit_host_to(sn_context* ctx, const void* const src[3], const int32_t src_pitch[3], void* const dst[3],
                      const int32_t dst_pitch[3], int32_t parity, int32_t* slot);

/* TurnRight (direction > 0, clockwise) / TurnLeft (direction < 0) of `nframes` device-resident planes of
 * width x height samples of the context's sample type, on the context's stream: dst is height wide and width
 * high.  For pipelines that keep frames on the GPU between the two SangNom2 passes of an anti-aliasing script
 * (TurnLeft().SangNom2().TurnRight().SangNom2()).  No counterpart in the reference. */
int sn_turn_device(sn_context* ctx, int32_t direction, int32_t nframes, const void* src, int64_t src_frame_stride,
                   int32_t src_pitch, int32_t width, int32_t height, void* dst, int64_t dst_frame_stride,
                   int32_t dst_pitch);

/* The anti-aliasing idiom TurnLeft().SangNom2(order, aa, aac).TurnRight().SangNom2(order, aa, aac) as ONE call with
 * host planes (README.md:3 of the reference: "mainly used in anti-aliasing scripts"): the frame
 * crosses PCIe once each way and stays on the device between the two passes (two filter instances -- one for the
 * turned clip, one for the clip itself -- on one stream, sn_turn_device in between).  `cfg` describes the clip
 * (dh must be 0; luma / chroma / isolated_planes / fresh_pool / device apply to both passes; max_batch, mode, stream and
 * host_depth are ignored); the turned clip must pass the reference's checks too (sn_aa_create reports the first
 * pass's message otherwise).  The result is what that script gives with the reference: two instances, so for widths
 * or heights that are not a multiple of 32 each pass carries its own pool history from frame to frame.  Not a
 * function of the reference; host/sangnom2_avs_plugin.cpp registers it as SangNomAA. */
typedef struct sn_aa_context sn_aa_context;
int sn_aa_create(const sn_config* cfg, sn_aa_context** out);
int sn_aa_create_with_policy(const sn_config* cfg, const sn_policy* policy /* NULL = defaults; both passes */, sn_aa_context** out);
int sn_aa_process_host(sn_aa_context* ctx, const void* const src[3], const int32_t src_pitch[3], void* const dst[3],
                       const int32_t dst_pitch[3], int32_t parity);
const char* sn_aa_last_error(const sn_aa_context* ctx); /* ctx == NULL: last failed sn_aa_create on this thread */
void sn_aa_destroy(sn_aa_context* ctx);

int sn_synchronize(sn_context* ctx);
void* sn_get_stream(sn_context* ctx); /* the hipStream_t the context launches on */
int sn_get_info(sn_context* ctx, sn_info* info);

/* Test hook: copy the device scratch pool of batch slot `slot` to host memory
 * (9 x pool_rows x pool_stride elements).  Synchronises the stream. */
int sn_debug_read_pool(sn_context* ctx, int32_t slot, void* host_dst, size_t bytes);

/* Test hook for the fused 4:2:0 sweeps: the smoothed rows one plane's sweep left for the next one
 * (which = 0: luma -> U, 1: U -> V) of batch slot 0, rearranged to 9 x coupled_rows x width samples --
 * what rows 0..coupled_rows-1 of the reference's shared pool (src/SangNom2.cpp:322-329) hold at that
 * point, row 0 unused.  Synchronises the stream. */
int sn_debug_read_coupled_rows(sn_context* ctx, int32_t which, void* host_dst, size_t bytes);

/* Test hook for the row-band sweeps that serve small launches in SN_MODE_AUTO (a frame is cut into bands of
 * rows that start `warm_rows` rows early from a guessed state and are verified; a frame that fails the check is
 * redone by the pool path): bands > 0 forces that many bands per frame, 0 restores the automatic choice, < 0
 * turns the bands off; warm_rows = 0 restores the default run-up.  A run-up of 1 makes nearly every frame fail. */
int sn_debug_set_bands(sn_context* ctx, int32_t bands, int32_t warm_rows);

/* Test hook for the chains over several workgroups per cost buffer (sn_policy.chain): the NEXT such launch starts with
 * its fault word up, as if a workgroup had given up waiting at once -- its waves then take rows before they are written,
 * as after a real time-out -- so that the guarded redo behind the launch has something to repair.  The frames must come
 * out right and sn_info.chain_redone must count the launch.  SN_ERR_UNSUPPORTED if the context has not run a chain yet. */
int sn_debug_raise_chain_fault(sn_context* ctx);

int sn_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif
