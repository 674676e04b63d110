// sn_api.hip -- the C ABI of libsangnom_hip.so (include/sangnom_hip.h): context life cycle,
// argument validation, and per-frame dispatch of the HIP kernels.
//
// Mirrors, on the host side, the reference's constructor and GetFrame
// (/root/reference/src/SangNom2.cpp:275-330 and :332-397); all pixel work happens in the kernels
// of sn_pool_kernels.hip and the fused sweeps (sn_fused_*_v3.hip).  There is no CPU fallback in this library.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "sn_copier.h"
#include "sn_internal.h"

constexpr int kSyncChunks = 4;  // chunks of rows a pageable plane of the synchronous call is staged in

namespace sn {



struct Context {
    sn_config cfg{};
    sn_policy policy{};  // zeros = defaults (sangnom_hip.h)
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    bool counted_live = false;  // this context is counted in g_live_contexts[device]
    int64_t uv_frames = 0;      // frames whose U and V passes ran as one sweep (sn_fused_u8_uv.hip)
    bool uv_geometry = false;   // ... and whether this clip's geometry allows that (then only the luma -> U pool exists from the start)

    int out_height = 0;  // vi.height after dh, SangNom2.cpp:284-285
    int stride_e = 0;    // SangNom2.cpp:287
    int bh = 0;          // SangNom2.cpp:288
    float aaf[3] = {0, 0, 0};
    bool process[3] = {true, true, true};
    bool history_free = false;
    bool use_fused = false;

    PoolArgs pool{};  // allocated on first use when the fused kernels serve the configuration

    // cfg.isolated_planes: every plane is its own filter instance (own pool geometry, own pool)
    bool isolated = false;
    PoolArgs plane_pool[3] = {};
    bool plane_fused[3] = {false, false, false};
    // cfg.fresh_pool: on top of that every frame starts from a zero-filled pool (planes narrower than their pool
    // stride are swept over the whole stride with zero costs in the padding columns)
    bool fresh = false;
    bool plane_padded[3] = {false, false, false};
    int slots = 1;    // frames of a batch the pool path runs at once (1: frames in order on slot 0)

    int fslots = 1;   // frames per chunk of the fused 4:2:0 sweeps (each needs its two hand-off pools)

    // fused 8-bit path with subsampled chroma: two scratch pools per batch slot (sn_fused_u8_v3.hip, Mode)
    bool fused420 = false;
    uint8_t* fpool[2] = {nullptr, nullptr};
    int64_t fpool_frame_bytes = 0;
    int fpool_rows = 0;

    // the host ring (sn_submit_host / sn_collect_host): ring_groups x ring_per_group frame slots; a group's
    // frames are swept by one launch on the group's stream
    struct HostGroup {
        hipStream_t stream = nullptr;
        hipEvent_t done = nullptr;   // D2H of the group's latest launch finished
        hipEvent_t swept = nullptr;  // its kernels finished (orders history-carrying configurations)
        int lo = 0, hi = 0;          // slots [lo, hi) of the group are staged and not yet launched
    };
    enum SlotState : uint8_t { kFree = 0, kStaged = 1, kInFlight = 2 };
    std::vector<HostGroup> ring;
    std::vector<uint8_t> slot_state;
    std::vector<int32_t> slot_parity;
    struct SlotDst {  // sn_submit_host_to: where the slot's output goes (pinned planes: straight from the device)
        void* ptr[3] = {nullptr, nullptr, nullptr};
        int32_t pitch[3] = {0, 0, 0};
        bool direct[3] = {false, false, false};
        bool given = false;
        bool kept_on_host = false;  // the output's kept lines were copied from the source at submission: only the interpolated lines travel
    };
    std::vector<SlotDst> slot_dst;
    uint8_t* ring_pin_in[3] = {nullptr, nullptr, nullptr};   // [slot][plane bytes], one allocation per plane
    uint8_t* ring_pin_out[3] = {nullptr, nullptr, nullptr};
    uint8_t* ring_dev_in[3] = {nullptr, nullptr, nullptr};
    uint8_t* ring_dev_out[3] = {nullptr, nullptr, nullptr};
    int64_t ring_bytes_in[3] = {0, 0, 0}, ring_bytes_out[3] = {0, 0, 0};  // per frame
    int ring_per_group = 0;
    Copier* copier = nullptr;
    int host_depth = 0;
    int ring_next = 0;        // slot the next submission takes
    int ring_last = -1;       // group of the latest launch (its `swept` event is what the next launch waits for)
    int ring_pitch_in[3] = {0, 0, 0}, ring_pitch_out[3] = {0, 0, 0};

    // sn_process_host pipelines its planes: copies in on one stream, out on another, the kernels of plane p between
    // the arrival of plane p and its way back (PlaneGate); run_group waits / records per plane where it can
    struct PlaneGate {
        hipStream_t in = nullptr, out = nullptr;
        hipEvent_t arrived[3] = {nullptr, nullptr, nullptr}, done[3] = {nullptr, nullptr, nullptr};
        bool waited[3] = {false, false, false}, recorded[3] = {false, false, false};
        bool on = false;
    } gate;

    // staging for sn_process_host
    uint8_t* stage_src[3] = {nullptr, nullptr, nullptr};
    uint8_t* stage_dst[3] = {nullptr, nullptr, nullptr};
    int stage_src_pitch[3] = {0, 0, 0};
    int stage_dst_pitch[3] = {0, 0, 0};
    // pinned host staging of the synchronous call for planes in PAGEABLE memory: the lines that travel, compact, at the
    // device staging's pitch (kept lines in, interpolated lines out).  The runtime never sees a pageable pointer.
    uint8_t* sync_in[3] = {nullptr, nullptr, nullptr};
    uint8_t* sync_out[3] = {nullptr, nullptr, nullptr};
    hipEvent_t sync_back[3][kSyncChunks] = {};  // chunk k of plane p's interpolated lines has arrived in sync_out[p]

    int64_t frames = 0, fused_frames = 0;
    std::string err;

    // row-band sweeps of small launches (sn_fused_v3_common.h): per scratch slot the bands' state
    // snapshots and the frame's flag; the verification counts failed frames in band_fallbacks_dev and mirrors the count
    // into band_fallbacks, host memory the device can write (what the pause heuristic of band_count looks at)
    // the chain of a history-carrying stream (run_chain): a ring of pool slots, one per pass in flight
    uint8_t* chain_base = nullptr;
    int chain_slots = 0, chain_origin = 0;
    uint32_t* chain_flags = nullptr;   // 8-bit chains over several workgroups per buffer: their round counters (device) ...
    uint32_t* chain_status = nullptr;  // host memory the device writes: launches of chains over several workgroups that timed out and were
                                       // redone on one workgroup per buffer (running count, mirrored after every such launch)
    bool chain_force_fault = false;    // sn_debug_raise_chain_fault: the next such launch starts with its fault word up
    int64_t chained_frames = 0;
    uint32_t* band_state = nullptr;
    int32_t* band_flags = nullptr;
    int64_t* band_fallbacks = nullptr;
    int64_t* band_fallbacks_dev = nullptr;
    int64_t band_words = 0;       // state words per slot
    int64_t banded_frames = 0;
    int band_force = 0;           // sn_debug_set_bands: bands per frame (0 = choose), < 0 = never
    int band_warm = 0;            // rows of run-up (0 = the default of the sample type)
    // content on which the check keeps failing (flat or periodic material) pays for the bands AND the pool path: after
    // a failure the next launches go straight to the pool path, twice as many after every further one
    int64_t band_fallbacks_seen = 0;
    int band_pause = 0, band_pause_next = 0, band_good = 0;

    int plane_w(int p) const { return p == 0 ? cfg.width : cfg.width >> cfg.sub_w; }
    int plane_h_in(int p) const { return p == 0 ? cfg.height : cfg.height >> cfg.sub_h; }
    int plane_h_out(int p) const { return p == 0 ? out_height : out_height >> cfg.sub_h; }
    int nplanes() const { return cfg.num_planes < 3 ? cfg.num_planes : 3; }
    double threshold(int p) const
    {
        switch (cfg.bytes_per_sample) {
        case 1: return (double)(uint8_t)aaf[p];   // float -> T at the finalizePlane_c call,
        case 2: return (double)(uint16_t)aaf[p];  // SangNom2.cpp:272
        default: return (double)aaf[p];
        }
    }
};

static thread_local std::string g_last_error;

// Host memory the caller has pinned through sn_pin_host_buffer (process-wide; a handful of ranges)
struct PinnedRange {
    uintptr_t lo, hi;
};
static std::mutex g_pin_mutex;
static std::vector<PinnedRange> g_pinned;
// Devices on which this process has live contexts (index = device ordinal, value = contexts): what sn_unpin_host_buffer
// has to wait for.  It used to walk every VISIBLE device, which in a multi-rank job that sees all GPUs created a primary
// context on each of them and touched devices that belong to other ranks (round-3 advisor).
static std::mutex g_live_mutex;
static std::vector<int> g_live_contexts;
static void live_context(int device, int delta)
{
    if (device < 0) return;
    std::lock_guard<std::mutex> lk(g_live_mutex);
    if ((size_t)device >= g_live_contexts.size()) g_live_contexts.resize((size_t)device + 1, 0);
    g_live_contexts[(size_t)device] += delta;
}

// is the plane [p, p + pitch * (rows - 1) + row_bytes) inside pinned memory?
static bool plane_is_pinned(const void* p, int pitch, int row_bytes, int rows)
{
    if (rows <= 0) return false;
    const uintptr_t lo = reinterpret_cast<uintptr_t>(p), hi = lo + (uintptr_t)pitch * (rows - 1) + row_bytes;
    std::lock_guard<std::mutex> lk(g_pin_mutex);
    for (const PinnedRange& r : g_pinned)
        if (lo >= r.lo && hi <= r.hi) return true;
    return false;
}

static int fail(Context* c, int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf;
    else g_last_error = buf;
    return code;
}

#define SN_HIP(ctx, call)                                                                      \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(ctx, SN_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_));       \
    } while (0)

// Create_SangNom2, SangNom2.cpp:407-422 (RGB / non-planar clips cannot be described by sn_config;
// `opt` does not exist on this path).
static const char* validate_text(const sn_config& c)
{
    const bool is420 = c.num_planes >= 3 && c.sub_w == 1 && c.sub_h == 1;
    if (c.height % 2 != 0) return "SangNom2: height must be even.";
    if (is420 && c.height % 4) return "SangNom2: height must be mod4.";
    if (c.order < 0 || c.order > 2) return "SangNom2: order must be between 0..2.";
    if (c.aa < 0 || c.aa > 128) return "SangNom2: aa must be between 0..128.";
    if (c.aac < 0 || c.aac > 128) return "SangNom2: aac must be between 0..128.";
    return nullptr;
}

static const char* structural_text(const sn_config& c)
{
    if (c.struct_size != (int32_t)sizeof(sn_config)) return "sn_config.struct_size mismatch";
    if (c.width <= 0 || c.height <= 0) return "width and height must be positive";
    if (c.bytes_per_sample != 1 && c.bytes_per_sample != 2 && c.bytes_per_sample != 4)
        return "bytes_per_sample must be 1, 2 or 4";
    if (c.bytes_per_sample == 1 && c.bits_per_sample != 8) return "8-bit samples need bits_per_sample 8";
    if (c.bytes_per_sample == 2 && (c.bits_per_sample < 9 || c.bits_per_sample > 16))
        return "16-bit containers need bits_per_sample 9..16";
    if (c.bytes_per_sample == 4 && c.bits_per_sample != 32) return "float samples need bits_per_sample 32";
    if (c.num_planes != 1 && c.num_planes != 3) return "num_planes must be 1 or 3";
    if (c.sub_w < 0 || c.sub_w > 2 || c.sub_h < 0 || c.sub_h > 2) return "sub_w / sub_h must be 0..2";
    if (c.num_planes == 3 && ((c.width & ((1 << c.sub_w) - 1)) || (c.height & ((1 << c.sub_h) - 1))))
        return "luma size must be a multiple of the chroma subsampling";
    if (c.max_batch < 0) return "max_batch must be >= 0";
    if (c.host_depth < 0 || c.host_depth > 256) return "host_depth must be 0..256";
    if (c.isolated_planes != 0 && c.isolated_planes != 1) return "isolated_planes must be 0 or 1";
    if (c.fresh_pool != 0 && c.fresh_pool != 1) return "fresh_pool must be 0 or 1";
    if (c.mode < SN_MODE_AUTO || c.mode > SN_MODE_FUSED) return "mode must be SN_MODE_AUTO/POOL/FUSED";
    return nullptr;
}

// A frame's result cannot depend on earlier frames iff every pool cell that a pass reads was
// written earlier in the same frame or is never written at all (rows 0 and bh).  See SURVEY.md
// section 0.7 / 7-H2: needs stride_e == width, and for subsampled chroma a luma pass before it.
static bool compute_history_free(const Context& c)
{
    if (c.stride_e != c.cfg.width) return false;
    if (c.nplanes() == 1) return true;
    const bool luma_pass = c.cfg.dh || c.cfg.luma;
    const bool chroma_pass = c.cfg.dh || c.cfg.chroma;
    if (!chroma_pass) return true;
    if (c.cfg.sub_w == 0 && c.cfg.sub_h == 0) return true;  // chroma rewrites the whole pool
    return luma_pass;
}

}  // namespace sn

using sn::Context;
using sn::Copier;

#ifdef SN_ABORT_TRACE  // diagnostic builds only (make EXTRA=-DSN_ABORT_TRACE): where did an abort() come from?
#include <execinfo.h>
#include <fcntl.h>
#include <signal.h>
#include <unistd.h>
namespace {
void abort_trace(int sig)
{
    const int fd = open("gpurun_out/abort_trace.txt", O_WRONLY | O_CREAT | O_APPEND, 0644);
    if (fd >= 0) {
        void* frames[64];
        const int n = backtrace(frames, 64);
        dprintf(fd, "signal %d, %d frames\n", sig, n);
        backtrace_symbols_fd(frames, n, fd);
        close(fd);
    }
    signal(sig, SIG_DFL);
    raise(sig);
}
struct AbortTraceInstaller {
    AbortTraceInstaller()
    {
        signal(SIGABRT, abort_trace);
        signal(SIGSEGV, abort_trace);
        signal(SIGBUS, abort_trace);
    }
} g_abort_trace_installer;
}  // namespace
#endif

#pragma GCC visibility push(default)
extern "C" {

int sn_abi_version(void) { return SN_ABI_VERSION; }

const char* sn_last_error(const sn_context* ctx)
{
    if (ctx) return reinterpret_cast<const Context*>(ctx)->err.c_str();
    return sn::g_last_error.c_str();
}

int sn_validate(const sn_config* cfg, char* msg, size_t msg_len)
{
    if (!cfg) return sn::fail(nullptr, SN_ERR_INVALID_ARG, "cfg is NULL");
    const char* t = sn::structural_text(*cfg);
    int rc = SN_OK;
    if (t) rc = SN_ERR_INVALID_ARG;
    else if ((t = sn::validate_text(*cfg)) != nullptr) rc = SN_ERR_CONFIG;
    if (msg && msg_len) snprintf(msg, msg_len, "%s", t ? t : "");
    if (t) sn::g_last_error = t;
    return rc;
}

void sn_destroy(sn_context* h)
{
    Context* c = reinterpret_cast<Context*>(h);
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->counted_live) sn::live_context(c->device, -1);
    // every stream that may still run kernels on this context's scratch drains first; only then is memory freed
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (auto& g : c->ring)
        if (g.stream) (void)hipStreamSynchronize(g.stream);
    if (c->pool.base) (void)hipFree(c->pool.base);
    if (c->chain_base) (void)hipFree(c->chain_base);
    if (c->chain_flags) (void)hipFree(c->chain_flags);
    if (c->chain_status) (void)hipHostFree(c->chain_status);
    for (int p = 0; p < 3; ++p)
        if (c->plane_pool[p].base) (void)hipFree(c->plane_pool[p].base);
    for (int i = 0; i < 2; ++i)
        if (c->fpool[i]) (void)hipFree(c->fpool[i]);
    if (c->gate.in) (void)hipStreamSynchronize(c->gate.in);
    if (c->gate.out) (void)hipStreamSynchronize(c->gate.out);
    for (int p = 0; p < 3; ++p) {
        if (c->gate.arrived[p]) (void)hipEventDestroy(c->gate.arrived[p]);
        if (c->gate.done[p]) (void)hipEventDestroy(c->gate.done[p]);
    }
    if (c->gate.in) (void)hipStreamDestroy(c->gate.in);
    if (c->gate.out) (void)hipStreamDestroy(c->gate.out);
    if (c->band_state) (void)hipFree(c->band_state);
    if (c->band_flags) (void)hipFree(c->band_flags);
    if (c->band_fallbacks) (void)hipHostFree(c->band_fallbacks);
    if (c->band_fallbacks_dev) (void)hipFree(c->band_fallbacks_dev);
    delete c->copier;
    for (int p = 0; p < 3; ++p) {
        if (c->ring_pin_in[p]) (void)hipHostFree(c->ring_pin_in[p]);
        if (c->ring_pin_out[p]) (void)hipHostFree(c->ring_pin_out[p]);
        if (c->ring_dev_in[p]) (void)hipFree(c->ring_dev_in[p]);
        if (c->ring_dev_out[p]) (void)hipFree(c->ring_dev_out[p]);
    }
    for (auto& g : c->ring) {
        if (g.done) (void)hipEventDestroy(g.done);
        if (g.swept) (void)hipEventDestroy(g.swept);
        if (g.stream) (void)hipStreamDestroy(g.stream);
    }
    for (int p = 0; p < 3; ++p) {
        if (c->stage_src[p]) (void)hipFree(c->stage_src[p]);
        if (c->stage_dst[p]) (void)hipFree(c->stage_dst[p]);
        if (c->sync_in[p]) (void)hipHostFree(c->sync_in[p]);
        if (c->sync_out[p]) (void)hipHostFree(c->sync_out[p]);
        for (int k = 0; k < kSyncChunks; ++k)
            if (c->sync_back[p][k]) (void)hipEventDestroy(c->sync_back[p][k]);
    }
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

// Environment overrides of the policy exist only in builds with -DSN_TEST_HOOKS (bisecting in the field); the shipped
// library reads no environment variable.
static const char* test_env(const char* name)
{
#ifdef SN_TEST_HOOKS
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

// Device scratch one context may hold per kind (pool-path slots; hand-off pools of the fused 4:2:0 sweeps):
// sn_policy.scratch_budget_mb, 24 GiB by default (the tests shrink it to exercise the chunking on small batches).
static int64_t scratch_budget(const Context* c)
{
    if (const char* e = test_env("SN_SCRATCH_BUDGET_MB"))
        if (atoll(e) > 0) return atoll(e) << 20;
    return c->policy.scratch_budget_mb > 0 ? (int64_t)c->policy.scratch_budget_mb << 20 : 24ll << 30;
}
static bool scratch_budget_is_default(const Context* c) { return c->policy.scratch_budget_mb <= 0 && !test_env("SN_SCRATCH_BUDGET_MB"); }
static bool sweeps_always(const Context* c)  // SN_SMALL_SWEEP
{
    if (const char* e = test_env("SN_PREFER_POOL")) return atoi(e) == 0;
    return c->policy.small_launches == SN_SMALL_SWEEP;
}

// The pool: zero-filled (the convention that makes the reference's output defined).
static int ensure_pool(Context* c, int plane = 0)
{
    sn::PoolArgs& pool = c->isolated ? c->plane_pool[plane] : c->pool;
    if (pool.base) return SN_OK;
    SN_HIP(c, hipMalloc(reinterpret_cast<void**>(&pool.base), (size_t)pool.slot_bytes * c->slots));
    SN_HIP(c, hipMemsetAsync(pool.base, 0, (size_t)pool.slot_bytes * c->slots, c->stream));
    SN_HIP(c, hipStreamSynchronize(c->stream));  // the first user may be a ring slot's stream
    return SN_OK;
}

// A hand-off pool of the coupled sweeps (0: luma -> U, 1: U -> V), fslots frames of it.
static int ensure_fpool(Context* c, int i)
{
    if (c->fpool[i]) return SN_OK;
    SN_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->fpool[i]), (size_t)c->fpool_frame_bytes * c->fslots));
    // cells outside the hand-off's dependency cone are never written: zero, so that the debug read-back is defined
    SN_HIP(c, hipMemsetAsync(c->fpool[i], 0, (size_t)c->fpool_frame_bytes * c->fslots, c->stream));
    if (i == 1) SN_HIP(c, hipStreamSynchronize(c->stream));  // allocated late: its first user may be a ring slot's stream
    return SN_OK;
}

static int create_impl(const sn_config* cfg, Context* c)
{
    c->cfg = *cfg;
    if (c->cfg.max_batch < 1) c->cfg.max_batch = 1;
    c->host_depth = cfg->host_depth > 0 ? cfg->host_depth : 4;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return sn::fail(c, SN_ERR_NO_DEVICE, "no HIP device available (libsangnom_hip has no CPU fallback)");
    if (cfg->device < 0 || cfg->device >= ndev)
        return sn::fail(c, SN_ERR_NO_DEVICE, "HIP device %d out of range (0..%d)", cfg->device, ndev - 1);
    c->device = cfg->device;
    SN_HIP(c, hipSetDevice(c->device));

    // constructor arithmetic, SangNom2.cpp:276-288
    const int aa[3] = {cfg->aa, cfg->aac, cfg->aac};
    for (int i = 0; i < c->nplanes(); ++i)
        c->aaf[i] = cfg->bytes_per_sample < 4
                        ? (aa[i] * 21.0f / 16.0f) * (float)(1 << (cfg->bits_per_sample - 8))
                        : (aa[i] * 21.0f / 16.0f) / 256.0f;
    c->process[0] = cfg->luma != 0;
    c->process[1] = c->process[2] = cfg->chroma != 0;
    c->out_height = cfg->dh ? cfg->height * 2 : cfg->height;
    c->stride_e = (cfg->width + 31) & ~31;
    c->bh = (c->out_height + 1) >> 1;
    if (c->stride_e > 8192)
        return sn::fail(c, SN_ERR_UNSUPPORTED, "width %d exceeds the supported maximum of 8192", cfg->width);
    if (c->out_height / 2 > 65535)
        return sn::fail(c, SN_ERR_UNSUPPORTED, "height %d exceeds the supported maximum", cfg->height);
    c->history_free = sn::compute_history_free(*c);
    c->fresh = cfg->fresh_pool != 0;
    c->isolated = c->fresh || (cfg->isolated_planes != 0 && c->nplanes() > 1);

    bool eligible = sn::fused_eligible(c->cfg);
    if (c->isolated) {
        // Every plane as its own Y clip: pool stride and height from the plane itself, nothing shared -- what a
        // script gets from ExtractY/U/V -> SangNom2 -> CombinePlanes with the reference.  A plane is history-free
        // iff its own width is a multiple of 32, and then the plain fused sweep serves it.
        c->history_free = true;
        eligible = true;
        for (int p = 0; p < c->nplanes(); ++p) {
            sn::PoolArgs& pool = c->plane_pool[p];
            pool.stride_e = (c->plane_w(p) + 31) & ~31;
            pool.bh = (c->plane_h_out(p) + 1) >> 1;
            pool.slot_bytes = ((int64_t)sn::kBuffers * (pool.bh + 1) * pool.stride_e * cfg->bytes_per_sample + 255) & ~(int64_t)255;
            if (!(cfg->dh || c->process[p])) continue;  // copied planes need nothing
            if (pool.stride_e != c->plane_w(p) && !c->fresh) c->history_free = false;
            c->plane_fused[p] = sn::fused_plane_eligible(cfg->bytes_per_sample, c->plane_w(p));
            if (c->fresh && !c->plane_fused[p] && sn::fused_padded_plane_eligible(cfg->bytes_per_sample, c->plane_w(p)))
                c->plane_fused[p] = c->plane_padded[p] = true;
            eligible = eligible && c->plane_fused[p];
        }
    }
    if (cfg->mode == SN_MODE_FUSED && !eligible)
        return sn::fail(c, SN_ERR_UNSUPPORTED, "SN_MODE_FUSED requested but this configuration is not eligible");
    c->use_fused = eligible && cfg->mode != SN_MODE_POOL;

    c->fused420 = c->use_fused && !c->isolated && sn::fused_needs_pools(c->cfg);
    if (c->isolated && cfg->mode == SN_MODE_POOL)
        for (int p = 0; p < 3; ++p) c->plane_fused[p] = false;
    if (cfg->stream) {
        c->stream = reinterpret_cast<hipStream_t>(cfg->stream);
    } else {
        SN_HIP(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        c->own_stream = true;
    }

    // Scratch is bounded: a batch larger than what scratch_budget() holds runs in chunks on the same slots
    // (only history-free configurations batch at all, and for them a slot's old content never matters).
    c->pool.stride_e = c->stride_e;
    c->pool.bh = c->bh;
    c->pool.slot_bytes = (int64_t)sn::kBuffers * (c->bh + 1) * c->stride_e * cfg->bytes_per_sample;
    c->pool.slot_bytes = (c->pool.slot_bytes + 255) & ~(int64_t)255;
    int rc = SN_OK;
    auto fit = [&](int64_t per_frame) {
        const int64_t n = scratch_budget(c) / per_frame;
        const int64_t want = c->cfg.max_batch > c->host_depth ? c->cfg.max_batch : c->host_depth;  // frames in flight
        return (int)(n < 1 ? 1 : n < want ? n : want);
    };
    c->slots = c->history_free ? fit(c->pool.slot_bytes) : 1;
    if (c->use_fused) {
        // the fused sweeps serve this configuration: the pool path only takes the launches that are too small to
        // be worth a sweep (prefer_pool below), so a few slots are enough
        const int64_t cap = (4ll << 30) / c->pool.slot_bytes;
        const int small = (int)(cap < 1 ? 1 : cap > 64 ? 64 : cap);
        if (c->slots > small) c->slots = small;
    }
    if (!c->use_fused) {
        for (int p = 0; p < (c->isolated ? c->nplanes() : 1); ++p) {
            if (c->isolated && (c->plane_fused[p] || !(cfg->dh || c->process[p]))) continue;
            rc = ensure_pool(c, p);
            if (rc != SN_OK) return rc;
        }
    }
    if (c->fused420) {
        // rows the chroma sweeps can reach: 1 .. min(nr_c + 2, bh - 1), plus row 0
        const int nr_c = c->plane_h_out(1) / 2 - 1;
        const int reach = nr_c + 2 < c->bh - 1 ? nr_c + 2 : c->bh - 1;
        c->fpool_rows = reach + 1;
        c->fpool_frame_bytes = cfg->bytes_per_sample == 4   ? sn::fused_f32_pool_bytes(cfg->width, c->fpool_rows)
                               : cfg->bytes_per_sample == 2 ? sn::fused_u16_pool_bytes(cfg->width, c->fpool_rows)
                                                            : sn::fused_v3_pool_bytes(cfg->width, c->fpool_rows);
        // a chunk is three launches of one workgroup per frame: whole rounds of resident workgroups
        // (two waves per SIMD, 256 CUs) leave no partly filled round at the end of each launch
        const int nw = cfg->bytes_per_sample == 4   ? sn::fused_f32_waves(cfg->width)
                       : cfg->bytes_per_sample == 2 ? sn::fused_u16_waves(cfg->width)
                                                    : sn::fused_v3_waves(cfg->width);
        const int round = 256 * (8 / nw);
        // 8-bit clips whose U and V passes run as one sweep (sn_fused_u8_uv.hip) need the luma -> U pool only; the U -> V pool of
        // the two-sweep form is allocated when that form first runs (sn_policy.chroma_sweeps = 1, or a launch of planes the one
        // sweep does not take)
        c->uv_geometry = cfg->bytes_per_sample == 1 && c->plane_w(1) == c->plane_w(2) && c->plane_h_out(1) == c->plane_h_out(2) &&
                         sn::fused_uv_ok(cfg->width, c->plane_w(1), c->plane_h_out(1) / 2, c->bh);
        const int npools = c->uv_geometry ? 1 : 2;
        c->fslots = fit(npools * c->fpool_frame_bytes);  // fused420 implies history-free (fused_eligible)
        if (c->fslots < round && scratch_budget_is_default(c)) {
            // wide float frames: a chunk below one round leaves compute units idle in every launch, so the hand-off
            // pools may take up to a quarter of the device's memory to reach one
            size_t free_b = 0, total_b = 0;
            SN_HIP(c, hipMemGetInfo(&free_b, &total_b));
            const int64_t cap = (int64_t)(total_b / 4 < free_b / 2 ? total_b / 4 : free_b / 2) / (npools * c->fpool_frame_bytes);
            const int64_t want = c->cfg.max_batch > c->host_depth ? c->cfg.max_batch : c->host_depth;
            int64_t n = cap < round ? cap : round;
            if (n > want) n = want;
            if (n > c->fslots) c->fslots = (int)n;
        }
        if (c->fslots > round) c->fslots -= c->fslots % round;
        for (int i = 0; i < npools; ++i) {
            const int rc2 = ensure_fpool(c, i);
            if (rc2 != SN_OK) return rc2;
        }
    }
    SN_HIP(c, hipStreamSynchronize(c->stream));
    return SN_OK;
}

static const char* policy_text(const sn_policy* p)
{
    if (!p) return nullptr;
    if (p->struct_size != (int32_t)sizeof(sn_policy)) return "sn_policy.struct_size mismatch";
    if (p->small_launches != SN_SMALL_AUTO && p->small_launches != SN_SMALL_SWEEP) return "sn_policy.small_launches must be SN_SMALL_AUTO or SN_SMALL_SWEEP";
    if (p->chain != 0 && p->chain != -1 && p->chain != 1 && p->chain != 2 && p->chain != 4 && p->chain != 8)
        return "sn_policy.chain must be 0 (on), -1 (off), or 1, 2, 4, 8 (workgroups per cost buffer of a chain)";
    if (p->copy_threads < 0 || p->copy_threads > 16) return "sn_policy.copy_threads must be 0..16";
    if (p->scratch_budget_mb < 0) return "sn_policy.scratch_budget_mb must not be negative";
    if (p->chroma_sweeps != 0 && p->chroma_sweeps != 1) return "sn_policy.chroma_sweeps must be 0 (U and V as one sweep) or 1 (a sweep each)";
    return nullptr;
}

int sn_create(const sn_config* cfg, sn_context** out) { return sn_create_with_policy(cfg, nullptr, out); }

int sn_create_with_policy(const sn_config* cfg, const sn_policy* policy, sn_context** out)
{
    if (!cfg || !out) return sn::fail(nullptr, SN_ERR_INVALID_ARG, "cfg / out is NULL");
    *out = nullptr;
    char msg[256];
    int rc = sn_validate(cfg, msg, sizeof msg);
    if (rc != SN_OK) return rc;  // g_last_error already holds the text
    if (const char* t = policy_text(policy)) return sn::fail(nullptr, SN_ERR_INVALID_ARG, "%s", t);
    Context* c = new (std::nothrow) Context();
    if (!c) return sn::fail(nullptr, SN_ERR_INVALID_ARG, "out of host memory");
    if (policy) c->policy = *policy;
    c->policy.struct_size = (int32_t)sizeof(sn_policy);
    rc = create_impl(cfg, c);
    if (rc != SN_OK) {
        sn::g_last_error = c->err;
        sn_destroy(reinterpret_cast<sn_context*>(c));
        return rc;
    }
    sn::live_context(c->device, +1);
    c->counted_live = true;
    *out = reinterpret_cast<sn_context*>(c);
    return SN_OK;
}

// GetFrame's field choice, SangNom2.cpp:336-341.
static int field_offset(const Context* c, int parity)
{
    switch (c->cfg.order) {
    case 0: return parity ? 0 : 1;
    case 1: return 0;
    default: return 1;
    }
}

// Of a processed plane only the kept field is ever read (the other lines are the ones being interpolated,
// SangNom2.cpp:361-366), so only those lines need to cross PCIe: line `first`, then every `step`-th, `rows` of them.  A
// copied plane and a double-height clip (whose every line is kept) go as a whole.
struct KeptLines {
    int first, step, rows;
};

static KeptLines kept_lines(const Context* c, int p, int parity)
{
    if (c->cfg.dh || !c->process[p]) return {0, 1, c->plane_h_in(p)};
    return {field_offset(c, parity), 2, c->plane_h_out(p) / 2};
}

// The host's copy threads (staging of the ring, kept lines of the synchronous call): sn_policy.copy_threads in all, the caller included.
static void ensure_copier(Context* c)
{
    if (c->copier) return;
    const char* e = test_env("SN_COPY_THREADS");
    const unsigned hw = std::thread::hardware_concurrency();
    const int asked = e ? atoi(e) : c->policy.copy_threads;  // in all, the caller included; 0 = by core count
    int workers = asked > 0 ? asked - 1 : (hw >= 8 ? 3 : hw >= 4 ? 1 : 0);
    if (workers < 0) workers = 0;
    if (workers > 15) workers = 15;
    c->copier = new Copier(workers);
}

// The other half of the same observation: the kept lines of the OUTPUT are copies of source lines (field copy, border
// line, SangNom2.cpp:361-391), and a plane that is not processed is a copy altogether.  Where the caller's source is
// still at hand (the synchronous call) the host copies those lines itself, while the device is busy, and only the
// interpolated lines come back over PCIe.
static int kept_line_jobs(const Context* c, int p, int parity, const void* src, int sp, void* dst, int dp, Copier::Job* jobs)
{
    const int row = c->plane_w(p) * c->cfg.bytes_per_sample;
    const uint8_t* s = static_cast<const uint8_t*>(src);
    uint8_t* d = static_cast<uint8_t*>(dst);
    if (!(c->cfg.dh || c->process[p])) {
        jobs[0] = {d, s, dp, sp, row, c->plane_h_in(p)};
        return 1;
    }
    const int off = field_offset(c, parity), nk = c->plane_h_out(p) / 2, h_out = c->plane_h_out(p);
    const int first = c->cfg.dh ? 0 : off, step = c->cfg.dh ? 1 : 2;  // kept line k in the source
    jobs[0] = {d + (size_t)off * dp, s + (size_t)first * sp, 2 * dp, step * sp, row, nk};
    if (off == 0) jobs[1] = {d + (size_t)(h_out - 1) * dp, s + (size_t)(first + step * (nk - 1)) * sp, dp, sp, row, 1};  // SangNom2.cpp:380-385
    else jobs[1] = {d, s + (size_t)first * sp, dp, sp, row, 1};                                                       // :386-391
    return 2;
}

static int check_planes(Context* c, const void* const src[3], const int32_t sp[3], void* const dst[3],
                        const int32_t dp[3])
{
    if (!src || !sp || !dst || !dp) return sn::fail(c, SN_ERR_INVALID_ARG, "plane array is NULL");
    for (int p = 0; p < c->nplanes(); ++p) {
        const int rb = c->plane_w(p) * c->cfg.bytes_per_sample;
        if (!src[p] || !dst[p]) return sn::fail(c, SN_ERR_INVALID_ARG, "plane %d pointer is NULL", p);
        if (sp[p] < rb || dp[p] < rb)
            return sn::fail(c, SN_ERR_INVALID_ARG, "plane %d pitch smaller than the row size %d", p, rb);
        if (sp[p] % c->cfg.bytes_per_sample || dp[p] % c->cfg.bytes_per_sample ||
            (uintptr_t)src[p] % c->cfg.bytes_per_sample || (uintptr_t)dst[p] % c->cfg.bytes_per_sample)
            return sn::fail(c, SN_ERR_INVALID_ARG, "plane %d pointer/pitch not aligned to the sample size", p);
    }
    return SN_OK;
}

// A fused sweep walks a plane row by row: its time hardly depends on how many frames the launch carries (up to a round of
// resident workgroups) but it never takes less than rows x 2.6 - 4.7 us.  The pool path spreads ONE frame over the whole chip
// and costs 13 - 33 ps per pool element and frame beyond that.  Both are exact, so in SN_MODE_AUTO a launch that is neither cut
// into row bands (band_count) nor large goes to whichever this estimate says is faster.  Constants: fitted in round 4 to
// tools/prefer_pool_calib.py (profiles/r4_prefer_pool.md: launches of 1 .. 128 device-resident frames through both paths, 1080p /
// 2160p / 4320p, the three sample types, 4:2:0) -- round 1's had the pool path's fixed cost three times too high (stage 2 has
// since moved to the strip kernels) and a row of a sweep that has its CU to itself 30 % too slow, which cancelled except at
// 4320p, where launches of 16 .. 32 frames took the sweeps at up to twice the pool path's time.
//   sweep: rows x t_row; t_row = 2.6 us (8-bit; 3.8 us for planes of more than eight strips: eight waves), 2.8 us (16-bit),
//          4.3 us (float); the chroma planes of a coupled 4:2:0 clip sweep the luma-wide pool: 8-bit U and V as one sweep
//          3.8 us per chroma row for both (1.9 us booked to each plane), otherwise 3.0 us (8-bit) / 3.15 us (16-bit) / 4.7 us
//          (float) per row and plane
//   pool : bh x (a + 0.036 us x ceil(stride / 1024)) + 60 us + n x stride x bh x per_elem per processed plane;
//          a = 0.25 / 0.22 / 0.33 us, per_elem = 12.9 / 19.4 / 32.6 ps for 8-bit / 16-bit / float
static bool prefer_pool(const Context* c, int n, int slot0)
{
    if (c->cfg.mode != SN_MODE_AUTO || !c->history_free || slot0 + n > c->slots) return false;
    if (sweeps_always(c)) return false;  // SN_SMALL_SWEEP (the tests use it to reach the sweeps with small clips)
    const int B = c->cfg.bytes_per_sample;
    const double per_elem = B == 1 ? 12.9e-12 : B == 2 ? 19.4e-12 : 32.6e-12;
    const double a_fix = B == 1 ? 0.25e-6 : B == 2 ? 0.22e-6 : 0.33e-6;
    double fused = 0, pool = 0;
    for (int p = 0; p < c->nplanes(); ++p) {
        if (!(c->cfg.dh || c->process[p])) continue;
        const bool own = c->isolated;  // own pool geometry per plane, else the luma-sized shared pool
        const int stride = own ? c->plane_pool[p].stride_e : c->stride_e;
        const int bh = own ? c->plane_pool[p].bh : c->bh;
        const int rows = c->plane_h_out(p) / 2;
        const bool coupled_chroma = c->fused420 && p > 0;
        double t_row = B == 1 ? (stride > 3840 ? 3.8e-6 : 2.6e-6) : B == 2 ? 2.8e-6 : 4.3e-6;
        if (coupled_chroma) t_row = B == 1 ? (c->uv_geometry && c->policy.chroma_sweeps == 0 ? 1.9e-6 : 3.0e-6) : B == 2 ? 3.15e-6 : 4.7e-6;
        fused += rows * t_row;
        const int cols_per_thread = (stride + 1023) / 1024;
        // (a pass of the shared pool stops at the last row a later pass of this frame can still read: run_group's stop[])
        const int bh_run = !own && rows + 2 < bh ? rows + 2 : bh;
        pool += bh_run * (a_fix + 0.036e-6 * cols_per_thread) + 60e-6 + (double)n * stride * bh_run * per_elem;
    }
    return pool < fused;
}

// Row bands: a launch of a few frames -- a synchronous GetFrame, a short look-ahead -- cannot fill the
// device with one workgroup per frame, so each frame is cut into bands of rows that start from a guessed state and are
// verified afterwards (sn_fused_v3_common.h, sn_band.hip).  The run-up is what the guess needs to be forgotten on
// ordinary content: 8-bit sums settle within 17-23 rows of noise, 16-bit within 30, float within 37.
constexpr int kMaxBands = 128;
constexpr int kMinBandRows = 8;

static int band_warm_rows(const Context* c)
{
    if (c->band_warm > 0) return c->band_warm;
    return c->cfg.bytes_per_sample == 1 ? 32 : c->cfg.bytes_per_sample == 2 ? 40 : 48;
}

constexpr int kMaxBandSlots = 192;  // launches of more frames than that fill the device without bands

// Rows of the shortest processed plane if this context can cut small launches into bands at all, else 0.
static int band_rows_available(const Context* c)
{
    if (c->band_force < 0 || c->cfg.mode != SN_MODE_AUTO || !c->history_free) return 0;
    if (c->isolated) {  // every processed plane must have the sweep for planes on their own (not the padded one)
        for (int p = 0; p < c->nplanes(); ++p)
            if ((c->cfg.dh || c->process[p]) && (!c->plane_fused[p] || c->plane_padded[p])) return 0;
    } else if (!c->use_fused) {
        return 0;
    }
    int nr_min = 1 << 30;
    for (int p = 0; p < c->nplanes(); ++p)
        if (c->cfg.dh || c->process[p]) nr_min = c->plane_h_out(p) / 2 - 1 < nr_min ? c->plane_h_out(p) / 2 - 1 : nr_min;
    return nr_min == (1 << 30) || nr_min < 2 * kMinBandRows ? 0 : nr_min;
}

// Bands per frame for a launch of n frames on slots slot0.., 0 = do not cut.
static int band_count(Context* c, int n, int slot0)
{
    const int nr_min = band_rows_available(c);
    if (nr_min == 0) return 0;
    if (slot0 + n > c->slots || slot0 + n > kMaxBandSlots) return 0;  // the fallback needs the frames' pool slots
    if (c->band_force == 0 && sweeps_always(c)) return 0;  // whole-plane sweeps always (see prefer_pool)
    int nb = c->band_force > 0 ? c->band_force : 512 / n;  // about two workgroups per CU in all
    if (nb > nr_min / kMinBandRows) nb = nr_min / kMinBandRows;
    if (nb > kMaxBands) nb = kMaxBands;
    if (nb < (c->band_force > 0 ? 2 : 3)) return 0;  // a launch of hundreds of frames fills the device with whole-plane sweeps
    if (c->band_force == 0 && nr_min / nb < band_warm_rows(c) / 4) nb = nr_min / (band_warm_rows(c) / 4);  // run-up <= 4 x own rows
    if (nb < 2) return 0;
    if (c->band_force == 0 && c->band_fallbacks) {
        const int64_t seen = *c->band_fallbacks;  // as of the last banded launch that has finished
        if (seen != c->band_fallbacks_seen) {
            c->band_fallbacks_seen = seen;
            c->band_pause_next = c->band_pause_next < 8 ? 8 : c->band_pause_next < 1024 ? 2 * c->band_pause_next : 1024;
            c->band_pause = c->band_pause_next;
            c->band_good = 0;
        } else if (c->band_pause == 0 && ++c->band_good >= 64) {
            c->band_pause_next = 0;
        }
        if (c->band_pause > 0) {
            --c->band_pause;
            return 0;
        }
    }
    return nb;
}

// The chroma half of a small 4:2:0 launch as chains (run_group): three ring slots per frame -- the luma plane's smoothed
// rows (left there by the bands), U's pass, V's pass -- for the frames on pool slots slot0 .. slot0 + n - 1 (launches of the
// host ring run side by side on streams of their own, each on its own slots).  The ring is this context's chain ring (a
// history-free context has no other use for it); false: no room, the pool kernels take the chroma planes one after the
// other as before.
static bool ensure_chroma_chains(Context* c, int slot0, int n, hipStream_t st)
{
    if (c->chain_slots < 0 || c->policy.chain < 0) return false;
    if (sn::pool_chain_lanes(c->cfg.bytes_per_sample, c->stride_e) < 2) return false;
    if (const char* e = test_env("SN_CHAIN"))
        if (atoi(e) == 0) return false;
    if (!c->chain_base) {
        int64_t fit = scratch_budget(c) / 8 / c->pool.slot_bytes;
        const int64_t want = 3 * (int64_t)(c->slots < kMaxBandSlots ? c->slots : kMaxBandSlots);
        if (fit > want) fit = want;
        const int64_t addressable = (int64_t)0xfffffffe / c->pool.slot_bytes;  // the chain kernels address the ring with 32-bit offsets
        if (fit > addressable) fit = addressable;
        fit -= fit % 3;
        if (fit < 3) return false;
        if (hipMalloc(reinterpret_cast<void**>(&c->chain_base), (size_t)c->pool.slot_bytes * fit) != hipSuccess) {
            (void)hipGetLastError();
            c->chain_base = nullptr;
            c->chain_slots = -1;
            return false;
        }
        c->chain_slots = (int)fit;
        // row 0 of every buffer stays zero, as in the pool; waited for once, because the next launch may come on another
        // stream (a group of the host ring)
        if (hipMemsetAsync(c->chain_base, 0, (size_t)c->pool.slot_bytes * fit, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return false;
        c->chain_origin = 0;
    }
    return 3 * (slot0 + n) <= c->chain_slots;
}

// The band fields of one sweep over pool rows 1 .. last, at most `want` bands.
static void set_bands(const Context* c, sn::FusedPool& fp, int want, int last, int slot0, bool first)
{
    int nb = want < last / kMinBandRows ? want : last / kMinBandRows;
    if (nb < 1) nb = 1;
    fp.band_rows = (last + nb - 1) / nb;
    fp.nbands = (last + fp.band_rows - 1) / fp.band_rows;
    if (fp.nbands < 2) {  // a launcher takes one band as "not cut"; two bands of which the second is empty cannot happen
        fp.band_rows = (last + 1) / 2;
        fp.nbands = 2;
    }
    fp.band_warm = band_warm_rows(c);
    fp.band_state = c->band_state + (int64_t)slot0 * c->band_words;
    fp.band_flags = c->band_flags + slot0;
    fp.band_reset = first ? 1 : 0;
}

static int band_threads(const Context* c)
{
    const int B = c->cfg.bytes_per_sample;
    return 64 * (B == 4 ? sn::fused_f32_waves(c->cfg.width) : B == 2 ? sn::fused_u16_waves(c->cfg.width) : sn::fused_v3_waves(c->cfg.width));
}

static int ensure_bands(Context* c)
{
    if (c->band_state) return SN_OK;
    c->band_words = sn::band_state_words(band_threads(c), kMaxBands);
    const int slots = c->slots < kMaxBandSlots ? c->slots : kMaxBandSlots;
    SN_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->band_state), (size_t)c->band_words * 4 * slots));
    SN_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->band_flags), sizeof(int32_t) * slots));
    SN_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->band_fallbacks_dev), sizeof(int64_t)));
    SN_HIP(c, hipMemsetAsync(c->band_fallbacks_dev, 0, sizeof(int64_t), c->stream));
    SN_HIP(c, hipStreamSynchronize(c->stream));
    SN_HIP(c, hipHostMalloc(reinterpret_cast<void**>(&c->band_fallbacks), sizeof(int64_t), hipHostMallocMapped));
    *c->band_fallbacks = 0;
    return SN_OK;
}

// The host-facing entry points (a synchronous GetFrame, the host ring) launch a few frames at a time, i.e. they will
// want the scratch of the small-launch paths: the pool-path pool and the band state.  It is allocated here, BEFORE the
// call queues its copies from pageable host memory -- allocating in the middle of such a call (as the lazy
// ensure_pool / ensure_bands of run_group would) puts hipMalloc / hipHostMalloc between the runtime's in-flight staging
// of those copies and their completion.
constexpr int SN_CHAIN_UNAVAILABLE = -1000;  // internal: run_chain could not get its ring, run_batch falls back
static int chain_planes(const Context* c, int planes[3]);
static int ensure_chain(Context* c, int pn, hipStream_t st);

static int prepare_small_launch_scratch(Context* c)
{
    if (!c->history_free) {
        // a history-carrying clip: even ONE frame with two processed planes is a chain of passes, so the chain's ring is
        // part of what the host-facing calls need -- allocated here, before their copies are queued, not by the first
        // chained launch in the middle of a pipelined call.  A ring that cannot be had leaves the
        // context on the pass-by-pass path (chain_slots = -1), as run_chain's own fallback does.
        int planes[3];
        const int pn = chain_planes(c, planes);
        if (pn > 0) {
            const int rc = ensure_pool(c);
            if (rc != SN_OK) return rc;
            const bool had_ring = c->chain_base != nullptr;
            const int rc2 = ensure_chain(c, pn, c->stream);
            if (rc2 != SN_OK && rc2 != SN_CHAIN_UNAVAILABLE) return rc2;  // (unavailable: the pass-by-pass path, no error pending)
            // only when the ring has just been allocated and zeroed on c->stream: its first user may be a ring slot's stream
            if (!had_ring && c->chain_base) SN_HIP(c, hipStreamSynchronize(c->stream));
        }
        return SN_OK;
    }
    if (c->cfg.mode != SN_MODE_AUTO) return SN_OK;
    int rc = SN_OK;
    if (c->isolated) {
        for (int p = 0; p < c->nplanes() && rc == SN_OK; ++p)
            if (c->cfg.dh || c->process[p]) rc = ensure_pool(c, p);
    } else if (c->use_fused) {
        rc = ensure_pool(c);
    }
    if (rc == SN_OK && band_rows_available(c) > 0) rc = ensure_bands(c);
    return rc;
}

// Runs frames [f0, f0 + n) of a strided batch with one common field offset.
// `st` is the stream to launch on and `slot0` the first scratch slot the frames may use (the batch entry points
// pass the context's stream and slot 0, the host ring one frame on its slot's stream and scratch).
static int run_group(Context* c, hipStream_t st, int slot0, int n, const void* const src[3], const int64_t sfs[3],
                     const int32_t sp[3], void* const dst[3], const int64_t dfs[3], const int32_t dp[3], int f0, int offset)
{
    sn::PlaneArgs pa[3];
    bool fused[3] = {false, false, false};
    for (int p = 0; p < c->nplanes(); ++p) {
        sn::PlaneArgs& a = pa[p];
        a = sn::PlaneArgs{};
        a.src = static_cast<const uint8_t*>(src[p]) + (int64_t)f0 * sfs[p];
        a.dst = static_cast<uint8_t*>(dst[p]) + (int64_t)f0 * dfs[p];
        a.src_frame_stride = sfs[p];
        a.dst_frame_stride = dfs[p];
        a.src_pitch = sp[p];
        a.dst_pitch = dp[p];
        a.w = c->plane_w(p);
        a.h_in = c->plane_h_in(p);
        a.h_out = c->plane_h_out(p);
        a.offset = offset;
        a.dh = c->cfg.dh;
        a.enabled = (c->cfg.dh || c->process[p]) ? 1 : 0;
        fused[p] = a.enabled && (c->isolated ? c->plane_fused[p] : c->use_fused) && sn::fused_layout_ok(a);
    }
    bool all_fused = true;
    for (int p = 0; p < c->nplanes(); ++p) all_fused = all_fused && (fused[p] || !pa[p].enabled);
    const int nbands = all_fused ? band_count(c, n, slot0) : 0;
    if (nbands == 0 && prefer_pool(c, n, slot0))
        for (int p = 0; p < 3; ++p) fused[p] = false;
    // sn_process_host's plane pipeline: a plane's kernels wait for its copy and announce their end; the paths that are
    // not written plane by plane wait for everything first (the caller records what was not announced)
    auto plane_in = [&](int p) -> hipError_t {
        if (!c->gate.on || c->gate.waited[p]) return hipSuccess;
        c->gate.waited[p] = true;
        return hipStreamWaitEvent(st, c->gate.arrived[p], 0);
    };
    auto plane_out = [&](int p) -> hipError_t {
        if (!c->gate.on) return hipSuccess;
        c->gate.recorded[p] = true;
        return hipEventRecord(c->gate.done[p], st);
    };
    if (nbands == 0)
        for (int p = 0; p < c->nplanes(); ++p) SN_HIP(c, plane_in(p));
    // The reference smooths the whole luma-sized pool in every pass (SangNom2.cpp:126-159).  A plane of fewer lines reads
    // back only rows 1 .. nr of it, and a later pass of this frame reads one row further than it smooths -- when nothing
    // is carried into the next frame, rows beyond that are never looked at again and stage 2 of the pool path stops
    // there (4:2:0: the two chroma passes take half the time).  SN_MODE_POOL keeps the full emulation, pool contents included.
    int stop[3] = {0, 0, 0};
    if (c->history_free && c->cfg.mode != SN_MODE_POOL) {
        int later = 0;
        for (int p = c->nplanes() - 1; p >= 0; --p) {
            if (!pa[p].enabled) continue;
            const int own = pa[p].h_out / 2;  // rows 1 .. nr = own - 1
            stop[p] = own > later + 1 ? own : later + 1;
            later = stop[p];
        }
    }
    auto frames_from = [](sn::PlaneArgs a, int i) {
        a.src += (int64_t)i * a.src_frame_stride;
        a.dst += (int64_t)i * a.dst_frame_stride;
        return a;
    };

    auto launch_plain_fused = [&](const sn::PlaneArgs& a, int p, int m) -> hipError_t {
        if (c->cfg.bytes_per_sample == 4) return sn::launch_fused_f32_v3(st, a, c->threshold(p), m, nullptr);
        if (c->cfg.bytes_per_sample == 2) return sn::launch_fused_u16_v3(st, a, c->threshold(p), m, nullptr);
        return sn::launch_fused_u8_v3(st, a, c->threshold(p), m, nullptr);
    };

    // The latency path of one plane: bands, their check, and the pool path for the frames that fail it (its launches
    // exit at once otherwise).
    auto banded_plane = [&](int p, sn::PlaneArgs a, sn::PoolArgs pool) -> int {
        sn::FusedPool fp{};
        fp.mode = 0;  // kPlain, in bands
        set_bands(c, fp, nbands, a.h_out / 2 - 1, slot0, true);
        const int B = c->cfg.bytes_per_sample;
        if (B == 4) SN_HIP(c, sn::launch_fused_f32_v3(st, a, c->threshold(p), n, &fp));
        else if (B == 2) SN_HIP(c, sn::launch_fused_u16_v3(st, a, c->threshold(p), n, &fp));
        else SN_HIP(c, sn::launch_fused_u8_v3(st, a, c->threshold(p), n, &fp));
        const int threads = 64 * (B == 4 ? sn::fused_f32_waves(a.w) : B == 2 ? sn::fused_u16_waves(a.w) : sn::fused_v3_waves(a.w));
        SN_HIP(c, sn::launch_band_verify(st, fp.band_state, threads, fp.nbands, n, fp.band_flags, c->band_fallbacks_dev, c->band_fallbacks));
        a.guard = fp.band_flags;
        pool.guard = fp.band_flags;
        // (no k_assemble: the kept lines the bands have copied are right whatever the check says)
        SN_HIP(c, sn::launch_pool_plane(st, a, pool, B, c->threshold(p), n, slot0));
        return SN_OK;
    };

    if (c->isolated && nbands) {
        int rc = ensure_bands(c);
        if (rc != SN_OK) return rc;
        for (int p = 0; p < c->nplanes(); ++p) {
            SN_HIP(c, plane_in(p));
            if (!pa[p].enabled) {
                SN_HIP(c, sn::launch_assemble(st, pa[p], c->cfg.bytes_per_sample, n));
                SN_HIP(c, plane_out(p));
                continue;
            }
            rc = ensure_pool(c, p);
            if (rc != SN_OK) return rc;
            const sn::PoolArgs& pool = c->plane_pool[p];
            if (c->fresh) SN_HIP(c, hipMemsetAsync(pool.base + (int64_t)slot0 * pool.slot_bytes, 0, (size_t)pool.slot_bytes * n, st));
            rc = banded_plane(p, pa[p], pool);
            if (rc != SN_OK) return rc;
            SN_HIP(c, plane_out(p));
        }
        c->fused_frames += n;
        c->banded_frames += n;
        return SN_OK;
    }

    if (c->isolated) {  // every plane on its own: plain fused sweep or the pool path over the plane's own pool
        bool counted = false;
        for (int p = 0; p < c->nplanes(); ++p) {
            const sn::PlaneArgs& a = pa[p];
            if (fused[p] && c->plane_fused[p]) {
                if (c->plane_padded[p]) {
                    // the sweep covers the whole pool stride: costs are zero outside the plane, the box filter clamps
                    // at the end of the stride (SangNom2.cpp:144-150)
                    sn::FusedPool fp{};
                    fp.mode = 3;  // kPadded
                    fp.sweep_w = c->plane_pool[p].stride_e;
                    fp.pool_rows = 1;
                    fp.sweep_rows = a.h_out / 2 - 1;
                    if (c->cfg.bytes_per_sample == 4) SN_HIP(c, sn::launch_fused_f32_v3(st, a, c->threshold(p), n, &fp));
                    else if (c->cfg.bytes_per_sample == 2) SN_HIP(c, sn::launch_fused_u16_v3(st, a, c->threshold(p), n, &fp));
                    else SN_HIP(c, sn::launch_fused_u8_v3(st, a, c->threshold(p), n, &fp));
                } else {
                    SN_HIP(c, launch_plain_fused(a, p, n));
                }
                if (!counted) c->fused_frames += n;
                counted = true;
                continue;
            }
            SN_HIP(c, sn::launch_assemble(st, a, c->cfg.bytes_per_sample, n));
            if (!a.enabled) continue;
            const int rc = ensure_pool(c, p);
            if (rc != SN_OK) return rc;
            const sn::PoolArgs& pool = c->plane_pool[p];
            for (int i = 0; i < n; i += c->slots) {
                const int m = n - i < c->slots ? n - i : c->slots;
                if (c->fresh) SN_HIP(c, hipMemsetAsync(pool.base + (int64_t)slot0 * pool.slot_bytes, 0, (size_t)pool.slot_bytes * m, st));
                SN_HIP(c, sn::launch_pool_plane(st, frames_from(a, i), pool, c->cfg.bytes_per_sample, c->threshold(p), m, slot0));
            }
        }
        return SN_OK;
    }

    // Fused 4:2:0: the luma sweep leaves its smoothed rows in hand-off pool 0, U reads pool 0 and leaves pool 1,
    // V reads pool 1 -- so the three sweeps of a chunk of frames run back to back on that chunk's pools.
    const bool coupled = c->fused420 && fused[0] && fused[1] && fused[2];
    if (coupled && nbands) {
        // A small launch of a 4:2:0 clip: the luma plane in bands, leaving its smoothed rows where the pool path's luma
        // pass would (the chroma sweeps of the coupling cannot be cut: sn_fused_v3_common.h); the luma plane of a
        // frame that fails the check is redone by the pool kernels; then the chroma planes by the pool kernels.
        int rc = ensure_bands(c);
        if (rc == SN_OK) rc = ensure_pool(c);
        if (rc != SN_OK) return rc;
        const int B = c->cfg.bytes_per_sample;
        // Round 3: U and V as ONE chain of two passes per frame (k_smooth_*_chain; 8-bit: both in the same waves, V fifteen rows
        // behind U) on ring slots 3 f (luma's smoothed rows), 3 f + 1 (U), 3 f + 2 (V), instead of two stage-2 launches one after
        // the other: a 2160p YUV420P8 frame 0.617 -> 0.486 ms on the device, YUV420P16 0.73 -> 0.65.  Not for float samples: two
        // float passes on one CU take longer than one after the other (0.94 -> 1.11 ms).
        const bool chroma_chain = c->cfg.bytes_per_sample != 4 && pa[1].enabled && pa[2].enabled && pa[1].w == pa[2].w && pa[1].h_out == pa[2].h_out && pa[1].w % 8 == 0 &&
                                  ensure_chroma_chains(c, slot0, n, st);
        sn::PoolArgs ring = c->pool;
        if (chroma_chain) {
            ring.base = c->chain_base;
            ring.guard = nullptr;
            ring.rows = 0;
            ring.slot_step = 3;
            ring.slot_mod = c->chain_slots;
        }
        sn::FusedPool fp{};
        fp.mode = 1;  // kLumaSpill
        fp.sweep_w = c->cfg.width;
        fp.pool_out = chroma_chain ? c->chain_base + (int64_t)3 * slot0 * c->pool.slot_bytes : c->pool.base + (int64_t)slot0 * c->pool.slot_bytes;
        fp.frame_stride = chroma_chain ? 3 * c->pool.slot_bytes : c->pool.slot_bytes;
        fp.pool_rows = c->bh + 1;
        fp.pool_row_bytes = c->stride_e * B;
        fp.rows_out = stop[1] < c->bh - 1 ? stop[1] : c->bh - 1;  // what U's stage 2 reads: rows up to the one it stops at
        fp.cone_nr = 1 << 20;  // every column is kept
        set_bands(c, fp, nbands, pa[0].h_out / 2 - 1, slot0, true);
        SN_HIP(c, plane_in(0));
        if (B == 4) SN_HIP(c, sn::launch_fused_f32_v3(st, pa[0], c->threshold(0), n, &fp));
        else if (B == 2) SN_HIP(c, sn::launch_fused_u16_v3(st, pa[0], c->threshold(0), n, &fp));
        else SN_HIP(c, sn::launch_fused_u8_v3(st, pa[0], c->threshold(0), n, &fp));
        SN_HIP(c, sn::launch_band_verify(st, fp.band_state, band_threads(c), fp.nbands, n, fp.band_flags, c->band_fallbacks_dev, c->band_fallbacks));
        sn::PlaneArgs a = pa[0];
        a.guard = fp.band_flags;
        sn::PoolArgs pool = chroma_chain ? ring : c->pool;
        pool.guard = fp.band_flags;
        pool.rows = stop[0];
        SN_HIP(c, sn::launch_pool_plane(st, a, pool, B, c->threshold(0), n, chroma_chain ? 3 * slot0 : slot0));  // (kept lines: already copied by the bands)
        SN_HIP(c, plane_out(0));
        if (chroma_chain) {
            sn::ChainArgs ch{};
            ch.npass = 2;
            ch.pn = 2;
            ch.origin = 3 * slot0;
            ch.rows = stop[1] > stop[2] ? stop[1] : stop[2];
            ch.nchains = n;
            ch.chain_step = 3;
            for (int p = 1; p < 3; ++p) {
                ch.w[p - 1] = pa[p].w;
                ch.nr[p - 1] = pa[p].h_out / 2 - 1;
                SN_HIP(c, plane_in(p));
                SN_HIP(c, sn::launch_assemble(st, pa[p], B, n));
                SN_HIP(c, sn::launch_pool_prepare(st, pa[p], ring, B, n, 3 * slot0 + p));
            }
            SN_HIP(c, sn::launch_pool_chain(st, ring, ch, B));
            for (int p = 1; p < 3; ++p) {
                SN_HIP(c, sn::launch_pool_finalize(st, pa[p], ring, B, c->threshold(p), n, 3 * slot0 + p));
                SN_HIP(c, plane_out(p));
            }
            c->fused_frames += n;
            c->banded_frames += n;
            return SN_OK;
        }
        for (int p = 1; p < 3; ++p) {
            pool = c->pool;
            pool.rows = stop[p];
            SN_HIP(c, plane_in(p));
            SN_HIP(c, sn::launch_assemble(st, pa[p], B, n));
            SN_HIP(c, sn::launch_pool_plane(st, pa[p], pool, B, c->threshold(p), n, slot0));
            SN_HIP(c, plane_out(p));
        }
        c->fused_frames += n;  // (luma through the sweeps; banded_frames counts a subset of fused_frames, sangnom_hip.h)
        c->banded_frames += n;
        return SN_OK;
    }
    if (coupled) {
        const int nr_c = c->plane_h_out(1) / 2 - 1;
        const int reach = c->fpool_rows - 1;
        const int sweep_u = nr_c + 1 < c->bh - 1 ? nr_c + 1 : c->bh - 1;
        // 8-bit: U and V as ONE sweep (sn_fused_u8_uv.hip) -- only the luma -> U hand-off goes through a pool
        const bool one_chroma_sweep = c->uv_geometry && c->policy.chroma_sweeps == 0 && pa[1].enabled && pa[2].enabled && pa[1].h_in == pa[2].h_in;
        if (!one_chroma_sweep) {
            const int rc2 = ensure_fpool(c, 1);
            if (rc2 != SN_OK) return rc2;
        }
        for (int i = 0; i < n; i += c->fslots) {
            const int m = n - i < c->fslots ? n - i : c->fslots;
            for (int p = 0; p < 3; ++p) {
                sn::FusedPool fp{};
                fp.sweep_w = c->cfg.width;
                fp.frame_stride = c->fpool_frame_bytes;
                fp.pool_rows = c->fpool_rows;
                fp.cone_w = c->plane_w(1);  // the hand-off's dependency cone (sn_fused_v3_common.h, Args)
                fp.cone_nr = nr_c;
                fp.cone_in = p == 1 ? 6 : 0;
                fp.cone_out = p == 0 ? 6 : 0;
                if (p == 0) {
                    fp.mode = 1;
                    fp.pool_out = c->fpool[0] + (int64_t)slot0 * c->fpool_frame_bytes;
                    fp.rows_out = reach;
                } else {
                    fp.mode = 2;
                    fp.pool_in = c->fpool[p - 1] + (int64_t)slot0 * c->fpool_frame_bytes;
                    fp.pool_out = p == 1 ? c->fpool[1] + (int64_t)slot0 * c->fpool_frame_bytes : nullptr;
                    fp.rows_in = p == 1 ? reach : sweep_u;
                    fp.sweep_rows = p == 1 ? sweep_u : nr_c;
                    fp.rows_out = p == 1 ? sweep_u : 0;
                }
                const sn::PlaneArgs a = frames_from(pa[p], i);
                if (p == 1 && one_chroma_sweep) {
                    SN_HIP(c, sn::launch_fused_u8_uv(st, a, frames_from(pa[2], i), c->threshold(1), c->threshold(2), m, fp));
                    break;
                }
                if (c->cfg.bytes_per_sample == 4) SN_HIP(c, sn::launch_fused_f32_v3(st, a, c->threshold(p), m, &fp));
                else if (c->cfg.bytes_per_sample == 2) SN_HIP(c, sn::launch_fused_u16_v3(st, a, c->threshold(p), m, &fp));
                else SN_HIP(c, sn::launch_fused_u8_v3(st, a, c->threshold(p), m, &fp));
            }
        }
        if (one_chroma_sweep) c->uv_frames += n;
        c->fused_frames += n;
        return SN_OK;
    }

    if (nbands && !c->fused420) {
        // the latency path: bands, their check, and the pool path for the frames that fail it (it exits at once otherwise)
        int rc = ensure_bands(c);
        if (rc == SN_OK) rc = ensure_pool(c);
        if (rc != SN_OK) return rc;
        for (int p = 0; p < c->nplanes(); ++p) {
            sn::PlaneArgs a = pa[p];
            SN_HIP(c, plane_in(p));
            if (!a.enabled) {
                SN_HIP(c, sn::launch_assemble(st, a, c->cfg.bytes_per_sample, n));
                SN_HIP(c, plane_out(p));
                continue;
            }
            rc = banded_plane(p, a, c->pool);
            if (rc != SN_OK) return rc;
            SN_HIP(c, plane_out(p));
        }
        c->fused_frames += n;
        c->banded_frames += n;
        return SN_OK;
    }

    bool counted = false, pool_path = false;
    for (int p = 0; p < c->nplanes(); ++p) {
        const sn::PlaneArgs& a = pa[p];
        if (fused[p] && !c->fused420) {
            SN_HIP(c, launch_plain_fused(a, p, n));
            if (!counted) c->fused_frames += n;
            counted = true;
            continue;
        }
        SN_HIP(c, sn::launch_assemble(st, a, c->cfg.bytes_per_sample, n));
        pool_path = pool_path || a.enabled;
    }
    if (!pool_path) return SN_OK;
    // Pool path, a chunk of frames at a time on the chunk's slots; within a chunk the planes run in the
    // reference's order, because with subsampled chroma a frame's chroma result depends on what its own luma
    // pass left in the slot (SangNom2.cpp:322-329).
    int rc = ensure_pool(c);
    if (rc != SN_OK) return rc;
    for (int i = 0; i < n; i += c->slots) {
        const int m = n - i < c->slots ? n - i : c->slots;
        for (int p = 0; p < c->nplanes(); ++p) {
            if (!pa[p].enabled || (fused[p] && !c->fused420)) continue;
            sn::PoolArgs pool = c->pool;
            pool.rows = stop[p];
            SN_HIP(c, sn::launch_pool_plane(st, frames_from(pa[p], i), pool, c->cfg.bytes_per_sample, c->threshold(p), m, slot0));
        }
    }
    return SN_OK;
}

// History-carrying clips, two passes or more of one field offset (several frames, or one frame with two or three processed
// planes): the reference's passes -- every processed plane of every
// frame, in order, each starting from the pool the pass before it left -- as ONE chain (sn_pool_kernels.hip,
// k_smooth_u8_chain): stage 1 of all passes into a slot each, one stage-2 launch that keeps several passes in
// flight, stage 3 of all passes.  The pool of slot 0 is where the chain starts and where its last pass's pool ends
// up, so frames that come one at a time (and the pool's readers) carry on from there.
constexpr size_t kChainFlagBytes = (size_t)sn::kBuffers * sn::kChainMaxGroups * 32 * sizeof(uint32_t);
// behind the round counters: [0] the fault word of the launch in flight (zeroed with the counters), [32] launches redone so far
constexpr size_t kChainFaultWord = kChainFlagBytes / sizeof(uint32_t), kChainRedoneWord = kChainFaultWord + 32;
constexpr size_t kChainFlagAlloc = kChainFlagBytes + 64 * sizeof(uint32_t);
#ifndef SN_CHAIN_DEFAULT_GROUPS
#define SN_CHAIN_DEFAULT_GROUPS 8
#endif
constexpr int kChainDefaultGroups = SN_CHAIN_DEFAULT_GROUPS;  // sn_policy.chain = 0
constexpr int kChainSlack = 2;  // rounds a workgroup's slots start later than lockstep with the workgroup before it would need

static int chain_planes(const Context* c, int planes[3])
{
    if (c->chain_slots < 0) return 0;
    if (c->history_free || c->isolated || c->cfg.mode == SN_MODE_FUSED) return 0;
    if (sn::pool_chain_lanes(c->cfg.bytes_per_sample, c->stride_e) < 2 || c->bh < 2) return 0;
    if (c->policy.chain < 0) return 0;
    if (const char* e = test_env("SN_CHAIN"))
        if (atoi(e) == 0) return 0;
    int pn = 0;
    for (int p = 0; p < c->nplanes(); ++p) {
        if (!(c->cfg.dh || c->process[p])) continue;
        if (c->plane_w(p) % 8 != 0 || c->plane_h_out(p) / 2 - 1 >= c->bh || c->plane_h_out(p) / 2 - 1 < 1) return 0;
        planes[pn++] = p;
    }
    return pn;
}

// The chain's ring of pool slots: one per pass in flight, sized for the launches this context can see (max_batch or the
// host ring's depth, at most an eighth of the scratch budget; a synchronous single-frame user gets 2 * pn + 1 slots).
static int ensure_chain(Context* c, int pn, hipStream_t st)
{
    if (c->chain_slots < 0) return SN_CHAIN_UNAVAILABLE;
    if (c->chain_base) return SN_OK;
    int64_t fit = scratch_budget(c) / 8 / c->pool.slot_bytes;  // passes per launch
    const int64_t want = (int64_t)(c->cfg.max_batch > c->host_depth ? c->cfg.max_batch : c->host_depth) * pn;
    fit = fit > want ? want : fit;
    fit = fit > 1536 ? 1536 : fit;
    const int64_t addressable = (int64_t)0xfffffffe / c->pool.slot_bytes - 1;  // the chain kernels address the ring with 32-bit offsets
    fit = fit > addressable ? addressable : fit;
    if (addressable < 2 * pn) {  // (a pool slot of more than 600 MB: no chain, the frame-by-frame path)
        c->chain_slots = -1;
        return SN_CHAIN_UNAVAILABLE;
    }
    if (fit < 2 * pn) fit = 2 * pn;
    c->chain_slots = (int)fit + 1;
    if (hipMalloc(reinterpret_cast<void**>(&c->chain_base), (size_t)c->pool.slot_bytes * c->chain_slots) != hipSuccess) {
        (void)hipGetLastError();  // no room for the ring: this context keeps the frame-by-frame path
        c->chain_base = nullptr;
        c->chain_slots = -1;
        return SN_CHAIN_UNAVAILABLE;
    }
    // the round counters and the status word of chains over several workgroups: without them the ring is no use (all or nothing)
    if (hipMalloc(reinterpret_cast<void**>(&c->chain_flags), kChainFlagAlloc) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void**>(&c->chain_status), sizeof(uint32_t), hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        if (c->chain_flags) (void)hipFree(c->chain_flags);
        (void)hipFree(c->chain_base);
        c->chain_flags = nullptr;
        c->chain_status = nullptr;
        c->chain_base = nullptr;
        c->chain_slots = -1;
        return SN_CHAIN_UNAVAILABLE;
    }
    *c->chain_status = 0;
    SN_HIP(c, hipMemsetAsync(c->chain_flags, 0, kChainFlagAlloc, st));
    SN_HIP(c, hipMemsetAsync(c->chain_base, 0, (size_t)c->pool.slot_bytes * c->chain_slots, st));
    c->chain_origin = 0;
    return SN_OK;
}

// A workgroup of a chain over several workgroups that gives up waiting for the one before it (sn_pool_kernels.hip, ChainSync:
// it was never scheduled next to its neighbour -- e.g. another context's sweep holds every CU slot, the reference's
// MT_MULTI_INSTANCE model, src/SangNom2.h:63-66) raises the launch's fault word.  Round 3 turned that into a sticky error
// after wrong frames had been delivered.  Now every such launch is followed by its own guarded redo (run_chain): stage 1 of
// all passes again and the chain on ONE workgroup per buffer, which waits for nobody; they exit at once unless the word is
// up.  The count of redone launches travels to sn_info.chain_redone.
static int chain_fault(Context*) { return SN_OK; }

// Workgroups per buffer for a chain of npass passes.  A short chain -- a single frame's two or three passes, two frames --
// stays on one workgroup, where a pass follows its predecessor without a trip through memory: 720x480 YUV420P8, one frame
// 0.168 ms against 0.193 with eight workgroups per buffer, two frames 0.188 either way, three 0.267 against 0.251, sixteen
// 0.761 against 0.577 (16-bit and float alike: tools/chain_small_groups.sh).  An explicit sn_policy.chain is taken as it is.
static int chain_groups(const Context* c, int npass)
{
    int want = c->policy.chain > 0 ? c->policy.chain : kChainDefaultGroups;
    if (const char* e = test_env("SN_CHAIN_GROUPS")) want = atoi(e);
    else if (c->policy.chain == 0 && npass < 9) want = 1;
    return c->chain_flags ? sn::pool_chain_groups(c->cfg.bytes_per_sample, c->stride_e, want) : 1;
}

static int run_chain(Context* c, hipStream_t st, int n, const void* const src[3], const int64_t sfs[3], const int32_t sp[3],
                     void* const dst[3], const int64_t dfs[3], const int32_t dp[3], int f0, int offset, const int planes[3], int pn)
{
    if (c->chain_slots < 0) return SN_CHAIN_UNAVAILABLE;
    int rc = ensure_pool(c);
    if (rc != SN_OK) return rc;
    const int B = c->cfg.bytes_per_sample;
    rc = ensure_chain(c, pn, st);  // (the host-facing entry points have done this before queueing their copies)
    if (rc != SN_OK) return rc;
    sn::PlaneArgs pa[3];
    for (int p = 0; p < c->nplanes(); ++p) {
        if (c->gate.on && !c->gate.waited[p]) {  // sn_process_host's plane pipeline: the chain needs every plane's copy
            c->gate.waited[p] = true;
            SN_HIP(c, hipStreamWaitEvent(st, c->gate.arrived[p], 0));
        }
        sn::PlaneArgs& a = pa[p];
        a = sn::PlaneArgs{};
        a.src = static_cast<const uint8_t*>(src[p]) + (int64_t)f0 * sfs[p];
        a.dst = static_cast<uint8_t*>(dst[p]) + (int64_t)f0 * dfs[p];
        a.src_frame_stride = sfs[p];
        a.dst_frame_stride = dfs[p];
        a.src_pitch = sp[p];
        a.dst_pitch = dp[p];
        a.w = c->plane_w(p);
        a.h_in = c->plane_h_in(p);
        a.h_out = c->plane_h_out(p);
        a.offset = offset;
        a.dh = c->cfg.dh;
        a.enabled = (c->cfg.dh || c->process[p]) ? 1 : 0;
        SN_HIP(c, sn::launch_assemble(st, a, B, n));
    }
    sn::PoolArgs ring = c->pool;
    ring.base = c->chain_base;
    ring.rows = 0;
    ring.guard = nullptr;
    ring.slot_step = pn;
    ring.slot_mod = c->chain_slots;
    const int per_launch = (c->chain_slots - 1) / pn;  // frames
    for (int i = 0; i < n; i += per_launch) {
        const int m = n - i < per_launch ? n - i : per_launch;
        uint8_t* first = c->chain_base + (int64_t)c->chain_origin * c->pool.slot_bytes;
        SN_HIP(c, hipMemcpyAsync(first, c->pool.base, (size_t)c->pool.slot_bytes, hipMemcpyDeviceToDevice, st));
        sn::ChainArgs ch{};
        ch.npass = m * pn;
        ch.pn = pn;
        ch.origin = c->chain_origin;
        ch.groups = chain_groups(c, ch.npass);
        uint32_t* const fault = c->chain_flags ? c->chain_flags + kChainFaultWord : nullptr;
        if (ch.groups > 1) {
            ch.slack = kChainSlack;
            ch.flags = c->chain_flags;
            ch.status = fault;
            SN_HIP(c, hipMemsetAsync(c->chain_flags, 0, kChainFlagBytes + sizeof(uint32_t), st));  // round counters and the fault word
            if (c->chain_force_fault) {  // test hook: as if the launch timed out at once
                SN_HIP(c, hipMemsetAsync(fault, 1, 1, st));
                c->chain_force_fault = false;
            }
        }
        for (int k = 0; k < pn; ++k) {
            sn::PlaneArgs a = pa[planes[k]];
            a.src += (int64_t)i * a.src_frame_stride;
            a.dst += (int64_t)i * a.dst_frame_stride;
            ch.w[k] = a.w;
            ch.nr[k] = a.h_out / 2 - 1;
            SN_HIP(c, sn::launch_pool_prepare(st, a, ring, B, m, (c->chain_origin + 1 + k) % c->chain_slots));
        }
        SN_HIP(c, sn::launch_pool_chain(st, ring, ch, B));
        if (ch.groups > 1) {
            // the guarded redo (see chain_fault): the costs again -- stage 2 has smoothed them in place -- and the chain on one
            // workgroup per buffer; every cell a pass's slot holds is rewritten by one of the two, the slot the chain starts
            // from (the copy of the pool before this launch) is only ever read
            sn::PoolArgs guarded = ring;
            guarded.guard = reinterpret_cast<const int32_t*>(fault);
            guarded.guard_single = 1;
            for (int k = 0; k < pn; ++k) {
                sn::PlaneArgs a = pa[planes[k]];
                a.src += (int64_t)i * a.src_frame_stride;
                a.dst += (int64_t)i * a.dst_frame_stride;
                a.guard = reinterpret_cast<const int32_t*>(fault);
                a.guard_single = 1;
                SN_HIP(c, sn::launch_pool_prepare(st, a, guarded, B, m, (c->chain_origin + 1 + k) % c->chain_slots));
            }
            sn::ChainArgs again = ch;
            again.groups = 1;
            again.slack = 0;
            again.flags = nullptr;
            again.status = nullptr;
            again.only_if = fault;
            SN_HIP(c, sn::launch_pool_chain(st, ring, again, B));
            SN_HIP(c, sn::launch_chain_redo_count(st, fault, c->chain_flags + kChainRedoneWord, c->chain_status));
        }
        for (int k = 0; k < pn; ++k) {
            sn::PlaneArgs a = pa[planes[k]];
            a.src += (int64_t)i * a.src_frame_stride;
            a.dst += (int64_t)i * a.dst_frame_stride;
            SN_HIP(c, sn::launch_pool_finalize(st, a, ring, B, c->threshold(planes[k]), m, (c->chain_origin + 1 + k) % c->chain_slots));
        }
        c->chain_origin = (c->chain_origin + ch.npass) % c->chain_slots;
        const uint8_t* last = c->chain_base + (int64_t)c->chain_origin * c->pool.slot_bytes;
        SN_HIP(c, hipMemcpyAsync(c->pool.base, last, (size_t)c->pool.slot_bytes, hipMemcpyDeviceToDevice, st));
    }
    c->chained_frames += n;
    return SN_OK;
}

// Splits a batch into runs of equal field offset; history-carrying configurations run one frame at a time on
// pool slot 0, exactly like one reference instance -- or, several frames of one offset, as a chain (run_chain).
static int run_batch(Context* c, hipStream_t st, int slot0, int nframes, const void* const src[3], const int64_t sfs[3],
                     const int32_t sp[3], void* const dst[3], const int64_t dfs[3], const int32_t dp[3], const int32_t* parity)
{
    int f = 0;
    int planes[3] = {0, 0, 0};
    if (int rc = chain_fault(c)) return rc;
    const int pn = slot0 == 0 ? chain_planes(c, planes) : 0;
    while (f < nframes) {
        const int off = field_offset(c, parity ? parity[f] : 1);
        int g = f + 1;
        if (c->history_free || pn)
            while (g < nframes && field_offset(c, parity ? parity[g] : 1) == off) ++g;
        // (a single frame with two or three processed planes is a chain too: its passes follow each other rows apart)
        int rc = !c->history_free && (g - f) * pn > 1 ? run_chain(c, st, g - f, src, sfs, sp, dst, dfs, dp, f, off, planes, pn)
                                               : run_group(c, st, slot0, g - f, src, sfs, sp, dst, dfs, dp, f, off);
        if (rc == SN_CHAIN_UNAVAILABLE) {  // nothing was queued yet: this frame and the rest one at a time
            g = f + 1;
            rc = run_group(c, st, slot0, 1, src, sfs, sp, dst, dfs, dp, f, off);
        }
        if (rc != SN_OK) return rc;
        f = g;
    }
    c->frames += nframes;
    return SN_OK;
}

int sn_process_device_strided(sn_context* h, int32_t nframes, const void* const src[3],
                              const int64_t sfs[3], const int32_t sp[3], void* const dst[3],
                              const int64_t dfs[3], const int32_t dp[3], const int32_t* parity)
{
    Context* c = reinterpret_cast<Context*>(h);
    if (!c) return sn::fail(nullptr, SN_ERR_INVALID_ARG, "ctx is NULL");
    if (nframes < 0 || nframes > c->cfg.max_batch)
        return sn::fail(c, SN_ERR_INVALID_ARG, "nframes %d outside 0..max_batch (%d)", nframes, c->cfg.max_batch);
    if (!sfs || !dfs) return sn::fail(c, SN_ERR_INVALID_ARG, "frame stride array is NULL");
    int rc = check_planes(c, src, sp, dst, dp);
    if (rc != SN_OK) return rc;
    if (nframes == 0) return SN_OK;
    SN_HIP(c, hipSetDevice(c->device));

    return run_batch(c, c->stream, 0, nframes, src, sfs, sp, dst, dfs, dp, parity);
}

int sn_process_device(sn_context* h, const void* const src[3], const int32_t sp[3], void* const dst[3],
                      const int32_t dp[3], int32_t parity)
{
    Context* c = reinterpret_cast<Context*>(h);
    if (!c) return sn::fail(nullptr, SN_ERR_INVALID_ARG, "ctx is NULL");
    const int64_t zero[3] = {0, 0, 0};
    return sn_process_device_strided(h, 1, src, zero, sp, dst, zero, dp, &parity);
}

int sn_process_host(sn_context* h, const void* const src[3], const int32_t sp[3], void* const dst[3],
                    const int32_t dp[3], int32_t parity)
{
    Context* c = reinterpret_cast<Context*>(h);
    if (!c) return sn::fail(nullptr, SN_ERR_INVALID_ARG, "ctx is NULL");
    int rc = check_planes(c, src, sp, dst, dp);
    if (rc != SN_OK) return rc;
    SN_HIP(c, hipSetDevice(c->device));
    rc = prepare_small_launch_scratch(c);
    if (rc != SN_OK) return rc;
    const int B = c->cfg.bytes_per_sample;
    for (int p = 0; p < c->nplanes(); ++p) {
        // every resource is checked on its own: a call that failed half-way (SN_HIP returns) leaves the rest to the next call
        c->stage_src_pitch[p] = (c->plane_w(p) * B + 255) & ~255;
        c->stage_dst_pitch[p] = c->stage_src_pitch[p];
        if (!c->stage_src[p])
            SN_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->stage_src[p]), (size_t)c->stage_src_pitch[p] * c->plane_h_in(p)));
        if (!c->stage_dst[p])
            SN_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->stage_dst[p]), (size_t)c->stage_dst_pitch[p] * c->plane_h_out(p)));
        if (!c->sync_in[p])
            SN_HIP(c, hipHostMalloc(reinterpret_cast<void**>(&c->sync_in[p]), (size_t)c->stage_src_pitch[p] * c->plane_h_in(p), hipHostMallocDefault));
        if (!c->sync_out[p])
            SN_HIP(c, hipHostMalloc(reinterpret_cast<void**>(&c->sync_out[p]), (size_t)c->stage_dst_pitch[p] * (c->plane_h_out(p) / 2 + 1), hipHostMallocDefault));
        for (int k = 0; k < kSyncChunks; ++k)
            if (!c->sync_back[p][k]) SN_HIP(c, hipEventCreateWithFlags(&c->sync_back[p][k], hipEventDisableTiming));
    }
    ensure_copier(c);
    // Several planes: the copies go on streams of their own, so that plane p + 1 arrives and plane p - 1 leaves while plane
    // p is in the kernels (a 2160p YUV420P8 call: 1.16 -> 0.9 ms).  One plane: everything in order on the context's stream.
    Context::PlaneGate& g = c->gate;
    const bool piped = c->nplanes() > 1;
    if (piped && !g.in) {
        SN_HIP(c, hipStreamCreateWithFlags(&g.in, hipStreamNonBlocking));
        SN_HIP(c, hipStreamCreateWithFlags(&g.out, hipStreamNonBlocking));
        for (int p = 0; p < 3; ++p) {
            SN_HIP(c, hipEventCreateWithFlags(&g.arrived[p], hipEventDisableTiming));
            SN_HIP(c, hipEventCreateWithFlags(&g.done[p], hipEventDisableTiming));
        }
    }
    for (int p = 0; p < c->nplanes(); ++p) {
        if (!(c->cfg.dh || c->process[p])) {  // a copied plane never visits the device: the host copies it (kept_line_jobs below)
            if (piped) SN_HIP(c, hipEventRecord(g.arrived[p], g.in));
            continue;
        }
        const KeptLines kl = kept_lines(c, p, parity);
        const uint8_t* from = static_cast<const uint8_t*>(src[p]) + (int64_t)kl.first * sp[p];
        size_t from_pitch = (size_t)kl.step * sp[p];
        if (!sn::plane_is_pinned(src[p], sp[p], c->plane_w(p) * B, c->plane_h_in(p))) {
            // A plane in pageable memory: the host's copy threads bring the lines that travel into the context's pinned
            // staging first (while the previous plane is on the link).  Rounds 1-2 handed the caller's pointer to
            // hipMemcpy2DAsync and left pinning or staging to the runtime; that path produced "DMA buffer failed"
            // fall-backs and, once in a few hundred calls in a long-lived process, a GPU access fault at an address of the
            // caller's heap (profiles/r3_page_fault.md).  A plane inside memory the caller pinned is DMA'd as it lies.
            // in a few chunks of rows, so that a chunk crosses PCIe while the threads copy the next one
            const int chunks = kl.rows >= 4 * kSyncChunks ? kSyncChunks : 1;
            for (int k = 0; k < chunks; ++k) {
                const int y0 = (int)((int64_t)kl.rows * k / chunks), y1 = (int)((int64_t)kl.rows * (k + 1) / chunks);
                const Copier::Job job = {c->sync_in[p] + (size_t)y0 * c->stage_src_pitch[p], from + (size_t)y0 * from_pitch, c->stage_src_pitch[p],
                                         (int)from_pitch, c->plane_w(p) * B, y1 - y0};
                c->copier->run(&job, 1);
                SN_HIP(c, hipMemcpy2DAsync(c->stage_src[p] + ((int64_t)kl.first + (int64_t)y0 * kl.step) * c->stage_src_pitch[p],
                                           (size_t)kl.step * c->stage_src_pitch[p], c->sync_in[p] + (size_t)y0 * c->stage_src_pitch[p],
                                           (size_t)c->stage_src_pitch[p], (size_t)c->plane_w(p) * B, y1 - y0, hipMemcpyHostToDevice,
                                           piped ? g.in : c->stream));
            }
        } else {
            SN_HIP(c, hipMemcpy2DAsync(c->stage_src[p] + (int64_t)kl.first * c->stage_src_pitch[p], (size_t)kl.step * c->stage_src_pitch[p],
                                       from, from_pitch, (size_t)c->plane_w(p) * B, kl.rows, hipMemcpyHostToDevice, piped ? g.in : c->stream));
        }
        if (piped) SN_HIP(c, hipEventRecord(g.arrived[p], g.in));
    }
    const void* dsrc[3] = {c->stage_src[0], c->stage_src[1], c->stage_src[2]};
    void* ddst[3] = {c->stage_dst[0], c->stage_dst[1], c->stage_dst[2]};
    if (piped) {
        g.on = true;
        for (int p = 0; p < 3; ++p) g.waited[p] = g.recorded[p] = false;
    }
    rc = sn_process_device(h, dsrc, c->stage_src_pitch, ddst, c->stage_dst_pitch, parity);
    g.on = false;
    if (rc != SN_OK) {
        if (piped) (void)hipStreamSynchronize(g.in);  // nothing of this call may still be reading the caller's planes
        return rc;
    }
    {   // while the device works: the output's kept lines, from the caller's source to the caller's destination
        Copier::Job jobs[6];
        int nj = 0;
        for (int p = 0; p < c->nplanes(); ++p) nj += kept_line_jobs(c, p, parity, src[p], sp[p], dst[p], dp[p], jobs + nj);
        ensure_copier(c);
        c->copier->run(jobs, nj);
    }
    bool staged_back[3] = {false, false, false};
    for (int p = 0; p < c->nplanes(); ++p) {
        if (piped) {
            if (!g.recorded[p]) SN_HIP(c, hipEventRecord(g.done[p], c->stream));  // a path that is not written plane by plane: all done here
            SN_HIP(c, hipStreamWaitEvent(g.out, g.done[p], 0));
        }
        // only the interpolated lines come back: offset + 1, offset + 3, ... (nr of them)
        const int off = field_offset(c, parity), nr = c->plane_h_out(p) / 2 - 1;
        staged_back[p] = false;
        if ((c->cfg.dh || c->process[p]) && nr > 0) {
            hipStream_t so = piped ? g.out : c->stream;
            const uint8_t* from = c->stage_dst[p] + (int64_t)(off + 1) * c->stage_dst_pitch[p];
            if (!sn::plane_is_pinned(dst[p], dp[p], c->plane_w(p) * B, c->plane_h_out(p))) {  // pageable: through the pinned staging, in chunks
                staged_back[p] = true;
                const int chunks = nr >= 4 * kSyncChunks ? kSyncChunks : 1;
                for (int k = 0; k < chunks; ++k) {
                    const int y0 = (int)((int64_t)nr * k / chunks), y1 = (int)((int64_t)nr * (k + 1) / chunks);
                    SN_HIP(c, hipMemcpy2DAsync(c->sync_out[p] + (size_t)y0 * c->stage_dst_pitch[p], (size_t)c->stage_dst_pitch[p],
                                               from + (size_t)y0 * 2 * c->stage_dst_pitch[p], (size_t)2 * c->stage_dst_pitch[p],
                                               (size_t)c->plane_w(p) * B, y1 - y0, hipMemcpyDeviceToHost, so));
                    SN_HIP(c, hipEventRecord(c->sync_back[p][k], so));
                }
            } else {
                SN_HIP(c, hipMemcpy2DAsync(static_cast<uint8_t*>(dst[p]) + (int64_t)(off + 1) * dp[p], (size_t)2 * dp[p], from,
                                           (size_t)2 * c->stage_dst_pitch[p], (size_t)c->plane_w(p) * B, nr, hipMemcpyDeviceToHost, so));
            }
        }
    }
    for (int p = 0; p < c->nplanes(); ++p) {  // the staged planes: from the pinned staging into the caller's lines, chunk by chunk as they arrive
        if (!staged_back[p]) continue;
        const int off = field_offset(c, parity), nr = c->plane_h_out(p) / 2 - 1;
        const int chunks = nr >= 4 * kSyncChunks ? kSyncChunks : 1;
        for (int k = 0; k < chunks; ++k) {
            const int y0 = (int)((int64_t)nr * k / chunks), y1 = (int)((int64_t)nr * (k + 1) / chunks);
            SN_HIP(c, hipEventSynchronize(c->sync_back[p][k]));
            const Copier::Job job = {static_cast<uint8_t*>(dst[p]) + (size_t)(off + 1 + 2 * y0) * dp[p], c->sync_out[p] + (size_t)y0 * c->stage_dst_pitch[p],
                                     2 * dp[p], c->stage_dst_pitch[p], c->plane_w(p) * B, y1 - y0};
            c->copier->run(&job, 1);
        }
    }
    if (piped) SN_HIP(c, hipStreamSynchronize(g.out));
    SN_HIP(c, hipStreamSynchronize(c->stream));
    // The copies on g.in are complete by now (the kernels waited for them), but the RUNTIME has to hear of it as well: a
    // stream that is never synchronised keeps its commands -- and with them the runtime's transient registration of the
    // caller's pageable planes -- alive after the call has returned and the caller has freed or recycled that memory.
    if (piped) SN_HIP(c, hipStreamSynchronize(g.in));
    return chain_fault(c);
}

// ---- the host ring: SURVEY 8(f)-1, pipelining behind GetFrame --------------------------------------------
// Frame slots with pinned staging and device copies of one source and one output frame each.  The slots form
// up to four groups; a group has a stream of its own, and when its slots are all staged (or one of them is
// collected early) ONE launch sweeps the group's frames: the runtime maps streams onto a handful of hardware
// queues (four by default), so one stream per frame would keep only four 5 ms sweeps in flight (870 frames/s
// at 2160p).  H2D of a frame is queued when it is submitted, D2H behind the group's sweeps; groups overlap.
// History-carrying configurations keep the reference's frame order: a group's sweeps wait for the previous
// launch's sweeps, and all frames use scratch slot 0.
static int ensure_ring(Context* c)
{
    if (!c->ring.empty()) return SN_OK;
    const int B = c->cfg.bytes_per_sample;
    int depth = c->host_depth;
    if (c->history_free) {  // every frame in flight needs scratch of its own
        if (!c->use_fused && depth > c->slots) depth = c->slots;
        if (c->fused420 && depth > c->fslots) depth = c->fslots;
    }
    const int groups = depth < 4 ? depth : 4;
    c->ring_per_group = depth / groups;
    depth = groups * c->ring_per_group;
    for (int p = 0; p < c->nplanes(); ++p) {
        c->ring_pitch_in[p] = (c->plane_w(p) * B + 255) & ~255;
        c->ring_pitch_out[p] = c->ring_pitch_in[p];
        c->ring_bytes_in[p] = (int64_t)c->ring_pitch_in[p] * c->plane_h_in(p);
        c->ring_bytes_out[p] = (int64_t)c->ring_pitch_out[p] * c->plane_h_out(p);
        SN_HIP(c, hipHostMalloc(reinterpret_cast<void**>(&c->ring_pin_in[p]), (size_t)c->ring_bytes_in[p] * depth, hipHostMallocDefault));
        SN_HIP(c, hipHostMalloc(reinterpret_cast<void**>(&c->ring_pin_out[p]), (size_t)c->ring_bytes_out[p] * depth, hipHostMallocDefault));
        SN_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->ring_dev_in[p]), (size_t)c->ring_bytes_in[p] * depth));
        SN_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->ring_dev_out[p]), (size_t)c->ring_bytes_out[p] * depth));
    }
    c->slot_state.assign(depth, Context::kFree);
    c->slot_parity.assign(depth, 1);
    c->slot_dst.assign(depth, Context::SlotDst{});
    c->ring.resize(groups);
    for (auto& g : c->ring) {
        SN_HIP(c, hipStreamCreateWithFlags(&g.stream, hipStreamNonBlocking));
        SN_HIP(c, hipEventCreateWithFlags(&g.done, hipEventDisableTiming));
        SN_HIP(c, hipEventCreateWithFlags(&g.swept, hipEventDisableTiming));
    }
    ensure_copier(c);
    return SN_OK;
}

// Sweeps the staged slots [lo, hi) of group gi and queues their D2H.
static int launch_ring_group(Context* c, int gi)
{
    Context::HostGroup& g = c->ring[gi];
    const int n = g.hi - g.lo;
    if (n <= 0) return SN_OK;
    const int first = gi * c->ring_per_group + g.lo;
    if (!c->history_free && c->ring_last >= 0 && c->ring_last != gi) SN_HIP(c, hipStreamWaitEvent(g.stream, c->ring[c->ring_last].swept, 0));
    const void* dsrc[3] = {nullptr, nullptr, nullptr};
    void* ddst[3] = {nullptr, nullptr, nullptr};
    for (int p = 0; p < c->nplanes(); ++p) {
        dsrc[p] = c->ring_dev_in[p] + (int64_t)first * c->ring_bytes_in[p];
        ddst[p] = c->ring_dev_out[p] + (int64_t)first * c->ring_bytes_out[p];
    }
    const int rc = run_batch(c, g.stream, c->history_free ? first : 0, n, dsrc, c->ring_bytes_in, c->ring_pitch_in, ddst, c->ring_bytes_out,
                             c->ring_pitch_out, &c->slot_parity[first]);
    if (rc != SN_OK) return rc;
    SN_HIP(c, hipEventRecord(g.swept, g.stream));
    const int B = c->cfg.bytes_per_sample;
    for (int p = 0; p < c->nplanes(); ++p) {
        bool one_transfer = true;  // the whole group's plane p in one transfer into the pinned staging?
        for (int k = 0; k < n; ++k) one_transfer = one_transfer && !c->slot_dst[first + k].direct[p] && !c->slot_dst[first + k].kept_on_host;
        if (one_transfer) {
            SN_HIP(c, hipMemcpyAsync(c->ring_pin_out[p] + (int64_t)first * c->ring_bytes_out[p], ddst[p], (size_t)c->ring_bytes_out[p] * n,
                                     hipMemcpyDeviceToHost, g.stream));
            continue;
        }
        for (int k = 0; k < n; ++k) {
            const Context::SlotDst& sd = c->slot_dst[first + k];
            const uint8_t* from = static_cast<const uint8_t*>(ddst[p]) + (int64_t)k * c->ring_bytes_out[p];
            uint8_t* to = sd.direct[p] ? static_cast<uint8_t*>(sd.ptr[p]) : c->ring_pin_out[p] + (int64_t)(first + k) * c->ring_bytes_out[p];
            const int to_pitch = sd.direct[p] ? sd.pitch[p] : c->ring_pitch_out[p];
            if (sd.kept_on_host) {  // only the interpolated lines: offset + 1, offset + 3, ...
                const int off = field_offset(c, c->slot_parity[first + k]), nr = c->plane_h_out(p) / 2 - 1;
                if ((c->cfg.dh || c->process[p]) && nr > 0)
                    SN_HIP(c, hipMemcpy2DAsync(to + (int64_t)(off + 1) * to_pitch, (size_t)2 * to_pitch, from + (int64_t)(off + 1) * c->ring_pitch_out[p],
                                               (size_t)2 * c->ring_pitch_out[p], (size_t)c->plane_w(p) * B, nr, hipMemcpyDeviceToHost, g.stream));
            } else if (!sd.direct[p]) {  // the context's own staging (same pitch on both sides, padding its own): one linear transfer
                SN_HIP(c, hipMemcpyAsync(to, from, (size_t)c->ring_bytes_out[p] - (c->ring_pitch_out[p] - c->plane_w(p) * B), hipMemcpyDeviceToHost, g.stream));
            } else {  // the caller's pinned plane, row by row: the bytes between its rows are not ours to write
                SN_HIP(c, hipMemcpy2DAsync(to, to_pitch, from, c->ring_pitch_out[p], (size_t)c->plane_w(p) * B, c->plane_h_out(p), hipMemcpyDeviceToHost, g.stream));
            }
        }
    }
    SN_HIP(c, hipEventRecord(g.done, g.stream));
    for (int k = 0; k < n; ++k) c->slot_state[first + k] = Context::kInFlight;
    g.lo = g.hi;
    c->ring_last = gi;
    return SN_OK;
}

int sn_host_slots(sn_context* h)
{
    Context* c = reinterpret_cast<Context*>(h);
    if (!c) return 0;
    if (hipSetDevice(c->device) != hipSuccess || ensure_ring(c) != SN_OK) return 0;
    return (int)c->slot_state.size();
}

static int submit_impl(sn_context* h, const void* const src[3], const int32_t sp[3], void* const dst[3], const int32_t dp[3], int32_t parity,
                       int32_t* slot_out)
{
    Context* c = reinterpret_cast<Context*>(h);
    if (!c) return sn::fail(nullptr, SN_ERR_INVALID_ARG, "ctx is NULL");
    if (!src || !sp || !slot_out) return sn::fail(c, SN_ERR_INVALID_ARG, "plane array / slot pointer is NULL");
    const int B = c->cfg.bytes_per_sample;
    for (int p = 0; p < c->nplanes(); ++p) {
        if (!src[p]) return sn::fail(c, SN_ERR_INVALID_ARG, "plane %d pointer is NULL", p);
        if (sp[p] < c->plane_w(p) * B) return sn::fail(c, SN_ERR_INVALID_ARG, "plane %d pitch smaller than the row size %d", p, c->plane_w(p) * B);
    }
    SN_HIP(c, hipSetDevice(c->device));
    int rc = ensure_ring(c);
    if (rc == SN_OK) rc = prepare_small_launch_scratch(c);
    if (rc != SN_OK) return rc;
    const int slot = c->ring_next;
    if (c->slot_state[slot] != Context::kFree)
        return sn::fail(c, SN_ERR_BUSY, "all %d host slots are in flight: collect slot %d first", (int)c->slot_state.size(), slot);
    const int gi = slot / c->ring_per_group, k = slot % c->ring_per_group;
    Context::HostGroup& g = c->ring[gi];
    if (k == 0) g.lo = g.hi = 0;  // the ring came round to this group again

    Context::SlotDst sd{};
    if (dst) {
        if (!dp) return sn::fail(c, SN_ERR_INVALID_ARG, "plane array is NULL");
        sd.given = true;
        for (int p = 0; p < c->nplanes(); ++p) {
            if (!dst[p]) return sn::fail(c, SN_ERR_INVALID_ARG, "plane %d pointer is NULL", p);
            if (dp[p] < c->plane_w(p) * B) return sn::fail(c, SN_ERR_INVALID_ARG, "plane %d pitch smaller than the row size %d", p, c->plane_w(p) * B);
            sd.ptr[p] = dst[p];
            sd.pitch[p] = dp[p];
            sd.direct[p] = sn::plane_is_pinned(dst[p], dp[p], c->plane_w(p) * B, c->plane_h_out(p));
        }
    }
    // (worth it where the interpolated lines then go straight into the caller's pinned planes; with a pageable destination
    // the extra host copy at submission costs the submitting thread more than the PCIe crossing it saves)
    sd.kept_on_host = sd.given;
    for (int p = 0; p < c->nplanes(); ++p) sd.kept_on_host = sd.kept_on_host && sd.direct[p];
    c->slot_dst[slot] = sd;
    // source planes: pinned ones are DMA'd as they lie, the others are staged (copy threads) first; with the destination
    // announced, the output's kept lines (copies of source lines) go there now, on the host
    Copier::Job jobs[9];
    bool direct_in[3] = {false, false, false};
    int njobs = 0;
    if (sd.kept_on_host)
        for (int p = 0; p < c->nplanes(); ++p) njobs += kept_line_jobs(c, p, parity, src[p], sp[p], sd.ptr[p], sd.pitch[p], jobs + njobs);
    bool on_device[3] = {true, true, true};  // a copied plane whose copy the host has just made never visits the device
    for (int p = 0; p < c->nplanes(); ++p) {
        on_device[p] = !(sd.kept_on_host && !(c->cfg.dh || c->process[p]));
        if (!on_device[p]) continue;
        direct_in[p] = sn::plane_is_pinned(src[p], sp[p], c->plane_w(p) * B, c->plane_h_in(p));
        if (!direct_in[p]) {
            const KeptLines kl = kept_lines(c, p, parity);  // only the lines that are read are staged ...
            jobs[njobs++] = {c->ring_pin_in[p] + (int64_t)slot * c->ring_bytes_in[p] + (int64_t)kl.first * c->ring_pitch_in[p],
                             static_cast<const uint8_t*>(src[p]) + (int64_t)kl.first * sp[p], kl.step * c->ring_pitch_in[p], kl.step * sp[p],
                             c->plane_w(p) * B, kl.rows};
        }
    }
    if (njobs) c->copier->run(jobs, njobs);
    for (int p = 0; p < c->nplanes(); ++p) {
        if (!on_device[p]) continue;
        uint8_t* dev = c->ring_dev_in[p] + (int64_t)slot * c->ring_bytes_in[p];
        const KeptLines kl = kept_lines(c, p, parity);  // ... and cross PCIe
        const uint8_t* from = direct_in[p] ? static_cast<const uint8_t*>(src[p]) : c->ring_pin_in[p] + (int64_t)slot * c->ring_bytes_in[p];
        const int from_pitch = direct_in[p] ? sp[p] : c->ring_pitch_in[p];
        if (kl.step == 1 && from_pitch == c->ring_pitch_in[p])  // every line, same pitch on both sides: one linear transfer
            SN_HIP(c, hipMemcpyAsync(dev, from, (size_t)c->ring_bytes_in[p] - (c->ring_pitch_in[p] - c->plane_w(p) * B), hipMemcpyHostToDevice, g.stream));
        else
            SN_HIP(c, hipMemcpy2DAsync(dev + (int64_t)kl.first * c->ring_pitch_in[p], (size_t)kl.step * c->ring_pitch_in[p],
                                       from + (int64_t)kl.first * from_pitch, (size_t)kl.step * from_pitch, (size_t)c->plane_w(p) * B, kl.rows,
                                       hipMemcpyHostToDevice, g.stream));
    }
    c->slot_state[slot] = Context::kStaged;
    c->slot_parity[slot] = parity;
    g.hi = k + 1;
    c->ring_next = (slot + 1) % (int)c->slot_state.size();
    *slot_out = slot;
    if (g.hi == c->ring_per_group) return launch_ring_group(c, gi);
    return SN_OK;
}

int sn_submit_host(sn_context* h, const void* const src[3], const int32_t sp[3], int32_t parity, int32_t* slot_out)
{
    return submit_impl(h, src, sp, nullptr, nullptr, parity, slot_out);
}

int sn_submit_host_to(sn_context* h, const void* const src[3], const int32_t sp[3], void* const dst[3], const int32_t dp[3], int32_t parity,
                      int32_t* slot_out)
{
    if (!dst || !dp) return sn::fail(reinterpret_cast<Context*>(h), SN_ERR_INVALID_ARG, "plane array is NULL");
    return submit_impl(h, src, sp, dst, dp, parity, slot_out);
}

int sn_collect_host(sn_context* h, int32_t slot, void* const dst_arg[3], const int32_t dp_arg[3])
{
    Context* c = reinterpret_cast<Context*>(h);
    if (!c) return sn::fail(nullptr, SN_ERR_INVALID_ARG, "ctx is NULL");
    if (slot < 0 || slot >= (int)c->slot_state.size() || c->slot_state[slot] == Context::kFree)
        return sn::fail(c, SN_ERR_INVALID_ARG, "slot %d holds no frame", slot);
    // the destination: named now, or at submission (sn_submit_host_to)
    const Context::SlotDst sd = c->slot_dst[slot];
    void* const* dst = dst_arg ? dst_arg : (sd.given ? sd.ptr : nullptr);
    const int32_t* dp = dst_arg ? dp_arg : (sd.given ? sd.pitch : nullptr);
    if (!dst || !dp) return sn::fail(c, SN_ERR_INVALID_ARG, "plane array is NULL");
    const int B = c->cfg.bytes_per_sample;
    for (int p = 0; p < c->nplanes(); ++p) {
        if (!dst[p]) return sn::fail(c, SN_ERR_INVALID_ARG, "plane %d pointer is NULL", p);
        if (dp[p] < c->plane_w(p) * B) return sn::fail(c, SN_ERR_INVALID_ARG, "plane %d pitch smaller than the row size %d", p, c->plane_w(p) * B);
    }
    SN_HIP(c, hipSetDevice(c->device));
    const int gi = slot / c->ring_per_group;
    if (c->slot_state[slot] == Context::kStaged) {  // its group is not full yet: sweep what is staged
        const int rc = launch_ring_group(c, gi);
        if (rc != SN_OK) return rc;
    }
    SN_HIP(c, hipEventSynchronize(c->ring[gi].done));
    if (int rc = chain_fault(c)) return rc;
    Copier::Job jobs[3];
    int njobs = 0;
    for (int p = 0; p < c->nplanes(); ++p) {
        if (sd.direct[p] && dst[p] == sd.ptr[p]) continue;  // already written there by the device
        const uint8_t* from = c->ring_pin_out[p] + (int64_t)slot * c->ring_bytes_out[p];
        if (sd.direct[p]) {  // collected into another place than announced: that plane never reached the staging
            for (int y = 0; y < c->plane_h_out(p); ++y)  // host to host: no business of the runtime's
                memcpy(static_cast<uint8_t*>(dst[p]) + (size_t)y * dp[p], static_cast<const uint8_t*>(sd.ptr[p]) + (size_t)y * sd.pitch[p], (size_t)c->plane_w(p) * B);
            continue;
        }
        // (sd.kept_on_host implies direct[p] for every plane: handled above)
        jobs[njobs++] = {static_cast<uint8_t*>(dst[p]), from, dp[p], c->ring_pitch_out[p], c->plane_w(p) * B, c->plane_h_out(p)};
    }
    if (njobs) c->copier->run(jobs, njobs);
    c->slot_state[slot] = Context::kFree;
    c->slot_dst[slot] = Context::SlotDst{};
    return SN_OK;
}

int sn_pin_host_buffer(void* ptr, size_t bytes)
{
    if (!ptr || !bytes) return sn::fail(nullptr, SN_ERR_INVALID_ARG, "sn_pin_host_buffer: NULL or empty");
    const hipError_t e = hipHostRegister(ptr, bytes, hipHostRegisterPortable);
    if (e != hipSuccess) return sn::fail(nullptr, SN_ERR_HIP, "hipHostRegister failed: %s", hipGetErrorString(e));
    std::lock_guard<std::mutex> lk(sn::g_pin_mutex);
    sn::g_pinned.push_back({reinterpret_cast<uintptr_t>(ptr), reinterpret_cast<uintptr_t>(ptr) + bytes});
    return SN_OK;
}

int sn_unpin_host_buffer(void* ptr)
{
    {
        std::lock_guard<std::mutex> lk(sn::g_pin_mutex);
        bool found = false;
        for (size_t i = 0; i < sn::g_pinned.size() && !found; ++i) found = sn::g_pinned[i].lo == reinterpret_cast<uintptr_t>(ptr);
        if (!found) return sn::fail(nullptr, SN_ERR_INVALID_ARG, "sn_unpin_host_buffer: not pinned through sn_pin_host_buffer");
    }
    // Nothing of this process may still be moving data through the mapping: the registry is process-wide and the
    // buffers are registered portable, so a context on any device may have a transfer in flight on a stream of its own.
    // Only devices that HAVE a context of this library are waited for (no primary context is created elsewhere).
    std::vector<int> live;
    {
        std::lock_guard<std::mutex> lk(sn::g_live_mutex);
        for (size_t d = 0; d < sn::g_live_contexts.size(); ++d)
            if (sn::g_live_contexts[d] > 0) live.push_back((int)d);
    }
    int cur = 0;
    const bool have_cur = hipGetDevice(&cur) == hipSuccess;
    if (!have_cur) (void)hipGetLastError();
    for (int d : live)
        if (hipSetDevice(d) == hipSuccess) (void)hipDeviceSynchronize();
    if (have_cur && !live.empty()) (void)hipSetDevice(cur);
    const hipError_t e = hipHostUnregister(ptr);
    if (e != hipSuccess) {  // still registered, and still known to the library: the caller may try again
        (void)hipGetLastError();
        return sn::fail(nullptr, SN_ERR_HIP, "hipHostUnregister failed: %s", hipGetErrorString(e));
    }
    std::lock_guard<std::mutex> lk(sn::g_pin_mutex);
    for (size_t i = 0; i < sn::g_pinned.size(); ++i)
        if (sn::g_pinned[i].lo == reinterpret_cast<uintptr_t>(ptr)) {
            sn::g_pinned.erase(sn::g_pinned.begin() + (long)i);
            break;
        }
    return SN_OK;
}

int sn_turn_device(sn_context* h, int32_t direction, int32_t nframes, const void* src, int64_t sfs, int32_t sp, int32_t width,
                   int32_t height, void* dst, int64_t dfs, int32_t dp)
{
    Context* c = reinterpret_cast<Context*>(h);
    if (!c) return sn::fail(nullptr, SN_ERR_INVALID_ARG, "ctx is NULL");
    const int B = c->cfg.bytes_per_sample;
    if (!src || !dst || direction == 0 || nframes < 0 || width <= 0 || height <= 0)
        return sn::fail(c, SN_ERR_INVALID_ARG, "sn_turn_device: bad pointer, direction or size");
    if (sp < width * B || dp < height * B || sp % B || dp % B || (uintptr_t)src % B || (uintptr_t)dst % B)
        return sn::fail(c, SN_ERR_INVALID_ARG, "sn_turn_device: pitch smaller than the row or misaligned");
    SN_HIP(c, hipSetDevice(c->device));
    SN_HIP(c, sn::launch_turn(c->stream, B, direction > 0 ? 1 : 0, nframes, static_cast<const uint8_t*>(src), sfs, sp, width, height,
                              static_cast<uint8_t*>(dst), dfs, dp));
    return SN_OK;
}

int sn_synchronize(sn_context* h)
{
    Context* c = reinterpret_cast<Context*>(h);
    if (!c) return sn::fail(nullptr, SN_ERR_INVALID_ARG, "ctx is NULL");
    SN_HIP(c, hipSetDevice(c->device));
    SN_HIP(c, hipStreamSynchronize(c->stream));
    for (auto& g : c->ring)
        if (g.stream) SN_HIP(c, hipStreamSynchronize(g.stream));
    return chain_fault(c);
}

void* sn_get_stream(sn_context* h)
{
    Context* c = reinterpret_cast<Context*>(h);
    return c ? reinterpret_cast<void*>(c->stream) : nullptr;
}

int sn_get_info(sn_context* h, sn_info* info)
{
    Context* c = reinterpret_cast<Context*>(h);
    if (!c) return sn::fail(nullptr, SN_ERR_INVALID_ARG, "ctx is NULL");
    if (!info || info->struct_size != (int32_t)sizeof(sn_info))
        return sn::fail(c, SN_ERR_INVALID_ARG, "sn_info.struct_size mismatch");
    info->out_height = c->out_height;
    info->pool_stride = c->stride_e;
    info->pool_rows = c->bh + 1;
    info->fused_eligible = sn::fused_eligible(c->cfg) ? 1 : 0;
    if (c->isolated) {
        info->fused_eligible = 1;
        for (int p = 0; p < c->nplanes(); ++p)
            if ((c->cfg.dh || c->process[p]) && !sn::fused_plane_eligible(c->cfg.bytes_per_sample, c->plane_w(p)) &&
                !(c->fresh && sn::fused_padded_plane_eligible(c->cfg.bytes_per_sample, c->plane_w(p))))
                info->fused_eligible = 0;
    }
    info->history_free = c->history_free ? 1 : 0;
    info->frames = c->frames;
    info->fused_frames = c->fused_frames;
    info->coupled_rows = c->fused420 ? c->fpool_rows : 0;
    info->uv_sweeps = c->uv_frames > 0 ? 1 : 0;
    info->chain_redone = c->chain_status ? (int64_t)__atomic_load_n(c->chain_status, __ATOMIC_RELAXED) : 0;
    info->banded_frames = c->banded_frames;
    info->chained_frames = c->chained_frames;
    info->band_fallbacks = 0;
    if (c->band_fallbacks_dev) {  // the device's own count (the host mirror may lag behind a launch with several failing frames)
        SN_HIP(c, hipSetDevice(c->device));
        SN_HIP(c, hipMemcpy(&info->band_fallbacks, c->band_fallbacks_dev, sizeof(int64_t), hipMemcpyDeviceToHost));
    }
    for (int p = 0; p < 3; ++p) info->threshold[p] = p < c->nplanes() ? c->threshold(p) : 0.0;
    return SN_OK;
}

int sn_debug_read_pool(sn_context* h, int32_t slot, void* host_dst, size_t bytes)
{
    Context* c = reinterpret_cast<Context*>(h);
    if (!c) return sn::fail(nullptr, SN_ERR_INVALID_ARG, "ctx is NULL");
    const size_t need = (size_t)sn::kBuffers * (c->bh + 1) * c->stride_e * c->cfg.bytes_per_sample;
    if (!c->pool.base) return sn::fail(c, SN_ERR_INVALID_ARG, "sn_debug_read_pool: the pool path has not run on this context");
    if (!host_dst || bytes < need || slot < 0 || slot >= c->slots)
        return sn::fail(c, SN_ERR_INVALID_ARG, "sn_debug_read_pool: bad slot or buffer too small (%zu needed)", need);
    SN_HIP(c, hipSetDevice(c->device));
    SN_HIP(c, hipStreamSynchronize(c->stream));
    SN_HIP(c, hipMemcpy(host_dst, c->pool.base + (int64_t)slot * c->pool.slot_bytes, need, hipMemcpyDeviceToHost));
    return SN_OK;
}

int sn_get_policy(sn_context* h, sn_policy* policy)
{
    Context* c = reinterpret_cast<Context*>(h);
    if (!c) return sn::fail(nullptr, SN_ERR_INVALID_ARG, "ctx is NULL");
    if (!policy || policy->struct_size != (int32_t)sizeof(sn_policy)) return sn::fail(c, SN_ERR_INVALID_ARG, "sn_policy.struct_size mismatch");
    *policy = c->policy;
    return SN_OK;
}

int sn_set_policy(sn_context* h, const sn_policy* policy)
{
    Context* c = reinterpret_cast<Context*>(h);
    if (!c) return sn::fail(nullptr, SN_ERR_INVALID_ARG, "ctx is NULL");
    if (!policy) return sn::fail(c, SN_ERR_INVALID_ARG, "policy is NULL");
    if (const char* t = policy_text(policy)) return sn::fail(c, SN_ERR_INVALID_ARG, "%s", t);
    c->policy.small_launches = policy->small_launches;
    c->policy.chain = policy->chain;
    c->policy.copy_threads = policy->copy_threads;  // (a copier that already runs keeps its threads)
    c->policy.chroma_sweeps = policy->chroma_sweeps;
    return SN_OK;
}

int sn_debug_raise_chain_fault(sn_context* h)
{
    Context* c = reinterpret_cast<Context*>(h);
    if (!c) return sn::fail(nullptr, SN_ERR_INVALID_ARG, "ctx is NULL");
    if (!c->chain_status) return sn::fail(c, SN_ERR_UNSUPPORTED, "sn_debug_raise_chain_fault: this context has not run a chain");
    c->chain_force_fault = true;
    return SN_OK;
}

int sn_debug_set_bands(sn_context* h, int32_t bands, int32_t warm_rows)
{
    Context* c = reinterpret_cast<Context*>(h);
    if (!c) return sn::fail(nullptr, SN_ERR_INVALID_ARG, "ctx is NULL");
    if (bands > kMaxBands || warm_rows < 0) return sn::fail(c, SN_ERR_INVALID_ARG, "sn_debug_set_bands: at most %d bands, warm_rows >= 0", kMaxBands);
    c->band_force = bands;
    c->band_warm = warm_rows;
    return SN_OK;
}

int sn_debug_read_coupled_rows(sn_context* h, int32_t which, void* host_dst, size_t bytes)
{
    Context* c = reinterpret_cast<Context*>(h);
    if (!c) return sn::fail(nullptr, SN_ERR_INVALID_ARG, "ctx is NULL");
    const size_t need = (size_t)sn::kBuffers * c->fpool_rows * c->cfg.width * c->cfg.bytes_per_sample;
    if (!c->fused420 || which < 0 || which > 1 || !c->fpool[which] || !host_dst || bytes < need)
        return sn::fail(c, SN_ERR_INVALID_ARG, "sn_debug_read_coupled_rows: no such hand-off or buffer too small (%zu needed)", need);
    SN_HIP(c, hipSetDevice(c->device));
    SN_HIP(c, hipStreamSynchronize(c->stream));
    std::vector<uint32_t> raw((size_t)c->fpool_frame_bytes / 4);
    SN_HIP(c, hipMemcpy(raw.data(), c->fpool[which], (size_t)c->fpool_frame_bytes, hipMemcpyDeviceToHost));
    if (c->cfg.bytes_per_sample == 4) sn::fused_f32_pool_unpack(raw.data(), c->cfg.width, c->fpool_rows, static_cast<float*>(host_dst));
    else if (c->cfg.bytes_per_sample == 2) sn::fused_u16_pool_unpack(raw.data(), c->cfg.width, c->fpool_rows, static_cast<uint16_t*>(host_dst));
    else sn::fused_v3_pool_unpack(raw.data(), c->cfg.width, c->fpool_rows, static_cast<uint8_t*>(host_dst));
    return SN_OK;
}

// ---- the anti-aliasing idiom as one call -------------------------------------------------------
struct sn_aa_context {
    sn_config cfg{};
    sn_context* first = nullptr;   // the turned clip
    sn_context* second = nullptr;  // the clip itself, on first's stream
    hipStream_t stream = nullptr;
    int planes = 1;
    int w[3] = {0, 0, 0}, h[3] = {0, 0, 0};  // clip geometry per plane
    int pitch[3] = {0, 0, 0}, tpitch[3] = {0, 0, 0};
    uint8_t* d_src[3] = {nullptr, nullptr, nullptr};  // clip geometry
    uint8_t* d_t1[3] = {nullptr, nullptr, nullptr};   // turned: input of the first pass
    uint8_t* d_u1[3] = {nullptr, nullptr, nullptr};   // turned: its output
    uint8_t* d_t2[3] = {nullptr, nullptr, nullptr};   // clip geometry: input of the second pass
    uint8_t* d_out[3] = {nullptr, nullptr, nullptr};
    uint8_t* h_src[3] = {nullptr, nullptr, nullptr};  // pinned staging for planes the caller holds in pageable memory
    uint8_t* h_dst[3] = {nullptr, nullptr, nullptr};
    std::string err;
};

static thread_local std::string g_aa_error;

const char* sn_aa_last_error(const sn_aa_context* a) { return a ? a->err.c_str() : g_aa_error.c_str(); }

void sn_aa_destroy(sn_aa_context* a)
{
    if (!a) return;
    (void)hipSetDevice(a->cfg.device);
    if (a->stream) (void)hipStreamSynchronize(a->stream);
    if (a->second) sn_destroy(a->second);
    if (a->first) sn_destroy(a->first);  // owns the stream
    for (int p = 0; p < 3; ++p)
        for (uint8_t* q : {a->d_src[p], a->d_t1[p], a->d_u1[p], a->d_t2[p], a->d_out[p]})
            if (q) (void)hipFree(q);
    for (int p = 0; p < 3; ++p)
        for (uint8_t* q : {a->h_src[p], a->h_dst[p]})
            if (q) (void)hipHostFree(q);
    delete a;
}

int sn_aa_create(const sn_config* cfg, sn_aa_context** out) { return sn_aa_create_with_policy(cfg, nullptr, out); }

int sn_aa_create_with_policy(const sn_config* cfg, const sn_policy* policy, sn_aa_context** out)
{
    auto fail_aa = [](sn_aa_context* a, int code, const std::string& msg) {
        g_aa_error = msg;
        if (a) sn_aa_destroy(a);
        return code;
    };
    if (!cfg || !out) return fail_aa(nullptr, SN_ERR_INVALID_ARG, "cfg / out is NULL");
    *out = nullptr;
    if (cfg->struct_size != (int32_t)sizeof(sn_config)) return fail_aa(nullptr, SN_ERR_INVALID_ARG, "sn_config.struct_size mismatch");
    if (cfg->dh) return fail_aa(nullptr, SN_ERR_UNSUPPORTED, "sn_aa_create: dh is not part of the anti-aliasing idiom");
    sn_aa_context* a = new (std::nothrow) sn_aa_context();
    if (!a) return fail_aa(nullptr, SN_ERR_INVALID_ARG, "out of host memory");
    a->cfg = *cfg;
    sn_config c1 = *cfg;  // TurnLeft: width <-> height, the chroma subsampling turns with it
    c1.width = cfg->height;
    c1.height = cfg->width;
    c1.sub_w = cfg->sub_h;
    c1.sub_h = cfg->sub_w;
    c1.max_batch = 1;
    c1.mode = SN_MODE_AUTO;
    c1.stream = nullptr;
    int rc = sn_create_with_policy(&c1, policy, &a->first);
    if (rc != SN_OK) return fail_aa(a, rc, sn_last_error(nullptr));
    sn_config c2 = *cfg;
    c2.max_batch = 1;
    c2.mode = SN_MODE_AUTO;
    c2.stream = sn_get_stream(a->first);
    rc = sn_create_with_policy(&c2, policy, &a->second);
    if (rc != SN_OK) return fail_aa(a, rc, sn_last_error(nullptr));
    a->stream = reinterpret_cast<hipStream_t>(sn_get_stream(a->first));
    a->planes = cfg->num_planes < 3 ? cfg->num_planes : 3;
    const int B = cfg->bytes_per_sample;
    for (int p = 0; p < a->planes; ++p) {
        a->w[p] = p ? cfg->width >> cfg->sub_w : cfg->width;
        a->h[p] = p ? cfg->height >> cfg->sub_h : cfg->height;
        a->pitch[p] = (a->w[p] * B + 255) & ~255;
        a->tpitch[p] = (a->h[p] * B + 255) & ~255;
        const size_t clip_bytes = (size_t)a->pitch[p] * a->h[p], turned_bytes = (size_t)a->tpitch[p] * a->w[p];
        for (uint8_t** q : {&a->d_src[p], &a->d_t2[p], &a->d_out[p]})
            if (hipMalloc(reinterpret_cast<void**>(q), clip_bytes) != hipSuccess) return fail_aa(a, SN_ERR_HIP, "hipMalloc failed (sn_aa_create)");
        for (uint8_t** q : {&a->h_src[p], &a->h_dst[p]})
            if (hipHostMalloc(reinterpret_cast<void**>(q), clip_bytes, hipHostMallocDefault) != hipSuccess) return fail_aa(a, SN_ERR_HIP, "hipHostMalloc failed (sn_aa_create)");
        for (uint8_t** q : {&a->d_t1[p], &a->d_u1[p]})
            if (hipMalloc(reinterpret_cast<void**>(q), turned_bytes) != hipSuccess) return fail_aa(a, SN_ERR_HIP, "hipMalloc failed (sn_aa_create)");
    }
    *out = a;
    return SN_OK;
}

int sn_aa_process_host(sn_aa_context* a, const void* const src[3], const int32_t sp[3], void* const dst[3], const int32_t dp[3],
                       int32_t parity)
{
    if (!a) return sn::fail(nullptr, SN_ERR_INVALID_ARG, "ctx is NULL");
    auto bad = [&](int code, const std::string& m) { a->err = m; return code; };
    if (!src || !sp || !dst || !dp) return bad(SN_ERR_INVALID_ARG, "plane array is NULL");
    const int B = a->cfg.bytes_per_sample;
    for (int p = 0; p < a->planes; ++p)
        if (!src[p] || !dst[p] || sp[p] < a->w[p] * B || dp[p] < a->w[p] * B) return bad(SN_ERR_INVALID_ARG, "plane pointer is NULL or pitch smaller than the row");
#define SN_AA_HIP(call)                                                                                      \
    do {                                                                                                     \
        hipError_t e_ = (call);                                                                              \
        if (e_ != hipSuccess) return bad(SN_ERR_HIP, std::string(#call " failed: ") + hipGetErrorString(e_)); \
    } while (0)
#define SN_AA_SN(ctx, call)                                                 \
    do {                                                                    \
        const int rc_ = (call);                                             \
        if (rc_ != SN_OK) return bad(rc_, sn_last_error(ctx));              \
    } while (0)
    SN_AA_HIP(hipSetDevice(a->cfg.device));
    SN_AA_SN(a->first, prepare_small_launch_scratch(reinterpret_cast<Context*>(a->first)));
    SN_AA_SN(a->second, prepare_small_launch_scratch(reinterpret_cast<Context*>(a->second)));
    // planes in pageable memory go through the context's pinned staging (the runtime never sees a pageable pointer, see
    // sn_process_host); planes inside memory the caller pinned are DMA'd as they lie
    for (int p = 0; p < a->planes; ++p) {
        const void* from = src[p];
        size_t from_pitch = (size_t)sp[p];
        if (!sn::plane_is_pinned(src[p], sp[p], a->w[p] * B, a->h[p])) {
            for (int y = 0; y < a->h[p]; ++y)
                memcpy(a->h_src[p] + (size_t)y * a->pitch[p], static_cast<const uint8_t*>(src[p]) + (size_t)y * sp[p], (size_t)a->w[p] * B);
            from = a->h_src[p];
            from_pitch = (size_t)a->pitch[p];
        }
        SN_AA_HIP(hipMemcpy2DAsync(a->d_src[p], a->pitch[p], from, from_pitch, (size_t)a->w[p] * B, a->h[p], hipMemcpyHostToDevice, a->stream));
    }
    for (int p = 0; p < a->planes; ++p)  // TurnLeft
        SN_AA_SN(a->first, sn_turn_device(a->first, -1, 1, a->d_src[p], 0, a->pitch[p], a->w[p], a->h[p], a->d_t1[p], 0, a->tpitch[p]));
    {
        const void* s3[3] = {a->d_t1[0], a->d_t1[1], a->d_t1[2]};
        void* d3[3] = {a->d_u1[0], a->d_u1[1], a->d_u1[2]};
        SN_AA_SN(a->first, sn_process_device(a->first, s3, a->tpitch, d3, a->tpitch, parity));
    }
    for (int p = 0; p < a->planes; ++p)  // TurnRight: the turned plane is h wide and w high
        SN_AA_SN(a->first, sn_turn_device(a->first, +1, 1, a->d_u1[p], 0, a->tpitch[p], a->h[p], a->w[p], a->d_t2[p], 0, a->pitch[p]));
    {
        const void* s3[3] = {a->d_t2[0], a->d_t2[1], a->d_t2[2]};
        void* d3[3] = {a->d_out[0], a->d_out[1], a->d_out[2]};
        SN_AA_SN(a->second, sn_process_device(a->second, s3, a->pitch, d3, a->pitch, parity));
    }
    bool staged[3] = {false, false, false};
    for (int p = 0; p < a->planes; ++p) {
        staged[p] = !sn::plane_is_pinned(dst[p], dp[p], a->w[p] * B, a->h[p]);
        SN_AA_HIP(hipMemcpy2DAsync(staged[p] ? a->h_dst[p] : dst[p], staged[p] ? (size_t)a->pitch[p] : (size_t)dp[p], a->d_out[p], a->pitch[p],
                                   (size_t)a->w[p] * B, a->h[p], hipMemcpyDeviceToHost, a->stream));
    }
    SN_AA_HIP(hipStreamSynchronize(a->stream));
    for (int p = 0; p < a->planes; ++p)
        if (staged[p])
            for (int y = 0; y < a->h[p]; ++y)
                memcpy(static_cast<uint8_t*>(dst[p]) + (size_t)y * dp[p], a->h_dst[p] + (size_t)y * a->pitch[p], (size_t)a->w[p] * B);
#undef SN_AA_HIP
#undef SN_AA_SN
    return SN_OK;
}

}  // extern "C"
#pragma GCC visibility pop

