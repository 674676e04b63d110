// sangnom2_vs_plugin.cpp -- a VapourSynth (API 4) front-end over libsangnom_hip.so.
//
// The reference is itself a port of the VapourSynth plugin (README.md:5 of the reference); this translation unit
// offers the same filter to VapourSynth scripts through the C ABI of include/sangnom_hip.h:
//     core.sangnomhip.SangNom(clip, order=1, dh=False, aa=48, aac=0, luma=True, chroma=True,
//                             isolated=False, fresh=False)
// It needs the VapourSynth SDK header VapourSynth4.h, a third-party file that is not part of this repository or
// of the build image: without it this file compiles to nothing, so it has NOT been compiled or run here
// (INTEGRATION.md says so).  With the SDK:  make -C host VS_INCLUDE_DIR=<dir of VapourSynth4.h>.
#if defined(__has_include)
#if __has_include(<VapourSynth4.h>)
#define SN_HAVE_VAPOURSYNTH 1
#endif
#endif

#ifdef SN_HAVE_VAPOURSYNTH
#include <VapourSynth4.h>

#include <mutex>
#include <string>

#include "sangnom_hip.h"

namespace {

struct SangNomData {
    VSNode* node = nullptr;
    VSVideoInfo vi{};
    sn_context* ctx = nullptr;
    int planes = 1;
    int order = 1;
    std::mutex mtx;  // one context == one filter instance: frames go through it one at a time, in request order
};

const VSFrame* VS_CC sangnomGetFrame(int n, int activationReason, void* instanceData, void**, VSFrameContext* frameCtx, VSCore* core,
                                     const VSAPI* vsapi)
{
    SangNomData* d = static_cast<SangNomData*>(instanceData);
    if (activationReason == arInitial) {
        vsapi->requestFrameFilter(n, d->node, frameCtx);
        return nullptr;
    }
    if (activationReason != arAllFramesReady) return nullptr;
    const VSFrame* src = vsapi->getFrameFilter(n, d->node, frameCtx);
    VSFrame* dst = vsapi->newVideoFrame(&d->vi.format, d->vi.width, d->vi.height, src, core);
    const void* sp[3] = {nullptr, nullptr, nullptr};
    void* dp[3] = {nullptr, nullptr, nullptr};
    int32_t spitch[3] = {0, 0, 0}, dpitch[3] = {0, 0, 0};
    for (int p = 0; p < d->planes; ++p) {
        sp[p] = vsapi->getReadPtr(src, p);
        dp[p] = vsapi->getWritePtr(dst, p);
        spitch[p] = (int32_t)vsapi->getStride(src, p);
        dpitch[p] = (int32_t)vsapi->getStride(dst, p);
    }
    int parity = 1;  // order == 0: the field to keep follows the frame's _Field / _FieldBased property (1 = top)
    if (d->order == 0) {
        int err = 0;
        const VSMap* props = vsapi->getFramePropertiesRO(src);
        int64_t field = vsapi->mapGetInt(props, "_Field", 0, &err);
        if (err) field = vsapi->mapGetInt(props, "_FieldBased", 0, &err) == 1 ? 0 : 1;
        parity = field ? 1 : 0;
    }
    int rc;
    {
        std::lock_guard<std::mutex> lk(d->mtx);
        rc = sn_process_host(d->ctx, sp, spitch, dp, dpitch, parity);
        if (rc != SN_OK) vsapi->setFilterError((std::string("SangNom: ") + sn_last_error(d->ctx)).c_str(), frameCtx);
    }
    vsapi->freeFrame(src);
    if (rc != SN_OK) {
        vsapi->freeFrame(dst);
        return nullptr;
    }
    return dst;
}

void VS_CC sangnomFree(void* instanceData, VSCore*, const VSAPI* vsapi)
{
    SangNomData* d = static_cast<SangNomData*>(instanceData);
    vsapi->freeNode(d->node);
    sn_destroy(d->ctx);
    delete d;
}

void VS_CC sangnomCreate(const VSMap* in, VSMap* out, void*, VSCore* core, const VSAPI* vsapi)
{
    SangNomData* d = new SangNomData;
    int err = 0;
    d->node = vsapi->mapGetNode(in, "clip", 0, &err);
    d->vi = *vsapi->getVideoInfo(d->node);
    auto geti = [&](const char* key, int def) {
        int e = 0;
        const int64_t v = vsapi->mapGetInt(in, key, 0, &e);
        return e ? def : (int)v;
    };
    auto fail = [&](const std::string& msg) {
        vsapi->mapSetError(out, ("SangNom: " + msg).c_str());
        vsapi->freeNode(d->node);
        delete d;
    };
    const VSVideoFormat& f = d->vi.format;
    if (f.colorFamily != cfGray && f.colorFamily != cfYUV) return fail("clip must be in Y/YUV planar format.");
    sn_config c{};
    c.struct_size = (int32_t)sizeof c;
    c.width = d->vi.width;
    c.height = d->vi.height;
    c.bytes_per_sample = f.bytesPerSample;
    c.bits_per_sample = f.bitsPerSample;
    c.num_planes = f.numPlanes < 3 ? 1 : 3;
    c.sub_w = c.num_planes > 1 ? f.subSamplingW : 0;
    c.sub_h = c.num_planes > 1 ? f.subSamplingH : 0;
    c.order = d->order = geti("order", 1);
    c.aa = geti("aa", 48);
    c.aac = geti("aac", 0);
    c.dh = geti("dh", 0) != 0;
    c.luma = geti("luma", 1) != 0;
    c.chroma = geti("chroma", 1) != 0;
    c.isolated_planes = geti("isolated", 0) != 0;
    c.fresh_pool = geti("fresh", 0) != 0;
    c.max_batch = 1;
    c.mode = SN_MODE_AUTO;
    char msg[256];
    if (sn_validate(&c, msg, sizeof msg) != SN_OK) return fail(msg + (std::string(msg).rfind("SangNom2: ", 0) == 0 ? 10 : 0));
    if (sn_create(&c, &d->ctx) != SN_OK) return fail(sn_last_error(nullptr));
    d->planes = c.num_planes;
    if (c.dh) d->vi.height *= 2;
    // The context keeps the reference's per-instance state (scratch pool, frame order), so frames are served one at
    // a time in request order: fmFrameState.  With fresh=True every frame is independent, but the context is still
    // not re-entrant (one stream, one staging area), hence the mutex above rather than a parallel mode.
    VSFilterDependency deps[] = {{d->node, rpStrictSpatial}};
    vsapi->createVideoFilter(out, "SangNom", &d->vi, sangnomGetFrame, sangnomFree, fmFrameState, deps, 1, d, core);
}

}  // namespace

VS_EXTERNAL_API(void) VapourSynthPluginInit2(VSPlugin* plugin, const VSPLUGINAPI* vspapi)
{
    vspapi->configPlugin("com.github.sangnom.hip", "sangnomhip", "SangNom2 edge-directed interpolation on AMD GPUs (libsangnom_hip)",
                         VS_MAKE_VERSION(1, 0), VAPOURSYNTH_API_VERSION, 0, plugin);
    vspapi->registerFunction("SangNom",
                             "clip:vnode;order:int:opt;dh:int:opt;aa:int:opt;aac:int:opt;luma:int:opt;chroma:int:opt;isolated:int:opt;fresh:int:opt;",
                             "clip:vnode;", sangnomCreate, nullptr, plugin);
}
#endif  // SN_HAVE_VAPOURSYNTH
