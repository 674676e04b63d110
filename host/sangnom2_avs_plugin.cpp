// sangnom2_avs_plugin.cpp -- the AviSynth+ plugin surface over libsangnom_hip.so.
//
// Drop-in for the reference's plugin (same exported symbol, same function names and parameter
// strings, /root/reference/src/SangNom2.cpp:474-484).  It needs the AviSynth+ SDK header, which
// is a third-party file that is not part of this repository or of the build image: without it this
// translation unit compiles to nothing (the filter logic it wraps, host/sangnom2_filter.hpp, is
// built and tested against host/sn_host_api.h instead).
//   make -C host AVS_INCLUDE_DIR=/usr/local/include/avisynth   ->   host/libsangnom2_hip_avs.so
#if defined(__has_include)
#if __has_include(<avisynth.h>)
#define SN_HAVE_AVISYNTH 1
#endif
#endif

#ifdef SN_HAVE_AVISYNTH
#include <avisynth.h>

#include "sangnom2_filter.hpp"

namespace {

struct AvsHost {
    using Env = IScriptEnvironment;
    using ClipPtr = PClip;
    using FramePtr = PVideoFrame;
    using Info = VideoInfo;
    static Info GetInfo(const ClipPtr& c) { return c->GetVideoInfo(); }
    static int Width(const Info& v) { return v.width; }
    static int Height(const Info& v) { return v.height; }
    static void SetHeight(Info& v, int h) { v.height = h; }
    static int ComponentSize(const Info& v) { return v.ComponentSize(); }
    static int BitsPerComponent(const Info& v) { return v.BitsPerComponent(); }
    static int NumComponents(const Info& v) { return v.NumComponents(); }
    static int SubW(const Info& v) { return v.GetPlaneWidthSubsampling(PLANAR_U); }
    static int SubH(const Info& v) { return v.GetPlaneHeightSubsampling(PLANAR_U); }
    static int NumFrames(const Info& v) { return v.num_frames; }
    static bool IsRGB(const Info& v) { return v.IsRGB(); }
    static bool IsPlanar(const Info& v) { return v.IsPlanar(); }
    static bool Is420(const Info& v) { return v.Is420(); }
    static FramePtr GetFrame(const ClipPtr& c, int n, Env* e) { return c->GetFrame(n, e); }
    static bool GetParity(const ClipPtr& c, int n) { return c->GetParity(n); }
    static FramePtr NewFrame(Env* e, const Info& v, const FramePtr& src)
    {
        return e->FunctionExists("propShow") ? e->NewVideoFrameP(v, const_cast<FramePtr*>(&src), 32)
                                             : e->NewVideoFrame(v);  // src/SangNom2.cpp:344
    }
    static int plane_id(int p) { static const int ids[4] = {PLANAR_Y, PLANAR_U, PLANAR_V, PLANAR_A}; return ids[p]; }
    static const uint8_t* ReadPtr(const FramePtr& f, int p) { return f->GetReadPtr(plane_id(p)); }
    static uint8_t* WritePtr(const FramePtr& f, int p) { return f->GetWritePtr(plane_id(p)); }
    static int Pitch(const FramePtr& f, int p) { return f->GetPitch(plane_id(p)); }
};

class SangNom2 : public GenericVideoFilter {
    sangnom::Filter<AvsHost> impl_;

public:
    SangNom2(PClip child, const sangnom::Args& a, IScriptEnvironment* env, const char* name)
        : GenericVideoFilter(child), impl_(child, a, env, name)
    {
        vi = impl_.GetInfo();
    }
    PVideoFrame __stdcall GetFrame(int n, IScriptEnvironment* env) override { return impl_.GetFrame(n, env); }
    int __stdcall SetCacheHints(int cachehints, int) override
    {
        return cachehints == CACHE_GET_MTMODE ? MT_MULTI_INSTANCE : 0;  // src/SangNom2.h:63-66
    }
};

static sangnom::Args reference_args(const AVSValue& args)  // Create_SangNom2's defaults, src/SangNom2.cpp:402-405,429-432
{
    sangnom::Args a;
    a.order = args[1].AsInt(1);
    a.aa = args[2].AsInt(48);
    a.aac = args[3].AsInt(0);
    a.threads = args[4].AsInt(0);
    a.dh = args[5].AsBool(false);
    a.luma = args[6].AsBool(true);
    a.chroma = args[7].AsBool(true);
    a.opt = args[8].AsInt(-1);
    return a;
}

AVSValue __cdecl Create_SangNom2(AVSValue args, void*, IScriptEnvironment* env)
{
    return new SangNom2(args[0].AsClip(), reference_args(args), env, "SangNom2");
}

// SangNom2HIP: SangNom2's arguments plus this implementation's extensions (include/sangnom_hip.h: isolated_planes,
// fresh_pool; look-ahead depth of GetFrame; HIP device).  A name of its own, so that "SangNom2" keeps the reference's
// signature byte for byte.
AVSValue __cdecl Create_SangNom2HIP(AVSValue args, void*, IScriptEnvironment* env)
{
    sangnom::Args a = reference_args(args);
    a.isolated = args[9].AsBool(false);
    a.fresh = args[10].AsBool(false);
    a.lookahead = args[11].AsInt(-1);
    a.device = args[12].AsInt(0);
    return new SangNom2(args[0].AsClip(), a, env, "SangNom2HIP");
}

// SangNomAA(clip, order, aa, aac): TurnLeft().SangNom2(order, aa, aac).TurnRight().SangNom2(order, aa, aac) with the frame
// staying on the device between the passes (sangnom::AAFilter).
class SangNomAA : public GenericVideoFilter {
    sangnom::AAFilter<AvsHost> impl_;

public:
    SangNomAA(PClip child, const sangnom::Args& a, IScriptEnvironment* env) : GenericVideoFilter(child), impl_(child, a, env, "SangNomAA") {}
    PVideoFrame __stdcall GetFrame(int n, IScriptEnvironment* env) override { return impl_.GetFrame(n, env); }
    int __stdcall SetCacheHints(int cachehints, int) override { return cachehints == CACHE_GET_MTMODE ? MT_MULTI_INSTANCE : 0; }
};

AVSValue __cdecl Create_SangNomAA(AVSValue args, void*, IScriptEnvironment* env)
{
    sangnom::Args a;
    a.order = args[1].AsInt(1);
    a.aa = args[2].AsInt(48);
    a.aac = args[3].AsInt(0);
    a.device = args[4].AsInt(0);
    return new SangNomAA(args[0].AsClip(), a, env);
}

AVSValue __cdecl Create_SangNom(AVSValue args, void*, IScriptEnvironment* env)
{
    const int order = args[1].AsInt(1);
    if (order < 0 || order > 2) env->ThrowError("SangNom: order must be between 0..2.");
    return new SangNom2(args[0].AsClip(), sangnom::legacy_args(order, args[2].AsInt(48), args[3].AsInt(-1)), env,
                        "SangNom");
}

}  // namespace

const AVS_Linkage* AVS_linkage;

extern "C" __declspec(dllexport) const char* __stdcall AvisynthPluginInit3(IScriptEnvironment* env,
                                                                           const AVS_Linkage* const vectors)
{
    AVS_linkage = vectors;
    env->AddFunction("SangNom2", "c[order]i[aa]i[aac]i[threads]i[dh]b[luma]b[chroma]b[opt]i", Create_SangNom2, 0);  // src/SangNom2.cpp:481, byte for byte
    env->AddFunction("SangNom", "c[order]i[aa]i[opt]i", Create_SangNom, 0);                                        // :482
    env->AddFunction("SangNom2HIP", "c[order]i[aa]i[aac]i[threads]i[dh]b[luma]b[chroma]b[opt]i[isolated]b[fresh]b[lookahead]i[device]i",
                     Create_SangNom2HIP, 0);
    env->AddFunction("SangNomAA", "c[order]i[aa]i[aac]i[device]i", Create_SangNomAA, 0);
    return "SangNom2";
}
#endif  // SN_HAVE_AVISYNTH
