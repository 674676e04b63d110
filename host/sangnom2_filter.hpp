// sangnom2_filter.hpp -- the SangNom2 filter object over the C ABI (include/sangnom_hip.h).
//
// Host-side mirror of the reference's class SangNom2 : GenericVideoFilter
// (/root/reference/src/SangNom2.h:40-67): same constructor arguments and defaults as
// Create_SangNom2 (src/SangNom2.cpp:399-435), same error text, GetFrame(n) with the same
// observable result (src/SangNom2.cpp:332-397).  All pixel work is done by libsangnom_hip.so:
// GetFrame hands the source planes to sn_process_host and receives the assembled output frame
// (kept field, border line, interpolated lines) in the destination planes.
//
// The class is a template over a host-traits type so that the same code serves the real AviSynth+
// SDK (host/sangnom2_avs_plugin.cpp) and this repository's test host (host/sn_host_api.h).
#pragma once

#include <algorithm>
#include <string>

#include "sangnom_hip.h"

namespace sangnom {

struct Args {  // SangNom2(clip, order, aa, aac, threads, dh, luma, chroma, opt)
    int order = 1;
    int aa = 48;
    int aac = 0;
    int threads = 0;  // dummy in the reference too (README.md:40-41)
    bool dh = false;
    bool luma = true;
    bool chroma = true;
    int opt = -1;     // the reference's CPU code-path switch; validated, otherwise unused
    int device = 0;   // HIP device ordinal (not a script argument)
};

template <class Host>
class Filter {
public:
    using Env = typename Host::Env;
    using ClipPtr = typename Host::ClipPtr;
    using FramePtr = typename Host::FramePtr;
    using Info = typename Host::Info;

    // Create_SangNom2's checks, in its order, with its text (src/SangNom2.cpp:407-422); `name`
    // is "SangNom2" or "SangNom" (the legacy wrapper reports under its own name, :446-459).
    Filter(ClipPtr child, const Args& a, Env* env, const char* name = "SangNom2") : child_(child), args_(a)
    {
        vi_ = Host::GetInfo(child);
        const std::string n(name);
        if (Host::IsRGB(vi_) || !Host::IsPlanar(vi_)) env->ThrowError("%s: clip must be in Y/YUV planar format.", name);
        if (Host::Height(vi_) % 2 != 0) env->ThrowError("%s: height must be even.", name);
        if (Host::Is420(vi_) && Host::Height(vi_) % 4) env->ThrowError("%s: height must be mod4.", name);
        if (a.order < 0 || a.order > 2) env->ThrowError("%s: order must be between 0..2.", name);
        if (a.aa < 0 || a.aa > 128) env->ThrowError("%s: aa must be between 0..128.", name);
        if (n == "SangNom2" && (a.aac < 0 || a.aac > 128)) env->ThrowError("%s: aac must be between 0..128.", name);
        if (a.opt < -1 || a.opt > 1) env->ThrowError("%s: opt must be between -1..2.", name);  // sic

        sn_config c{};
        c.struct_size = (int32_t)sizeof c;
        c.width = Host::Width(vi_);
        c.height = Host::Height(vi_);
        c.bytes_per_sample = Host::ComponentSize(vi_);
        c.bits_per_sample = Host::BitsPerComponent(vi_);
        c.num_planes = std::min(Host::NumComponents(vi_), 3);
        c.sub_w = c.num_planes > 1 ? Host::SubW(vi_) : 0;
        c.sub_h = c.num_planes > 1 ? Host::SubH(vi_) : 0;
        c.order = a.order;
        c.aa = a.aa;
        c.aac = a.aac;
        c.dh = a.dh;
        c.luma = a.luma;
        c.chroma = a.chroma;
        c.device = a.device;
        c.max_batch = 1;
        c.mode = SN_MODE_AUTO;
        const int rc = sn_create(&c, &ctx_);
        if (rc != SN_OK) env->ThrowError("%s: %s", name, sn_last_error(nullptr));
        if (a.dh) Host::SetHeight(vi_, Host::Height(vi_) * 2);  // src/SangNom2.cpp:284-285
        planes_ = c.num_planes;
    }
    Filter(const Filter&) = delete;
    Filter& operator=(const Filter&) = delete;
    ~Filter() { sn_destroy(ctx_); }

    const Info& GetInfo() const { return vi_; }

    // SangNom2::GetFrame, src/SangNom2.cpp:332-397.
    FramePtr GetFrame(int n, Env* env)
    {
        FramePtr src = Host::GetFrame(child_, n, env);
        FramePtr dst = Host::NewFrame(env, vi_, src);
        const void* sp[3] = {nullptr, nullptr, nullptr};
        void* dp[3] = {nullptr, nullptr, nullptr};
        int32_t spitch[3] = {0, 0, 0}, dpitch[3] = {0, 0, 0};
        for (int p = 0; p < planes_; ++p) {
            sp[p] = Host::ReadPtr(src, p);
            dp[p] = Host::WritePtr(dst, p);
            spitch[p] = Host::Pitch(src, p);
            dpitch[p] = Host::Pitch(dst, p);
        }
        const int parity = args_.order == 0 ? (Host::GetParity(child_, n) ? 1 : 0) : 1;
        if (sn_process_host(ctx_, sp, spitch, dp, dpitch, parity) != SN_OK)
            env->ThrowError("SangNom2: %s", sn_last_error(ctx_));
        return dst;
    }

    // One instance per host thread, like the reference (src/SangNom2.h:63-66): a context owns its
    // stream, device pool and staging buffers.
    static constexpr bool kMultiInstance = true;

private:
    ClipPtr child_;
    Args args_;
    Info vi_{};
    sn_context* ctx_ = nullptr;
    int planes_ = 1;
};

// Legacy SangNom(clip, order, aa, opt): order 0/1/2 = bottom/top/double-rate is remapped to
// SangNom2's 2/1/0 (src/SangNom2.cpp:441,463).  The reference additionally reads arguments its
// signature does not have (args[3] lands in aac, :443); that quirk is not reproduced: aac = 0.
inline Args legacy_args(int order, int aa, int opt)
{
    Args a;
    static const int ord[3] = {2, 1, 0};
    a.order = (order >= 0 && order <= 2) ? ord[order] : order;
    a.aa = aa;
    a.opt = opt;
    return a;
}

}  // namespace sangnom
