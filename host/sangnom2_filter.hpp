// sangnom2_filter.hpp -- the SangNom2 filter object over the C ABI (include/sangnom_hip.h).
//
// Host-side mirror of the reference's class SangNom2 : GenericVideoFilter
// (/root/reference/src/SangNom2.h:40-67): same constructor arguments and defaults as
// Create_SangNom2 (src/SangNom2.cpp:399-435), same error text, GetFrame(n) with the same
// observable result (src/SangNom2.cpp:332-397).  All pixel work is done by libsangnom_hip.so:
// GetFrame hands the source planes to sn_process_host and receives the assembled output frame
// (kept field, border line, interpolated lines) in the destination planes.
//
// With Args::lookahead > 1 GetFrame(n) keeps frames n .. n+lookahead-1 in flight on the library's host ring
// (sn_submit_host / sn_collect_host): it requests the child's next frames ahead of the
// caller, so their transfers and sweeps overlap; a request out of sequence drains the ring and restarts there.
//
// The class is a template over a host-traits type so that the same code serves the real AviSynth+
// SDK (host/sangnom2_avs_plugin.cpp) and this repository's test host (host/sn_host_api.h).
#pragma once

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <deque>
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
    int lookahead = -1;  // frames in flight behind GetFrame; -1: $SANGNOM_LOOKAHEAD or 1 (synchronous)
    bool isolated = false;  // extension: every plane filtered as a Y clip of its own (sn_config.isolated_planes)
    bool fresh = false;     // extension: ... and every frame by a new instance (sn_config.fresh_pool)
    sn_policy policy{};     // scheduling only (sangnom_hip.h); zeros = the defaults
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
        if (n != "SangNom" && (a.aac < 0 || a.aac > 128)) env->ThrowError("%s: aac must be between 0..128.", name);
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
        int la = a.lookahead;
        if (la < 0) {
            const char* e = std::getenv("SANGNOM_LOOKAHEAD");
            la = e ? std::atoi(e) : 1;
        }
        c.host_depth = std::max(1, std::min(la, 256));
        c.isolated_planes = a.isolated ? 1 : 0;
        c.fresh_pool = a.fresh ? 1 : 0;
        sn_policy pol = a.policy;
        pol.struct_size = (int32_t)sizeof pol;
        const int rc = sn_create_with_policy(&c, &pol, &ctx_);
        if (rc != SN_OK) env->ThrowError("%s: %s", name, sn_last_error(nullptr));
        if (a.dh) Host::SetHeight(vi_, Host::Height(vi_) * 2);  // src/SangNom2.cpp:284-285
        planes_ = c.num_planes;
        alpha_ = Host::NumComponents(vi_) == 4;
        num_frames_ = Host::NumFrames(vi_);
        // look-ahead processes frames the caller may never ask for; where results depend on the order of
        // processing (history-carrying configurations) GetFrame therefore stays synchronous
        sn_info info{};
        info.struct_size = (int32_t)sizeof info;
        const bool ahead = c.host_depth > 1 && sn_get_info(ctx_, &info) == SN_OK && info.history_free;
        slots_ = ahead ? sn_host_slots(ctx_) : 1;
    }
    Filter(const Filter&) = delete;
    Filter& operator=(const Filter&) = delete;
    ~Filter()
    {
        inflight_.clear();  // frames still in the ring are dropped with the context
        sn_destroy(ctx_);
    }

    const Info& GetInfo() const { return vi_; }

    // SangNom2::GetFrame, src/SangNom2.cpp:332-397.
    FramePtr GetFrame(int n, Env* env)
    {
        if (slots_ > 1) return GetFrameAhead(n, env);
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
        CopyAlpha(src, dst);
        return dst;
    }

    // One instance per host thread, like the reference (src/SangNom2.h:63-66): a context owns its
    // stream, device pool and staging buffers.
    static constexpr bool kMultiInstance = true;

private:
    struct Pending {
        int n;
        FramePtr src, dst;  // src is held until its planes have been staged (submit copies them at once)
        int32_t slot;
    };

    void Planes(const FramePtr& src, const FramePtr& dst, const void* sp[3], int32_t spitch[3], void* dp[3], int32_t dpitch[3])
    {
        for (int p = 0; p < planes_; ++p) {
            if (src) { sp[p] = Host::ReadPtr(src, p); spitch[p] = Host::Pitch(src, p); }
            if (dst) { dp[p] = Host::WritePtr(dst, p); dpitch[p] = Host::Pitch(dst, p); }
        }
    }

    // EXTENSION: the reference never writes a fourth plane (src/SangNom2.cpp:346-348 stops at
    // min(NumComponents, 3)), so the alpha of a YUVA clip comes out as whatever the new frame held.  Here it is passed
    // through: copied line by line, and with dh every source line fills two output lines.
    void CopyAlpha(const FramePtr& src, const FramePtr& dst) const
    {
        if (!alpha_) return;
        const uint8_t* s = Host::ReadPtr(src, 3);
        uint8_t* d = Host::WritePtr(dst, 3);
        const int sp = Host::Pitch(src, 3), dp = Host::Pitch(dst, 3);
        const size_t row = (size_t)Host::Width(vi_) * Host::ComponentSize(vi_);
        const int h_out = Host::Height(vi_);
        for (int y = 0; y < h_out; ++y) std::memcpy(d + (size_t)y * dp, s + (size_t)(args_.dh ? y / 2 : y) * sp, row);
    }

    FramePtr Collect(Env* env)
    {
        Pending p = inflight_.front();
        inflight_.pop_front();
        const void* sp[3] = {nullptr, nullptr, nullptr};
        void* dp[3] = {nullptr, nullptr, nullptr};
        int32_t spitch[3] = {0, 0, 0}, dpitch[3] = {0, 0, 0};
        Planes(FramePtr(), p.dst, sp, spitch, dp, dpitch);
        if (sn_collect_host(ctx_, p.slot, dp, dpitch) != SN_OK) env->ThrowError("SangNom2: %s", sn_last_error(ctx_));
        return p.dst;
    }

    FramePtr GetFrameAhead(int n, Env* env)
    {
        if (inflight_.empty() || inflight_.front().n != n) {  // not the frame at the head: start over at n
            while (!inflight_.empty()) Collect(env);
            next_ = n;
        }
        while ((int)inflight_.size() < slots_ && next_ < num_frames_) {
            Pending p;
            p.n = next_++;
            p.src = Host::GetFrame(child_, p.n, env);
            p.dst = Host::NewFrame(env, vi_, p.src);
            const void* sp[3] = {nullptr, nullptr, nullptr};
            void* dp[3] = {nullptr, nullptr, nullptr};
            int32_t spitch[3] = {0, 0, 0}, dpitch[3] = {0, 0, 0};
            Planes(p.src, FramePtr(), sp, spitch, dp, dpitch);
            const int parity = args_.order == 0 ? (Host::GetParity(child_, p.n) ? 1 : 0) : 1;
            if (sn_submit_host(ctx_, sp, spitch, parity, &p.slot) != SN_OK) env->ThrowError("SangNom2: %s", sn_last_error(ctx_));
            CopyAlpha(p.src, p.dst);
            // p.src stays referenced until the slot is collected: a source plane that lies in pinned memory is read by
            // the device straight from the host's frame, asynchronously (sn_submit_host, sangnom_hip.h)
            inflight_.push_back(p);
        }
        if (inflight_.empty()) env->ThrowError("SangNom2: frame %d is outside the clip", n);
        return Collect(env);
    }

    std::deque<Pending> inflight_;
    int next_ = 0;
    int num_frames_ = 0;
    int slots_ = 1;
    ClipPtr child_;
    Args args_;
    Info vi_{};
    sn_context* ctx_ = nullptr;
    int planes_ = 1;
    bool alpha_ = false;
};

// SangNomAA(clip, order, aa, aac): the anti-aliasing idiom TurnLeft().SangNom2(order, aa, aac).TurnRight().SangNom2(order,
// aa, aac) as one filter over sn_aa_process_host -- the frame crosses PCIe once each way instead of four times
// (README.md:3 of the reference: "mainly used in anti-aliasing scripts").  Same checks and messages
// as SangNom2 for the clip itself; the turned clip must pass them as well (e.g. an even WIDTH).
template <class Host>
class AAFilter {
public:
    using Env = typename Host::Env;
    using ClipPtr = typename Host::ClipPtr;
    using FramePtr = typename Host::FramePtr;
    using Info = typename Host::Info;

    AAFilter(ClipPtr child, const Args& a, Env* env, const char* name = "SangNomAA") : child_(child), args_(a)
    {
        vi_ = Host::GetInfo(child);
        if (Host::IsRGB(vi_) || !Host::IsPlanar(vi_)) env->ThrowError("%s: clip must be in Y/YUV planar format.", name);
        if (Host::Height(vi_) % 2 != 0) env->ThrowError("%s: height must be even.", name);
        if (Host::Is420(vi_) && Host::Height(vi_) % 4) env->ThrowError("%s: height must be mod4.", name);
        if (a.order < 0 || a.order > 2) env->ThrowError("%s: order must be between 0..2.", name);
        if (a.aa < 0 || a.aa > 128) env->ThrowError("%s: aa must be between 0..128.", name);
        if (a.aac < 0 || a.aac > 128) env->ThrowError("%s: aac must be between 0..128.", name);
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
        c.luma = a.luma;
        c.chroma = a.chroma;
        c.device = a.device;
        c.max_batch = 1;
        c.isolated_planes = a.isolated ? 1 : 0;
        c.fresh_pool = a.fresh ? 1 : 0;
        sn_policy pol = a.policy;
        pol.struct_size = (int32_t)sizeof pol;
        if (sn_aa_create_with_policy(&c, &pol, &ctx_) != SN_OK) env->ThrowError("%s: %s", name, sn_aa_last_error(nullptr));
        planes_ = c.num_planes;
        alpha_ = Host::NumComponents(vi_) == 4;
    }
    AAFilter(const AAFilter&) = delete;
    AAFilter& operator=(const AAFilter&) = delete;
    ~AAFilter() { sn_aa_destroy(ctx_); }

    const Info& GetInfo() const { return vi_; }

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
        if (sn_aa_process_host(ctx_, sp, spitch, dp, dpitch, parity) != SN_OK) env->ThrowError("SangNomAA: %s", sn_aa_last_error(ctx_));
        if (alpha_) {  // passed through (see Filter::CopyAlpha)
            const size_t row = (size_t)Host::Width(vi_) * Host::ComponentSize(vi_);
            for (int y = 0; y < Host::Height(vi_); ++y)
                std::memcpy(Host::WritePtr(dst, 3) + (size_t)y * Host::Pitch(dst, 3), Host::ReadPtr(src, 3) + (size_t)y * Host::Pitch(src, 3), row);
        }
        return dst;
    }

private:
    ClipPtr child_;
    Args args_;
    Info vi_{};
    sn_aa_context* ctx_ = nullptr;
    int planes_ = 1;
    bool alpha_ = false;
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
