// sn_host_api.h -- the host-side call shape the SangNom2 filter adapter is written against.
//
// This is NOT avisynth.h and does not try to be: it is this repository's own minimal test host
// (namespace snhost, own type names) with just the operations SangNom2::GetFrame uses
// (/root/reference/src/SangNom2.cpp:332-397: child->GetFrame / GetParity, env->NewVideoFrame,
// per-plane pointer / pitch / row size / height, ThrowError).  host/sangnom2_filter.hpp is a
// template over a "host traits" type; host/sangnom2_avs_plugin.cpp instantiates it with the real
// AviSynth+ SDK types when <avisynth.h> is available, tests instantiate it with these.
#pragma once

#include <cstdint>
#include <cstdio>
#include <cstdarg>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace snhost {

struct Error : std::runtime_error {  // what env->ThrowError raises
    using std::runtime_error::runtime_error;
};

struct ClipInfo {  // the VideoInfo fields the filter reads (SangNom2.cpp:281-288,407-411)
    int width = 0, height = 0;
    int component_size = 1;    // bytes per sample
    int bits_per_component = 8;
    int num_components = 1;    // 1 = Y, 3 = YUV, 4 = YUVA (alpha: luma-sized)
    int sub_w = 0, sub_h = 0;  // log2 chroma subsampling
    int num_frames = 0;
    bool rgb = false, planar = true;
    bool Is420() const { return num_components >= 3 && sub_w == 1 && sub_h == 1; }
    int PlaneWidth(int p) const { return (p == 0 || p == 3) ? width : width >> sub_w; }
    int PlaneHeight(int p) const { return (p == 0 || p == 3) ? height : height >> sub_h; }
};

class Frame {  // a pitched planar frame
public:
    Frame(const ClipInfo& vi, int align = 32) : vi_(vi)
    {
        for (int p = 0; p < vi.num_components && p < 4; ++p) {
            const int row = vi.PlaneWidth(p) * vi.component_size;
            pitch_[p] = (row + align - 1) / align * align + align;  // deliberately wider than the row
            data_[p].assign((size_t)pitch_[p] * vi.PlaneHeight(p), 0xEE);
        }
    }
    uint8_t* Ptr(int p) { return data_[p].data(); }
    const uint8_t* Ptr(int p) const { return data_[p].data(); }
    int Pitch(int p) const { return pitch_[p]; }
    int RowSize(int p) const { return vi_.PlaneWidth(p) * vi_.component_size; }
    int Height(int p) const { return vi_.PlaneHeight(p); }
    const ClipInfo& Info() const { return vi_; }

private:
    ClipInfo vi_;
    std::vector<uint8_t> data_[4];
    int pitch_[4] = {0, 0, 0, 0};
};
using FramePtr = std::shared_ptr<Frame>;

class Clip {  // IClip
public:
    virtual ~Clip() = default;
    virtual FramePtr GetFrame(int n) = 0;
    virtual bool GetParity(int n) = 0;
    virtual const ClipInfo& GetInfo() const = 0;
};
using ClipPtr = std::shared_ptr<Clip>;

class Env {  // IScriptEnvironment
public:
    FramePtr NewVideoFrame(const ClipInfo& vi) { return std::make_shared<Frame>(vi); }
    [[noreturn]] void ThrowError(const char* fmt, ...)
    {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        throw Error(buf);
    }
};

// Traits that present this host to sangnom2_filter.hpp.
struct TestHost {
    using Env = snhost::Env;
    using ClipPtr = snhost::ClipPtr;
    using FramePtr = snhost::FramePtr;
    using Info = snhost::ClipInfo;
    static Info GetInfo(const ClipPtr& c) { return c->GetInfo(); }
    static int Width(const Info& v) { return v.width; }
    static int Height(const Info& v) { return v.height; }
    static void SetHeight(Info& v, int h) { v.height = h; }
    static int ComponentSize(const Info& v) { return v.component_size; }
    static int BitsPerComponent(const Info& v) { return v.bits_per_component; }
    static int NumComponents(const Info& v) { return v.num_components; }
    static int SubW(const Info& v) { return v.sub_w; }
    static int SubH(const Info& v) { return v.sub_h; }
    static int NumFrames(const Info& v) { return v.num_frames; }
    static bool IsRGB(const Info& v) { return v.rgb; }
    static bool IsPlanar(const Info& v) { return v.planar; }
    static bool Is420(const Info& v) { return v.Is420(); }
    static FramePtr GetFrame(const ClipPtr& c, int n, Env*) { return c->GetFrame(n); }
    static bool GetParity(const ClipPtr& c, int n) { return c->GetParity(n); }
    static FramePtr NewFrame(Env* e, const Info& v, const FramePtr&) { return e->NewVideoFrame(v); }
    static const uint8_t* ReadPtr(const FramePtr& f, int plane) { return f->Ptr(plane); }
    static uint8_t* WritePtr(const FramePtr& f, int plane) { return f->Ptr(plane); }
    static int Pitch(const FramePtr& f, int plane) { return f->Pitch(plane); }
};

}  // namespace snhost
