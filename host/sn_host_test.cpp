// sn_host_test.cpp -- drives host/sangnom2_filter.hpp through the test host, the way a script
// engine drives the reference plugin: build a source clip, construct SangNom2(clip, ...), request
// frames with GetFrame(n).  tests/test_host_adapter.py feeds it frames and compares the output with
// the oracle.
//   sn_host_test <in.bin> <out.bin> [lookahead | aa [first-frame-order...]]
// lookahead > 1 runs GetFrame over the host ring; "aa" constructs SangNomAA (sangnom::AAFilter) instead of SangNom2;
// the optional list gives the order in which frames are requested (default 0 .. nframes-1), e.g. to exercise a seek.
// planes = 4 in the header is a YUVA clip: the fourth plane is luma-sized and passed through.
// in.bin : 14 x int32 {w,h,bytes,bits,planes,subw,subh,order,aa,aac,dh,luma,chroma,nframes}, then per
//          frame: int32 parity + the planes, tightly packed.
// out.bin: per frame the output planes, tightly packed.  On a constructor error: exit code 3 and
//          the message on stdout.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "sangnom2_filter.hpp"
#include "sn_host_api.h"

using namespace snhost;

class MemoryClip : public Clip {
public:
    ClipInfo vi;
    std::vector<std::vector<uint8_t>> frames;  // tight planes, concatenated
    std::vector<int> parity;
    FramePtr GetFrame(int n) override
    {
        auto f = std::make_shared<Frame>(vi, 64);
        const uint8_t* s = frames[n].data();
        for (int p = 0; p < vi.num_components; ++p) {
            const int row = f->RowSize(p);
            for (int y = 0; y < f->Height(p); ++y, s += row) memcpy(f->Ptr(p) + (size_t)y * f->Pitch(p), s, row);
        }
        return f;
    }
    bool GetParity(int n) override { return parity[n] != 0; }
    const ClipInfo& GetInfo() const override { return vi; }
};

int main(int argc, char** argv)
{
    if (argc < 3) return 2;
    FILE* in = fopen(argv[1], "rb");
    if (!in) return 2;
    int32_t h[14];
    if (fread(h, sizeof h, 1, in) != 1) return 2;
    auto clip = std::make_shared<MemoryClip>();
    clip->vi.width = h[0];
    clip->vi.height = h[1];
    clip->vi.component_size = h[2];
    clip->vi.bits_per_component = h[3];
    clip->vi.num_components = h[4];
    clip->vi.sub_w = h[5];
    clip->vi.sub_h = h[6];
    sangnom::Args a;
    a.order = h[7];
    a.aa = h[8];
    a.aac = h[9];
    a.dh = h[10] != 0;
    a.luma = h[11] != 0;
    a.chroma = h[12] != 0;
    const int nframes = h[13];
    clip->vi.num_frames = nframes;
    // the test clips are small: SN_HOST_TEST_SWEEPS=1 (read by this TEST program, the library reads no environment) asks
    // for the whole-plane sweeps instead of the small-launch paths
    if (const char* e = getenv("SN_HOST_TEST_SWEEPS")) a.policy.small_launches = atoi(e) ? SN_SMALL_SWEEP : SN_SMALL_AUTO;
    const bool aa_idiom = argc > 3 && strcmp(argv[3], "aa") == 0;
    if (argc > 3 && !aa_idiom) a.lookahead = atoi(argv[3]);
    std::vector<int> order;
    for (int i = 4; i < argc; ++i) order.push_back(atoi(argv[i]));
    if (order.empty())
        for (int n = 0; n < nframes; ++n) order.push_back(n);
    size_t frame_bytes = 0;
    for (int p = 0; p < clip->vi.num_components; ++p)
        frame_bytes += (size_t)clip->vi.PlaneWidth(p) * clip->vi.component_size * clip->vi.PlaneHeight(p);
    for (int f = 0; f < nframes; ++f) {
        int32_t par;
        if (fread(&par, 4, 1, in) != 1) return 2;
        std::vector<uint8_t> buf(frame_bytes);
        if (frame_bytes && fread(buf.data(), 1, frame_bytes, in) != frame_bytes) return 2;
        clip->parity.push_back(par);
        clip->frames.push_back(std::move(buf));
    }
    fclose(in);

    Env env;
    try {
        FILE* out = fopen(argv[2], "wb");
        if (!out) return 2;
        auto dump = [&](const FramePtr& d) {
            for (int p = 0; p < clip->vi.num_components; ++p)
                for (int y = 0; y < d->Height(p); ++y)
                    fwrite(d->Ptr(p) + (size_t)y * d->Pitch(p), 1, d->RowSize(p), out);
        };
        if (aa_idiom) {
            sangnom::AAFilter<TestHost> flt(clip, a, &env);
            for (int n : order) dump(flt.GetFrame(n, &env));
        } else {
            sangnom::Filter<TestHost> flt(clip, a, &env);
            for (int n : order) dump(flt.GetFrame(n, &env));
        }
        fclose(out);
    } catch (const Error& e) {
        printf("%s\n", e.what());
        return 3;
    }
    return 0;
}
