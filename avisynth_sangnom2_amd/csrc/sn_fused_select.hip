// sn_fused_select.hip -- which configurations the fused sweeps serve (host code only).
//
// A configuration is eligible when every plane fits one workgroup (width a multiple of 32; up to 7680 columns for
// 8-bit, 3840 for 16-bit and float samples) and either every processed plane is as large as the pool (no pass can
// see another pass's leftovers: SURVEY.md 0.7) or the chroma planes are subsampled AND luma is processed first --
// then the sweeps couple the passes through hand-off pools; without a luma pass the chroma passes
// would see the previous FRAME's leftovers, which only the pool path reproduces.
#include <stdint.h>

#include "sn_internal.h"

namespace sn {

static bool chroma_subsampled_and_processed(const sn_config& c)
{
    const int np = c.num_planes < 3 ? c.num_planes : 3;
    return np == 3 && (c.dh || c.chroma) && (c.sub_w != 0 || c.sub_h != 0);
}

bool fused_needs_pools(const sn_config& c) { return chroma_subsampled_and_processed(c); }

bool fused_plane_eligible(int bytes_per_sample, int w)
{
    if (bytes_per_sample == 1) return fused_v3_plane_ok(w);
    if (bytes_per_sample == 2) return fused_u16_plane_ok(w);
    return fused_f32_plane_ok(w);
}

bool fused_padded_plane_eligible(int bytes_per_sample, int w)
{
    if (w % 8 != 0) return false;  // the plane must end on a lane boundary (8 columns per lane)
    return fused_plane_eligible(bytes_per_sample, (w + 31) & ~31);
}

bool fused_eligible(const sn_config& c)
{
    if (!fused_plane_eligible(c.bytes_per_sample, c.width)) return false;
    if (chroma_subsampled_and_processed(c)) {
        if (!(c.dh || c.luma)) return false;
        if ((c.width >> c.sub_w) % 8 != 0) return false;
    }
    return true;
}

// Pointer / pitch alignment the 8-byte vector accesses need.
bool fused_layout_ok(const PlaneArgs& p)
{
    auto a8 = [](uintptr_t v) { return (v & 7) == 0; };
    return a8((uintptr_t)p.src) && a8((uintptr_t)p.dst) && a8((uintptr_t)p.src_pitch) && a8((uintptr_t)p.dst_pitch) &&
           a8((uintptr_t)p.src_frame_stride) && a8((uintptr_t)p.dst_frame_stride);
}

}  // namespace sn
