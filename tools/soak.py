#!/usr/bin/env python3
"""Soak test on the GPU: the fused sweeps must give the same bytes run after run, whatever the scratch memory,
LDS and registers held before (an intermittent store hazard was found this way), and the same bytes
as the pool path.  usage: python tools/soak.py [--seconds 120]"""
import argparse
import os
import sys
import time


sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from avisynth_sangnom2_amd import SangNom2, clip_format  # noqa: E402

CASES = [
    ("Y8", 3840, 2160, dict(aa=48), 24),
    ("YUV420P8", 3840, 2160, dict(aa=48, aac=48), 12),
    ("YUV420P16", 3840, 2160, dict(aa=48, aac=48), 8),
    ("Y16", 3840, 2160, dict(aa=48), 16),
    ("Y32", 3840, 2160, dict(aa=48), 12),
    ("Y8", 1920, 1080, dict(aa=48, order=2), 64),
    ("Y8", 7680, 4320, dict(aa=48), 6),
    ("YUV420P8", 720, 480, dict(aa=48, aac=48, fresh_pool=True), 96),
    ("Y8", 2160, 3840, dict(aa=48, fresh_pool=True), 24),
    ("YUV420P8", 1920, 1080, dict(aa=48, aac=48, isolated_planes=True), 32),
    ("YUV420PS", 3840, 2160, dict(aa=48, aac=48), 6),
    ("YUV420P8", 512, 1024, dict(aa=128, aac=128), 64),    # 64 lanes: lanes 62 / 63 of the only strip own columns
    ("YUV422P8", 992, 540, dict(aa=128, aac=128), 48),     # 124 lanes: the same in the second strip
    # U and V as one sweep (sn_fused_u8_uv.hip; the 2160p and 512-wide cases above take it as well): four strips with the
    # region's edge inside the second, three strips with it inside the second, and the region's edge exactly on a seam
    ("YUV420P8", 1920, 1080, dict(aa=48, aac=48), 32),
    ("YUV420P8", 1280, 720, dict(aa=48, aac=128, order=2), 48),
    ("YUV420P8", 992, 720, dict(aa=128, aac=128), 48),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    tdt = {1: torch.uint8, 2: torch.int16, 4: torch.float32}
    t_end = time.time() + a.seconds
    rounds, bad = 0, 0
    state = []
    for fmt, w, h, kw, n in CASES:
        clip = clip_format(fmt, w, h)
        g = torch.Generator(device=dev)
        g.manual_seed(7)
        src = []
        for p in range(clip.planes):
            hp, wp = h >> (clip.subh if p else 0), w >> (clip.subw if p else 0)
            if clip.bytes == 4:
                src.append(torch.rand((n, hp, wp), device=dev, generator=g))
            elif clip.bytes == 2:
                src.append(torch.randint(0, 1 << clip.bits, (n, hp, wp), device=dev, generator=g, dtype=torch.int32).to(torch.int16))
            else:
                src.append(torch.randint(0, 256, (n, hp, wp), device=dev, generator=g, dtype=torch.uint8))
        flt = SangNom2(clip, max_batch=n, small_launches=1, **kw)  # SN_SMALL_SWEEP: this is about the sweeps, never hand a small launch to the pool path
        assert flt.info().fused_eligible == 1, fmt
        ref = [torch.empty_like(s) for s in src]
        with SangNom2(clip, max_batch=n, mode="pool", **kw) as pool:
            torch.cuda.synchronize()
            pool.process_batch(src, ref)
            pool.synchronize()
        state.append((fmt, w, h, flt, src, ref))
    while time.time() < t_end:
        for fmt, w, h, flt, src, ref in state:
            junk = torch.randint(0, 255, (64 << 20,), device=dev, dtype=torch.uint8)  # dirty some freed memory
            del junk
            dst = [torch.full_like(s, 77 if s.dtype != torch.float32 else 0.5) for s in src]
            torch.cuda.synchronize()
            flt.process_batch(src, dst)
            flt.synchronize()
            for p in range(len(src)):
                if not torch.equal(dst[p].view(torch.uint8), ref[p].view(torch.uint8)):
                    ne = (dst[p] != ref[p]).nonzero()
                    print(f"MISMATCH {fmt} {w}x{h} plane {p}: {len(ne)} samples, first {ne[:3].tolist()}", flush=True)
                    bad += 1
        rounds += 1
        if rounds % 5 == 0:
            print(f"round {rounds}, mismatches so far {bad}", flush=True)
    print(f"soak: {rounds} rounds over {len(state)} configurations, {bad} mismatches")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
