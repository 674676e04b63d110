#!/usr/bin/env python3
"""Where the whole-plane sweeps overtake the pool path, by frames per launch -- the crossover prefer_pool() (sn_api.hip) estimates
from the geometry for launches that are neither cut into row bands nor large.  Device-resident frames; per geometry and sample
type the time of one launch of n frames through mode="fused" (sweeps whatever the size) and mode="pool", the measured crossover,
and what SN_MODE_AUTO chooses with the bands switched off (sn_debug_set_bands(-1)): fused_frames > 0 means the sweeps.
    python3 tools/prefer_pool_calib.py [fmt:WxH ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avisynth_sangnom2_amd import SangNom2, clip_format  # noqa: E402

CASES = ["Y8:1920x1080", "Y8:3840x2160", "Y8:7680x4320", "YUV420P8:1920x1080", "YUV420P8:3840x2160", "Y16:3840x2160", "YUV420P16:3840x2160",
         "Y32:3840x2160", "YUV420PS:3840x2160"]
if len(sys.argv) > 1:
    CASES = sys.argv[1:]
NS = (1, 2, 4, 8, 16, 24, 32, 48, 64, 96, 128)
dev = torch.device("cuda:0")
tdt = {1: torch.uint8, 2: torch.int16, 4: torch.float32}


def launch_ms(flt, src, dst, reps):
    st = torch.cuda.ExternalStream(flt.stream_handle())
    flt.process_batch(src, dst)
    flt.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps):
        flt.process_batch(src, dst)
    e1.record(st)
    flt.synchronize()
    return e0.elapsed_time(e1) / reps


for case in CASES:
    fmt, wh = case.split(":")
    w, h = (int(v) for v in wh.split("x"))
    clip = clip_format(fmt, w, h)
    kw = dict(aa=48, aac=48)
    g = torch.Generator(device=dev)
    g.manual_seed(11)
    nmax = max(NS)
    full = []
    for p in range(clip.planes):
        hp, wp = h >> (clip.subh if p else 0), w >> (clip.subw if p else 0)
        if clip.bytes == 4:
            full.append(torch.rand((nmax, hp, wp), device=dev, generator=g))
        elif clip.bytes == 2:
            full.append(torch.randint(0, 1 << clip.bits, (nmax, hp, wp), device=dev, generator=g, dtype=torch.int32).to(torch.int16))
        else:
            full.append(torch.randint(0, 256, (nmax, hp, wp), device=dev, generator=g, dtype=torch.uint8))
    out = [torch.empty_like(t) for t in full]
    rows = []
    for n in NS:
        src, dst = [t[:n] for t in full], [t[:n] for t in out]
        ms = {}
        for mode in ("fused", "pool"):
            with SangNom2(clip, max_batch=n, mode=mode, small_launches=1, **kw) as flt:
                ms[mode] = launch_ms(flt, src, dst, 5 if n <= 16 else 3)
        with SangNom2(clip, max_batch=n, mode="auto", small_launches=0, **kw) as flt:
            flt.set_bands(-1, 0)
            flt.process_batch(src, dst)
            flt.synchronize()
            auto = "sweeps" if flt.info().fused_frames > 0 else "pool"
        rows.append((n, ms["fused"], ms["pool"], auto))
    cross = next((n for n, f, p, _ in rows if f < p), None)
    print(f"{case}: measured crossover at n = {cross}")
    for n, f, p, auto in rows:
        best = "sweeps" if f < p else "pool"
        print(f"    n = {n:4d}  sweeps {f:8.3f} ms  pool path {p:8.3f} ms  faster: {best:6s}  auto (bands off) takes: {auto}{'' if auto == best else '   <-- not the faster one'}")
