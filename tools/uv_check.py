#!/usr/bin/env python3
"""tools/uv_check.py: the one-sweep chroma passes (sn_fused_u8_uv.hip) against the pool path and the two-sweep form,
plane by plane, over geometries that put the region's right edge inside a strip, on a seam and next to one."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avisynth_sangnom2_amd import SangNom2, clip_format, synth  # noqa: E402

GEOS = [(512, 64), (512, 640), (992, 720), (1024, 720), (544, 400), (1472, 1000), (1920, 1080), (1280, 720), (2560, 1440), (3840, 2160),
        (256, 64), (640, 48), (3840, 32), (2048, 1080), (3200, 1800), (960, 540 // 4 * 4)]
FMT = "YUV420P8"
args = sys.argv[1:]
if args and not args[0][0].isdigit():
    FMT, args = args[0], args[1:]  # e.g. YUV422P8
if args:
    GEOS = [tuple(int(v) for v in a.split("x")) for a in args]
bad = 0
for w, h in GEOS:
    if FMT == "YUV422P8":
        h = max(16, h // 2)
    clip = clip_format(FMT, w, h)
    for pattern, kw in (("noise", dict(aa=48, aac=48)), ("edges", dict(aa=128, aac=128)), ("noise", dict(aa=10, aac=0, order=2))):
        src = synth.frame(clip, pattern, seed=w + h)
        outs, uv = {}, {}
        for name, extra in (("uv", dict(mode="fused")), ("two", dict(mode="fused", chroma_sweeps=1)), ("pool", dict(mode="pool"))):
            with SangNom2(clip, **kw, **extra) as flt:
                outs[name] = flt.get_frame(src)
                uv[name] = flt.info().uv_sweeps
        line = f"{w}x{h} {pattern} {kw}: uv_sweeps={uv['uv']}/{uv['two']}"
        for p in range(3):
            d1 = int((outs["uv"][p] != outs["pool"][p]).sum())
            d2 = int((outs["two"][p] != outs["pool"][p]).sum())
            line += f"  plane{p}: uv-pool {d1} two-pool {d2}"
            if d1:
                ys, xs = np.nonzero(outs["uv"][p] != outs["pool"][p])
                line += f" [rows {ys.min()}..{ys.max()} cols {xs.min()}..{xs.max()}]"
                bad += 1
        print(line, flush=True)
print("MISMATCHING PLANES:", bad)
sys.exit(1 if bad else 0)
