#!/usr/bin/env python3
"""Single-frame latency of the synchronous GetFrame (sn_process_host) and of one device-resident frame:
tools/latency_bench.py [--fmt Y8] [--w 3840] [--h 2160] [--bands N] [--warm N]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avisynth_sangnom2_amd import SangNom2, clip_format, synth, pin_host_array, unpin_host_array

ap = argparse.ArgumentParser()
ap.add_argument("--fmt", default="Y8")
ap.add_argument("--w", type=int, default=3840)
ap.add_argument("--h", type=int, default=2160)
ap.add_argument("--bands", type=int, nargs="*", default=[-1, 0, 32, 64, 128])
ap.add_argument("--warm", type=int, default=0)
ap.add_argument("--iters", type=int, default=200)
ap.add_argument("--frames", type=int, default=1)
args = ap.parse_args()
clip = clip_format(args.fmt, args.w, args.h)
kw = dict(aa=48, aac=48)
src = synth.frame(clip, "noise", seed=1)
dev = torch.device("cuda:0")
for bands in args.bands:
    with SangNom2(clip, max_batch=max(args.frames, 1), **kw) as flt:
        flt.set_bands(bands, args.warm)
        n = args.frames
        dsrc = [torch.from_numpy(np.stack([p] * n)).to(dev) for p in src]
        ddst = [torch.zeros_like(t) for t in dsrc]
        torch.cuda.synchronize()
        for _ in range(5):
            flt.process_batch(dsrc, ddst, parity=[1] * n)
        flt.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.iters):
            flt.process_batch(dsrc, ddst, parity=[1] * n)
            flt.synchronize()
        t_dev = (time.perf_counter() - t0) / args.iters
        # synchronous GetFrame, pinned host frames
        hs = [np.ascontiguousarray(p) for p in src]
        hd = [np.zeros_like(p) for p in src]
        for a in hs + hd:
            pin_host_array(a)
        for _ in range(5):
            flt.get_frame(hs, dst=hd)
        t0 = time.perf_counter()
        for _ in range(args.iters):
            flt.get_frame(hs, dst=hd)
        t_sync = (time.perf_counter() - t0) / args.iters
        for a in hs + hd:
            unpin_host_array(a)
        info = flt.info()
        print(json.dumps({"frame": f"{args.w}x{args.h} {args.fmt}", "bands": bands, "warm": args.warm, "frames_per_launch": n,
                          "device_ms": round(t_dev * 1e3, 4), "sync_pinned_ms": round(t_sync * 1e3, 4),
                          "sync_pinned_fps": round(1 / t_sync, 1), "banded_frames": info.banded_frames,
                          "band_fallbacks": info.band_fallbacks}), flush=True)
