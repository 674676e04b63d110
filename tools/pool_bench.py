#!/usr/bin/env python3
"""Rate of the pool path (mode="pool": k_prepare -> stage 2 -> k_finalize) on device-resident frames:
tools/pool_bench.py [--fmt Y8 Y16 Y32] [--w 3840] [--h 2160] [--frames 1 16] [--iters 50]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avisynth_sangnom2_amd import SangNom2, clip_format, synth

ap = argparse.ArgumentParser()
ap.add_argument("--fmt", nargs="*", default=["Y8", "Y16", "Y32"])
ap.add_argument("--w", type=int, default=3840)
ap.add_argument("--h", type=int, default=2160)
ap.add_argument("--frames", type=int, nargs="*", default=[1, 16])
ap.add_argument("--iters", type=int, default=50)
args = ap.parse_args()
dev = torch.device("cuda:0")
for fmt in args.fmt:
    clip = clip_format(fmt, args.w, args.h)
    src = synth.frame(clip, "noise", seed=1)
    for n in args.frames:
        with SangNom2(clip, max_batch=n, mode="pool", aa=48, aac=48) as flt:
            dsrc = [torch.from_numpy(np.stack([p] * n)).to(dev) for p in src]
            ddst = [torch.zeros_like(t) for t in dsrc]
            torch.cuda.synchronize()
            for _ in range(3):
                flt.process_batch(dsrc, ddst, parity=[1] * n)
            flt.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.iters):
                flt.process_batch(dsrc, ddst, parity=[1] * n)
            flt.synchronize()
            t = (time.perf_counter() - t0) / args.iters
            print(json.dumps({"frame": f"{args.w}x{args.h} {fmt}", "mode": "pool", "frames_per_launch": n,
                              "ms_per_launch": round(t * 1e3, 4), "frames_per_s": round(n / t, 1)}), flush=True)
