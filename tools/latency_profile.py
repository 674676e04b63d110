#!/usr/bin/env python3
"""One device-resident frame at a time, for rocprofv3 --kernel-trace --stats: which kernels the small-launch path runs
and how long each takes.  usage: tools/latency_profile.py [--fmt Y8] [--iters 50]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avisynth_sangnom2_amd import SangNom2, clip_format, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--fmt", default="Y8")
ap.add_argument("--w", type=int, default=3840)
ap.add_argument("--h", type=int, default=2160)
ap.add_argument("--iters", type=int, default=50)
ap.add_argument("--bands", type=int, default=0)
args = ap.parse_args()
clip = clip_format(args.fmt, args.w, args.h)
src = synth.frame(clip, "noise", seed=1)
dev = torch.device("cuda:0")
with SangNom2(clip, aa=48, aac=48) as flt:
    flt.set_bands(args.bands, 0)
    dsrc = [torch.from_numpy(p[None]).to(dev) for p in src]
    ddst = [torch.zeros_like(t) for t in dsrc]
    torch.cuda.synchronize()
    for _ in range(args.iters):
        flt.process_batch(dsrc, ddst, parity=[1])
        flt.synchronize()
