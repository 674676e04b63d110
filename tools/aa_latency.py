#!/usr/bin/env python3
"""Synchronous SangNomAA (sn_aa_process_host: turn, SangNom2, turn back, SangNom2) on one host frame at a time:
tools/aa_latency.py [--fmt Y8] [--w 1920] [--h 1080].  --sweeps switches the row bands off (sn_policy.small_launches = SN_SMALL_SWEEP)
(whole-plane sweeps), for comparison."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avisynth_sangnom2_amd import SangNomAAHost, clip_format, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--fmt", default="Y8")
ap.add_argument("--w", type=int, default=1920)
ap.add_argument("--h", type=int, default=1080)
ap.add_argument("--iters", type=int, default=100)
ap.add_argument("--sweeps", action="store_true")
a = ap.parse_args()
clip = clip_format(a.fmt, a.w, a.h)
src = synth.frame(clip, "noise", seed=1)
with SangNomAAHost(clip, aa=48, aac=48, small_launches=1 if a.sweeps else 0) as flt:
    for _ in range(5):
        flt.get_frame(src)
    t0 = time.perf_counter()
    for _ in range(a.iters):
        flt.get_frame(src)
    dt = (time.perf_counter() - t0) / a.iters
print(json.dumps({"frame": f"{a.w}x{a.h} {a.fmt}", "aa_sync_ms": round(dt * 1e3, 3), "aa_sync_fps": round(1 / dt, 1),
                  "bands": not a.sweeps}))
