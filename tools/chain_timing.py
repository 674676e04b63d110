#!/usr/bin/env python3
"""Where the waves of k_smooth_u8_chain spend their cycles (a library built with -DSN_CHAIN_TIMING -DSN_TEST_HOOKS):
tools/chain_timing.py -- the 720x480 YUV420P8 history-carrying stream with 1 and 8 workgroups per buffer (SN_CHAIN_GROUPS),
and single 2160p YUV420P8 frames (U and V as a chain of two passes)."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avisynth_sangnom2_amd import SangNom2, capi, clip_format, synth  # noqa: E402

lib = capi.load()
lib.sn_debug_chain_cycles.argtypes = [ctypes.POINTER(ctypes.c_ulonglong * 6), ctypes.c_int]
NAMES = ("barrier", "hand-off wait", "schedule + ghost refresh", "rows", "publish")


def report(tag):
    out = (ctypes.c_ulonglong * 6)()
    assert lib.sn_debug_chain_cycles(ctypes.byref(out), 1) == 0
    n = max(out[5], 1)
    tot = sum(out[k] for k in range(5))
    print(f"{tag}: {out[5]} wave-rounds with rows, {tot / n:.0f} cycles per wave-round")
    for k in range(5):
        print(f"    {NAMES[k]:26s} {out[k] / n:8.0f} cycles  {100.0 * out[k] / max(tot, 1):5.1f} %")


dev = torch.device("cuda:0")
for groups in (1, 8):
    os.environ["SN_CHAIN_GROUPS"] = str(groups)
    clip = clip_format("YUV420P8", 720, 480)
    n = 256
    frames = [synth.frame(clip, "noise", seed=s) for s in range(4)]
    with SangNom2(clip, max_batch=n, aa=48, aac=48) as flt:
        src = [torch.from_numpy(np.stack([frames[f % 4][p] for f in range(n)])).to(dev) for p in range(3)]
        dst = [torch.zeros((n,) + flt.plane_shape_out(p), dtype=torch.uint8, device=dev) for p in range(3)]
        flt.process_batch(src, dst)
        flt.synchronize()
        report(f"warm-up, {groups} workgroup(s) per buffer")
        flt.process_batch(src, dst)
        flt.synchronize()
        report(f"720x480 YUV420P8 x {n}, {groups} workgroup(s) per buffer")
os.environ.pop("SN_CHAIN_GROUPS")
clip = clip_format("YUV420P8", 3840, 2160)
frame = synth.frame(clip, "noise", seed=1)
with SangNom2(clip, aa=48, aac=48) as flt:
    for _ in range(3):
        flt.get_frame(frame)
    report("warm-up, single 2160p frames")
    for _ in range(20):
        flt.get_frame(frame)
    report("single 2160p YUV420P8 frames (U, V as one chain)")
