#!/usr/bin/env python3
"""Device-resident rate of the anti-aliasing idiom TurnLeft().SangNom2().TurnRight().SangNom2() (SangNomAA) and of
the turn kernel alone.  usage: python tools/aa_bench.py [--frames 512] [--fresh 1]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from avisynth_sangnom2_amd import SangNom2, SangNomAA, clip_format  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=512)
    ap.add_argument("--fresh", type=int, default=1)
    ap.add_argument("--fmt", default="Y8")
    ap.add_argument("--size", default="3840x2160")
    a = ap.parse_args()
    w, h = [int(x) for x in a.size.split("x")]
    clip = clip_format(a.fmt, w, h)
    dev = torch.device("cuda:0")
    N = a.frames
    src = [torch.randint(0, 256, (N, h, w), device=dev, dtype=torch.uint8)]
    dst = [torch.empty_like(src[0])]
    out = {"clip": f"{a.fmt} {w}x{h}", "frames": N, "fresh_pool": bool(a.fresh)}
    with SangNomAA(clip, max_batch=N, fresh_pool=bool(a.fresh)) as aa:
        torch.cuda.synchronize()
        for _ in range(2):
            aa.process_batch(src, dst)
        aa.synchronize()
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            aa.process_batch(src, dst)
        aa.synchronize()
        out["aa_fps"] = round(N * reps / (time.perf_counter() - t0), 1)
    with SangNom2(clip, max_batch=N) as flt:
        t = torch.empty((N, w, h), device=dev, dtype=torch.uint8)
        torch.cuda.synchronize()
        flt.turn(src[0], t, 1)
        flt.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            flt.turn(src[0], t, 1)
        flt.synchronize()
        dt = (time.perf_counter() - t0) / reps
        out["turn_fps"] = round(N / dt, 1)
        out["turn_GBps"] = round(2 * N * w * h / dt / 1e9, 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
