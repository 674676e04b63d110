#!/usr/bin/env python3
"""PCIe-inclusive rate of the plugin path (host frame -> H2D -> sweeps -> D2H -> host frame) on one GPU:
the synchronous sn_process_host against the pipelined host ring (sn_submit_host / sn_collect_host) at several
depths.  Never bench.py's `value` -- that one is device-resident.
usage: python tools/host_path_bench.py [--workload 2160p-Y8] [--frames 256]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avisynth_sangnom2_amd import SangNom2, clip_format, synth  # noqa: E402
from bench import WORKLOADS  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="2160p-Y8", choices=sorted(WORKLOADS))
    ap.add_argument("--frames", type=int, default=256)
    ap.add_argument("--depths", default="2,4,8,16,32,64")
    a = ap.parse_args()
    fmt, w, h, kw = WORKLOADS[a.workload]
    clip = clip_format(fmt, w, h)
    ring = [synth.frame(clip, "noise", seed=s) for s in range(8)]  # distinct host frames, reused round-robin
    out = {"workload": a.workload, "frames": a.frames}

    with SangNom2(clip, **kw) as flt:
        dst = flt.get_frame(ring[0])
        n = max(8, a.frames // 8)
        t0 = time.perf_counter()
        for f in range(n):
            flt.get_frame(ring[f % 8], dst=dst)
        out["sync_fps"] = round(n / (time.perf_counter() - t0), 1)

    for depth in [int(x) for x in a.depths.split(",")]:
        with SangNom2(clip, host_depth=depth, **kw) as flt:
            slots = flt.host_slots()
            dst = [np.zeros(flt.plane_shape_out(p), dtype=clip.dtype) for p in range(flt.nplanes)]
            inflight = []
            for f in range(slots):  # warm-up: allocations, first launches
                inflight.append(flt.submit(ring[f % 8]))
            while inflight:
                flt.collect(inflight.pop(0), dst)
            t0 = time.perf_counter()
            for f in range(a.frames):
                if len(inflight) == slots:
                    flt.collect(inflight.pop(0), dst)
                inflight.append(flt.submit(ring[f % 8]))
            while inflight:
                flt.collect(inflight.pop(0), dst)
            out[f"ring{slots}_fps"] = round(a.frames / (time.perf_counter() - t0), 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
