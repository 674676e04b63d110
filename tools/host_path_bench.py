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

    # pageable frames, the destination named at submission (what GetFrame has: NewVideoFrame precedes the work): the
    # output's kept lines are copied on the host at submission, only the interpolated lines come back over PCIe
    for depth in [int(x) for x in a.depths.split(",")][-3:]:
        with SangNom2(clip, host_depth=depth, **kw) as flt:
            slots = flt.host_slots()
            dsts = [[np.zeros(flt.plane_shape_out(p), dtype=clip.dtype) for p in range(flt.nplanes)] for _ in range(slots)]
            inflight = []
            for f in range(slots):
                inflight.append((flt.submit(ring[f % 8], dst=dsts[f % slots]), dsts[f % slots]))
            while inflight:
                sl, d = inflight.pop(0)
                flt.collect(sl, d, announced=True)
            t0 = time.perf_counter()
            for f in range(a.frames):
                if len(inflight) == slots:
                    sl, d = inflight.pop(0)
                    flt.collect(sl, d, announced=True)
                inflight.append((flt.submit(ring[f % 8], dst=dsts[f % slots]), dsts[f % slots]))
            while inflight:
                sl, d = inflight.pop(0)
                flt.collect(sl, d, announced=True)
            out[f"ring{slots}_to_fps"] = round(a.frames / (time.perf_counter() - t0), 1)

    # the same with the host's frames in pinned memory (sn_pin_host_buffer) and the destination named at submission
    from avisynth_sangnom2_amd import pin_host_array, unpin_host_array
    for depth in [int(x) for x in a.depths.split(",")][-3:]:
        with SangNom2(clip, host_depth=depth, **kw) as flt:
            slots = flt.host_slots()
            n = flt.nplanes
            src_arena = [np.stack([ring[i][p] for i in range(8)]) for p in range(n)]
            dst_arena = [np.zeros((slots,) + flt.plane_shape_out(p), dtype=clip.dtype) for p in range(n)]
            for x in src_arena + dst_arena:
                pin_host_array(x)
            try:
                with SangNom2(clip, **kw) as sync:
                    m = max(8, a.frames // 8)
                    sync.get_frame([src_arena[p][0] for p in range(n)], dst=[dst_arena[p][0] for p in range(n)])
                    t0 = time.perf_counter()
                    for f in range(m):
                        sync.get_frame([src_arena[p][f % 8] for p in range(n)], dst=[dst_arena[p][0] for p in range(n)])
                    out["sync_pinned_fps"] = round(m / (time.perf_counter() - t0), 1)
                inflight = []
                t0 = None
                for f in range(a.frames + slots):
                    if f == slots:
                        t0 = time.perf_counter()  # the first round is the warm-up
                    if len(inflight) == slots:
                        flt.collect(inflight.pop(0), announced=True)
                    k = f % slots
                    inflight.append(flt.submit([src_arena[p][f % 8] for p in range(n)], dst=[dst_arena[p][k] for p in range(n)]))
                while inflight:
                    flt.collect(inflight.pop(0), announced=True)
                out[f"ring{slots}_pinned_fps"] = round(a.frames / (time.perf_counter() - t0), 1)
            finally:
                for x in src_arena + dst_arena:
                    unpin_host_array(x)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
