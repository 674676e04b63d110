#!/usr/bin/env python3
"""Randomised parity run on the GPU: random formats, sizes, arguments, extension flags and frame sequences through
the C ABI (host frames, device batches and the host ring) against oracle instances.  usage: python tools/fuzz.py [--seconds 120] [--seed 1]"""
import argparse
import os
import random
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avisynth_sangnom2_amd import ClipFormat, SangNom2, clip_format, synth  # noqa: E402
from oracle.oracle import Config, Oracle  # noqa: E402
from tests.util import to_host  # noqa: E402

FORMATS = ["Y8", "Y8", "Y8", "Y10", "Y16", "Y32", "YUV420P8", "YUV420P8", "YUV420P16", "YUV422P8", "YUV444P8", "YUV444PS", "YUV420PS"]


def cfg_of(clip, **kw):
    return Config(width=clip.width, height=clip.height, bytes=clip.bytes, bits=clip.bits, planes=clip.planes, subw=clip.subw, subh=clip.subh, **kw)


def same(a, b):
    return np.array_equal(a.view(np.uint8), b.view(np.uint8))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    rng = random.Random(a.seed)
    t_end = time.time() + a.seconds
    n = bad = 0
    stats = {"fused": 0, "pool": 0, "ring": 0, "host": 0, "batch": 0, "frames": 0, "pixels": 0, "banded_frames": 0, "band_fallbacks": 0, "chained_frames": 0,
             "uv_sweep_configs": 0, "chain_redone": 0}
    while time.time() < t_end:
        fmt = rng.choice(FORMATS)
        wide = rng.random() < 0.15
        w = rng.choice([32, 64, 96, 128, 160, 256, 320, 512, 544, 640, 992, 1024]) if not wide else rng.choice([1472, 1920, 1952, 2048, 2432, 2880, 3840, 4096, 5120, 7680])
        if rng.random() < 0.25:
            w = rng.choice([40, 72, 100, 200, 360, 720, 1080]) if not wide else 2160  # not a multiple of 32
        h = 2 * rng.randint(1, 40 if not wide else 12)
        if rng.random() < 0.3:
            h = 2 * rng.randint(20, 200 if not wide else 60)  # tall enough for the row bands of the latency path
            if rng.random() < 0.15:
                h = 2 * rng.randint(200, 540)  # ... and for the hand-off's dependency cone to reach the last columns
        probe = clip_format(fmt, 64, 32)
        if probe.planes == 3:
            w -= w % (4 if probe.subw else 1) or 0
            if probe.subh:
                h += (-h) % 4
            if probe.subw and (w // 2) % 2:
                w += 2
        if probe.bytes > 1 and w > 3840 and rng.random() < 0.7:
            w = 3840
        clip = clip_format(fmt, w, h)
        kw = dict(order=rng.randint(0, 2), aa=rng.choice([0, 1, 20, 48, 100, 128]), aac=rng.choice([0, 48, 128]),
                  dh=rng.random() < 0.2, luma=rng.random() < 0.85, chroma=rng.random() < 0.8)
        ext = rng.choice(["none", "none", "isolated", "fresh"])
        way = rng.choice(["host", "ring", "batch"])  # batch: device-resident frames, one launch (history-carrying 8-bit clips: the chain)
        nframes = rng.randint(1, 4) if way != "batch" else rng.randint(2, 10)
        pattern = rng.choice(["noise", "noise", "checker", "edges", "sine"])
        frames = [synth.frame(clip, pattern, seed=rng.randint(0, 1 << 20)) for _ in range(nframes)]
        parity = [rng.randint(0, 1) for _ in range(nframes)]
        # expected
        if ext == "none":
            ora = Oracle(cfg_of(clip, **kw))
            want = [ora.process(frames[f], parity=parity[f]) for f in range(nframes)]
        else:
            per_plane = {}
            want = []
            for f in range(nframes):
                outs = []
                for p in range(clip.planes):
                    pl = frames[f][p]
                    yclip = ClipFormat(width=pl.shape[1], height=pl.shape[0], bytes=clip.bytes, bits=clip.bits)
                    enabled = kw["luma"] if p == 0 else kw["chroma"]
                    mk = lambda: Oracle(cfg_of(yclip, order=kw["order"], aa=kw["aa"] if p == 0 else kw["aac"], dh=kw["dh"], luma=enabled))
                    o = mk() if ext == "fresh" else per_plane.setdefault(p, mk())
                    outs.append(o.process([pl], parity=parity[f])[0])
                want.append(outs)
        small = rng.choice([1, 0])  # SN_SMALL_SWEEP: the whole-plane sweeps, or auto mode's small-launch paths (bands, pool kernels)
        try:
            sweeps = rng.choice([0, 0, 0, 1])  # 8-bit 4:2:0: U and V as one sweep (default) or a sweep each
            flt = SangNom2(clip, host_depth=rng.choice([1, 2, 3, 4, 5, 8, 12]), max_batch=nframes, isolated_planes=ext == "isolated", fresh_pool=ext == "fresh", small_launches=small,
                           chroma_sweeps=sweeps, **kw)
        except Exception as e:  # a geometry the library rejects must be one it documents
            if "exceeds the supported maximum" in str(e):
                continue
            raise
        with flt:
            band_set = None
            if small == 0:  # auto mode: the row bands too, now and then with a run-up that is too short
                band_set = (rng.choice([0, 0, 2, 3, 5, 9, 16]), rng.choice([0, 0, 0, 1, 6, 12]))
                flt.set_bands(*band_set)
            got = []
            fused = False
            stats[way] += 1
            stats["frames"] += nframes
            stats["pixels"] += nframes * w * h
            if way == "host":
                got = [flt.get_frame(frames[f], parity=parity[f]) for f in range(nframes)]
            elif way == "batch":
                import torch
                dev = torch.device("cuda:0")
                view = {1: np.uint8, 2: np.int16, 4: np.float32}[clip.bytes]
                src = [torch.from_numpy(np.stack([fr[p] for fr in frames]).view(view)).to(dev) for p in range(clip.planes)]
                dst = [torch.zeros((nframes,) + flt.plane_shape_out(p), dtype=src[p].dtype, device=dev) for p in range(clip.planes)]
                torch.cuda.synchronize()
                flt.process_batch(src, dst, parity=parity)
                flt.synchronize()
                got = [[to_host(dst[p][f]).view(clip.dtype) for p in range(clip.planes)] for f in range(nframes)]
            else:
                slots = flt.host_slots()
                inflight = []
                for f in range(nframes):
                    if len(inflight) == slots:
                        got.append(flt.collect(inflight.pop(0)))
                    inflight.append(flt.submit(frames[f], parity=parity[f]))
                while inflight:
                    got.append(flt.collect(inflight.pop(0)))
            info = flt.info()
            fused = info.fused_frames > 0
            stats["banded_frames"] += info.banded_frames
            stats["band_fallbacks"] += info.band_fallbacks
            stats["chained_frames"] += info.chained_frames
            stats["uv_sweep_configs"] += info.uv_sweeps
            stats["chain_redone"] += info.chain_redone
        stats["fused" if fused else "pool"] += 1
        for f in range(nframes):
            for p in range(clip.planes):
                if not same(want[f][p], got[f][p]):
                    bad += 1
                    d = np.argwhere(want[f][p] != got[f][p])
                    np.savez(f"gpurun_out/fuzz_mismatch_{n}.npz", **{f"src{q}": frames[f][q] for q in range(clip.planes)},
                             **{f"want{q}": want[f][q] for q in range(clip.planes)}, **{f"got{q}": got[f][q] for q in range(clip.planes)})
                    print(f"MISMATCH {fmt} {w}x{h} {kw} ext={ext} way={way} frame {f}/{nframes} plane {p} pattern={pattern} parity={parity} "
                          f"small_launches={small} chroma_sweeps={sweeps} uv={info.uv_sweeps} bands={band_set} banded={info.banded_frames} fallbacks={info.band_fallbacks} "
                          f"fused={info.fused_frames} n={len(d)} rows {d[:, 0].min()}..{d[:, 0].max()} cols {d[:, 1].min()}..{d[:, 1].max()}", flush=True)
        n += 1
        if n % 50 == 0:
            print(f"{n} configurations, {bad} mismatches", flush=True)
    print(f"fuzz: {n} configurations, {bad} mismatches (seed {a.seed}); {stats}")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
