"""Regenerates tests/golden/*.npz from the C oracle (oracle/sangnom_oracle.c).

These are REGRESSION vectors, not reference-pinned ones: the reference has no fixtures and cannot
be built in this image ("Oracle").  Each file holds the configuration, the input planes
of every frame and the oracle's output planes; sizes are a few KB.

    python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from avisynth_sangnom2_amd import clip_format, synth  # noqa: E402
from oracle.oracle import Config, Oracle  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

CASES = [
    ("y8_top", "Y8", 64, 32, dict(order=1, aa=48), "edges", 1),
    ("y8_bottom_noise", "Y8", 96, 24, dict(order=2, aa=48), "noise", 1),
    ("y8_w100_3frames", "Y8", 100, 24, dict(order=0, aa=128), "noise", 3),
    ("y16", "Y16", 64, 24, dict(aa=48), "checker", 1),
    ("y10", "Y10", 64, 24, dict(aa=48), "noise", 1),
    ("y32", "Y32", 64, 24, dict(aa=48), "noise", 1),
    ("yuv420p8", "YUV420P8", 64, 32, dict(aa=48, aac=48), "noise", 2),
    ("yuv420p16", "YUV420P16", 64, 32, dict(aa=48, aac=48), "edges", 1),
    ("yuv444ps_dh", "YUV444PS", 64, 16, dict(aa=48, aac=48, dh=True), "sine", 1),
    ("yuv420p8_lumaonly", "YUV420P8", 64, 32, dict(chroma=False), "noise", 1),
]


def main():
    for name, fmt, w, h, kw, pattern, nframes in CASES:
        clip = clip_format(fmt, w, h)
        ora = Oracle(Config(width=w, height=h, bytes=clip.bytes, bits=clip.bits, planes=clip.planes,
                            subw=clip.subw, subh=clip.subh, **kw))
        arrays = {}
        for f in range(nframes):
            src = synth.frame(clip, pattern, seed=100 + f)
            out = ora.process(src, parity=(f + 1) & 1)
            for p in range(len(src)):
                arrays[f"in_f{f}_p{p}"] = src[p]
                arrays[f"out_f{f}_p{p}"] = out[p]
        meta = dict(fmt=fmt, width=w, height=h, kw=kw, pattern=pattern, nframes=nframes, seed0=100)
        arrays["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **arrays)
        print(name, sum(a.nbytes for a in arrays.values()), "bytes raw")


if __name__ == "__main__":
    main()
