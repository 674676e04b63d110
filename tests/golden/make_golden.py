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


# BASELINE.json's configurations at full size: too large to commit as planes, so the SHA-256 of every output plane
# is kept; inputs come from the portable generator (avisynth_sangnom2_amd/synth.py).
FULL_SIZE = [
    ("1080p Y8", "Y8", 1920, 1080, dict(order=1, aa=48), "noise", 2),
    ("2160p Y8", "Y8", 3840, 2160, dict(order=1, aa=48), "checker", 1),
    ("2160p YUV420P8", "YUV420P8", 3840, 2160, dict(aa=48, aac=48), "noise", 1),
    ("2160p YUV420P16", "YUV420P16", 3840, 2160, dict(aa=48, aac=48), "noise", 1),
    ("2160p-out YUV444PS dh", "YUV444PS", 3840, 1080, dict(aa=48, aac=48, dh=True), "sine", 1),
    ("4320p Y8", "Y8", 7680, 4320, dict(order=1, aa=48), "edges", 1),
    # round 3: more content for the default single-frame path (row bands / luma bands + pool chroma)
    ("2160p Y8 noise", "Y8", 3840, 2160, dict(order=1, aa=48), "noise", 1),
    ("2160p YUV420P8 checker", "YUV420P8", 3840, 2160, dict(aa=48, aac=48), "checker", 1),
    ("2160p YUV420PS", "YUV420PS", 3840, 2160, dict(aa=48, aac=48), "noise", 1),
    ("2160p YUV420P16 checker", "YUV420P16", 3840, 2160, dict(aa=48, aac=48), "checker", 1),
]


def full_size_hashes():
    import hashlib
    out = []
    for name, fmt, w, h, kw, pattern, nframes in FULL_SIZE:
        clip = clip_format(fmt, w, h)
        ora = Oracle(Config(width=w, height=h, bytes=clip.bytes, bits=clip.bits, planes=clip.planes,
                            subw=clip.subw, subh=clip.subh, **kw))
        frames = []
        for f in range(nframes):
            res = ora.process(synth.frame(clip, pattern, seed=500 + f), parity=1)
            frames.append([hashlib.sha256(np.ascontiguousarray(pl).tobytes()).hexdigest() for pl in res])
        out.append(dict(name=name, fmt=fmt, width=w, height=h, kw=kw, pattern=pattern, seed0=500, sha256=frames))
        print(name, "hashed")
    with open(os.path.join(HERE, "full_size_sha256.json"), "w") as f:
        json.dump(out, f, indent=1)


def main():
    full_size_hashes()
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
    if "--hashes-only" in sys.argv:
        full_size_hashes()
    else:
        main()
