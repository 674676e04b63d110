"""Helpers shared by the parity tests."""
import numpy as np

from avisynth_sangnom2_amd import ClipFormat, synth
from oracle.oracle import Config


def oracle_cfg(clip: ClipFormat, **kw) -> Config:
    return Config(width=clip.width, height=clip.height, bytes=clip.bytes, bits=clip.bits, planes=clip.planes,
                  subw=clip.subw, subh=clip.subh, **kw)


def same(a: np.ndarray, b: np.ndarray) -> bool:
    """Bit-exact comparison (float planes compared on their bit patterns)."""
    if a.dtype == np.float32:
        return np.array_equal(a.view(np.uint32), b.view(np.uint32))
    return np.array_equal(a, b)


def describe_diff(a, b):
    ne = a != b
    idx = np.argwhere(ne)
    return f"{int(ne.sum())} differing samples, first at {idx[:4].tolist()}: {a[ne][:4]} vs {b[ne][:4]}"


def make_frames(clip, pattern, n, seed0=0):
    return [synth.frame(clip, pattern, seed=seed0 + i) for i in range(n)]
