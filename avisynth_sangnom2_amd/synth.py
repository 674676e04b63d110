"""Portable synthetic frame generator ("Synthetic inputs").

Everything is spelled out on uint64 arithmetic (splitmix64) so that this container and the GPU
box produce identical frames from a seed; nothing depends on numpy's or torch's own generators.

Patterns:
  noise    uniform over the full sample range (worst case for the wrap paths)
  checker  hard 0/max checker ((x//5 + y//3) & 1)  (maximises wrap in stage 2)
  sine     smooth diagonal sinusoid (typical anti-aliasing input)
  edges    noise-modulated slanted edges (exercises every direction of the ladder)
"""
from __future__ import annotations

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(idx: np.ndarray, seed: int) -> np.ndarray:
    """splitmix64 output for counter values seed + idx (vectorised, wraps modulo 2^64)."""
    with np.errstate(over="ignore"):
        z = (idx.astype(np.uint64) + np.uint64(seed & 0xFFFFFFFFFFFFFFFF)) * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def plane(h: int, w: int, bytes: int = 1, bits: int = 8, pattern: str = "noise", seed: int = 0) -> np.ndarray:
    """One h x w plane of the given sample format."""
    idx = np.arange(h * w, dtype=np.uint64).reshape(h, w)
    r = splitmix64(idx, seed * 0x1000193 + 0x5bd1e995)
    if bytes == 4:
        maxv = 1.0
    else:
        maxv = (1 << bits) - 1
    y, x = np.mgrid[0:h, 0:w]
    if pattern == "noise":
        if bytes == 4:
            return ((r >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / (1 << 24))).astype(np.float32)
        return (r % np.uint64(maxv + 1)).astype({1: np.uint8, 2: np.uint16}[bytes])
    if pattern == "checker":
        v = (((x // 5 + y // 3) & 1) * maxv)
    elif pattern == "sine":
        v = (0.5 + 0.5 * np.sin((x * 0.043 + y * 0.071) + seed)) * maxv
    elif pattern == "edges":
        ph = ((x + (y * ((seed % 7) - 3)) // 2) // 11) & 1
        jitter = (r >> np.uint64(58)).astype(np.float64) / 64.0 * 0.08
        v = (0.15 + 0.7 * ph + jitter) * maxv
    else:
        raise ValueError(pattern)
    if bytes == 4:
        return v.astype(np.float32)
    return np.clip(np.rint(v), 0, maxv).astype({1: np.uint8, 2: np.uint16}[bytes])


def frame(cfg, pattern: str = "noise", seed: int = 0):
    """All planes of one input frame for a Config-like object (width, height, bytes, bits, planes,
    subw, subh)."""
    out = []
    for p in range(min(cfg.planes, 3)):
        h = cfg.height >> (cfg.subh if p else 0)
        w = cfg.width >> (cfg.subw if p else 0)
        out.append(plane(h, w, cfg.bytes, cfg.bits, pattern, seed * 3 + p))
    return out
