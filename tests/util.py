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


def to_host(t) -> np.ndarray:
    """A device tensor as a numpy array, copied through a PINNED buffer -- spelt out at every call site of the GPU suite.
    Why not `t.cpu()`: on these boxes a process that has registered and unregistered host memory (the pinned-frames test,
    sn_pin_host_buffer / sn_unpin_host_buffer = hipHostRegister / hipHostUnregister) aborts now and then inside a LATER
    pageable copy of the HIP runtime, torch's included (about one fresh suite run in twenty; 0 in 44 with pinned
    transfers: profiles/r3_page_fault.md 6.).  The hazard stays visible outside the suite:
    tools/repro_pageable_after_unpin.py runs exactly that sequence in young child processes, and
    include/sangnom_hip.h marks the pin entry points accordingly."""
    import torch
    if not t.is_cuda:
        return t.numpy()
    buf = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    buf.copy_(t)
    return buf.numpy()
