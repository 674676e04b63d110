#!/usr/bin/env python3
"""Which wave of a coupled 16-bit / float sweep does the workgroup wait for?  Per mode (1 luma with hand-off, 2 U, 4 V) and wave
index: shader-clock ticks between a wave's start and its end, and the part of them spent at the seam barriers (a library whose
sn_fused_u16_v3.o and sn_fused_f32_v3.o were built with -DSN_WAVE_TIMING:
`make -C avisynth_sangnom2_amd/csrc EXTRA=-DSN_WAVE_TIMING -B sn_fused_u16_v3.o sn_fused_f32_v3.o && make -C avisynth_sangnom2_amd/csrc`,
keep the result as ab/wave_timing.so and rebuild the two objects without the flag):
    SN_LIB=ab/wave_timing.so python3 tools/wave_timing.py [frames = 256]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avisynth_sangnom2_amd import SangNom2, capi, clip_format  # noqa: E402

if os.environ.get("SN_LIB"):
    capi.LIB_PATH = os.path.abspath(os.environ["SN_LIB"])
lib = capi.load()
Out = ctypes.c_ulonglong * (5 * 8 * 2)
lib.sn_debug_wave_ticks.argtypes = [ctypes.POINTER(Out), ctypes.c_int]

lib.sn_debug_wave_ticks_f32.argtypes = [ctypes.POINTER(Out), ctypes.c_int]
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
for fmt, fn in (("YUV420P16", lib.sn_debug_wave_ticks), ("YUV420PS", lib.sn_debug_wave_ticks_f32)):
    clip = clip_format(fmt, 3840, 2160)
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    shape = lambda p: (n, 2160 >> (1 if p else 0), 3840 >> (1 if p else 0))
    if clip.bytes == 2:
        src = [torch.randint(0, 32767, shape(p), device=dev, generator=g, dtype=torch.int16) for p in range(3)]
    else:
        src = [torch.rand(shape(p), device=dev, generator=g, dtype=torch.float32) for p in range(3)]
    dst = [torch.empty_like(t) for t in src]
    with SangNom2(clip, max_batch=n, aa=48, aac=48) as flt:
        flt.process_batch(src, dst)
        flt.synchronize()
        out = Out()
        assert fn(ctypes.byref(out), 1) == 0
        flt.process_batch(src, dst)
        flt.synchronize()
        assert fn(ctypes.byref(out), 1) == 0
    del src, dst
    print(f"2160p {fmt}, {n} frames: ticks per frame (in the kernel / of them at barriers / working)")
    for mode, name in ((1, "luma + hand-off"), (2, "U"), (4, "V")):
        print(f"  mode {mode} ({name})")
        for w in range(8):
            tot, bar = out[(mode * 8 + w) * 2] / n, out[(mode * 8 + w) * 2 + 1] / n
            print(f"    wave {w}: {tot:10.0f} {bar:10.0f} {tot - bar:10.0f}")
