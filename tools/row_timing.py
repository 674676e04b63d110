#!/usr/bin/env python3
"""Where a wave of the plain 8-bit sweep spends its cycles, by phase of a row (a library whose plain translation unit was built
with -DSN_ROW_TIMING: `make -C avisynth_sangnom2_amd/csrc EXTRA=-DSN_ROW_TIMING sn_fused_u8_v3_plain.o -B` and relink):
    SN_LIB=ab/row_timing.so python3 tools/row_timing.py [workload ...]
s_memtime around the phases, summed over all waves of a launch; compares the 4-wave instance (two workgroups per CU) with the
8-wave one (one workgroup per CU)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avisynth_sangnom2_amd import SangNom2, capi, clip_format  # noqa: E402

if os.environ.get("SN_LIB"):
    capi.LIB_PATH = os.path.abspath(os.environ["SN_LIB"])
lib = capi.load()
lib.sn_debug_row_cycles.argtypes = [ctypes.POINTER(ctypes.c_ulonglong * 8), ctypes.c_int]
NAMES = ("wait for the prefetched line", "unpack, windows, park, kept-line store, prefetch", "seam barrier (every 5th row)",
         "ghost refresh + row set-up", "nine buffer steps + stage 3 + store", "publish (every 5th row)", "turn-taking / loop")

dev = torch.device("cuda:0")
if os.environ.get("SN_ROW_TIMING_COUPLED"):
    # the library's OTHER object of sn_fused_u8_v3.hip carries the counters (-DSN_ROW_TIMING -DSN_ROW_TIMING_COUPLED for
    # sn_fused_u8_v3.o): the luma sweep of a 4:2:0 clip (kLumaSpill); U and V run in sn_fused_u8_uv.hip and are not counted
    clip = clip_format("YUV420P8", 3840, 2160)
    n = 512
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    src = [torch.randint(0, 256, (n, 2160 >> (1 if p else 0), 3840 >> (1 if p else 0)), device=dev, generator=g, dtype=torch.uint8) for p in range(3)]
    dst = [torch.empty_like(t) for t in src]
    with SangNom2(clip, max_batch=n, aa=48, aac=48) as flt:
        flt.process_batch(src, dst)
        flt.synchronize()
        out = (ctypes.c_ulonglong * 8)()
        assert lib.sn_debug_row_cycles(ctypes.byref(out), 1) == 0
        flt.process_batch(src, dst)
        flt.synchronize()
        assert lib.sn_debug_row_cycles(ctypes.byref(out), 1) == 0
    rows = max(out[7], 1)
    tot = sum(out[k] for k in range(7))
    print(f"2160p-YUV420P8 luma sweep (kLumaSpill): {n} frames; {rows} wave-rows, {tot / rows:.0f} s_memtime ticks per wave-row")
    for k in range(7):
        print(f"    [{k}] {NAMES[k]:52s} {out[k] / rows:8.1f}  {100.0 * out[k] / tot:5.1f} %")
    sys.exit(0)
for wl, (w, h, n) in {"1080p-Y8": (1920, 1080, 2048), "2160p-Y8": (3840, 2160, 1024), "4320p-Y8": (7680, 4320, 256)}.items():
    if len(sys.argv) > 1 and wl not in sys.argv[1:]:
        continue
    clip = clip_format("Y8", w, h)
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    src = [torch.randint(0, 256, (n, h, w), device=dev, generator=g, dtype=torch.uint8)]
    dst = [torch.empty_like(src[0])]
    with SangNom2(clip, max_batch=n, aa=48) as flt:
        flt.process_batch(src, dst)
        flt.synchronize()
        out = (ctypes.c_ulonglong * 8)()
        assert lib.sn_debug_row_cycles(ctypes.byref(out), 1) == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(torch.cuda.ExternalStream(flt.stream_handle()))
        flt.process_batch(src, dst)
        e1.record(torch.cuda.ExternalStream(flt.stream_handle()))
        flt.synchronize()
        assert lib.sn_debug_row_cycles(ctypes.byref(out), 1) == 0
    rows = max(out[7], 1)
    tot = sum(out[k] for k in range(7))
    print(f"{wl}: {n} frames in {e0.elapsed_time(e1):.2f} ms; {rows} wave-rows, {tot / rows:.0f} s_memtime ticks per wave-row")
    for k in range(7):
        print(f"    [{k}] {NAMES[k]:52s} {out[k] / rows:8.1f}  {100.0 * out[k] / tot:5.1f} %")
