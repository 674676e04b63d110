import os, sys, itertools
sys.path.insert(0, '.')
import numpy as np
os.environ["SN_PREFER_POOL"] = "0"
from avisynth_sangnom2_amd import SangNom2, clip_format, synth
for fmt, w, dh, h, pattern, aa in itertools.product(("YUV422P8", "YUV420P8"), (512, 992), (False, True), (240, 320, 480), ("edges", "noise"), (48, 128)):
    clip = clip_format(fmt, w, h)
    kw = dict(aa=aa, aac=aa, dh=dh)
    src = synth.frame(clip, pattern, seed=5)
    outs = {}
    for mode in ("fused", "pool"):
        with SangNom2(clip, mode=mode, **kw) as flt:
            outs[mode] = flt.get_frame(src)
    res = [int((outs["fused"][p] != outs["pool"][p]).sum()) for p in range(3)]
    if any(res):
        print(fmt, w, "dh", dh, "h", h, pattern, "aa", aa, "differing per plane", res, flush=True)
print("done")
