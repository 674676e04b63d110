import os, sys
sys.path.insert(0, '.')
import numpy as np
os.environ["SN_PREFER_POOL"] = "0"
from avisynth_sangnom2_amd import SangNom2, clip_format, synth
for fmt in ("YUV420P8", "YUV422P8", "YUV420P16", "YUV420PS"):
  for w in (256, 480, 512, 544, 640, 960, 992, 1024, 1472, 1504, 1952, 1984):
    h = 640 if fmt != "YUV422P8" else 320
    clip = clip_format(fmt, w, h)
    kw = dict(aa=48, aac=48)
    src = synth.frame(clip, "edges", seed=5)
    outs = {}
    for mode in ("fused", "pool"):
        try:
            with SangNom2(clip, mode=mode, **kw) as flt:
                outs[mode] = flt.get_frame(src)
        except Exception as e:
            outs[mode] = None
    if outs["fused"] is None:
        print(fmt, w, "n/a"); continue
    res = []
    for p in range(3):
        a, b = outs["fused"][p].view(np.uint8), outs["pool"][p].view(np.uint8)
        d = np.argwhere(a != b)
        res.append(len(d))
    print(fmt, w, "lanes", w // 8, "differing bytes per plane", res, flush=True)
