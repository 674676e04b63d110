import os, sys
sys.path.insert(0, '.')
import numpy as np
os.environ["SN_PREFER_POOL"] = "0"
from avisynth_sangnom2_amd import SangNom2, clip_format, synth
w, h = 512, 640
clip = clip_format("YUV420P8", w, h)
kw = dict(aa=48, aac=48)
src = synth.frame(clip, "noise", seed=5)
with SangNom2(clip, mode="fused", **kw) as flt:
    for i in range(3):
        flt.get_frame(src)
        g1 = flt.read_coupled_rows(1).astype(np.int64)
        print("call", i, "U row 1 b4", g1[4, 1, 484:512].tolist())
