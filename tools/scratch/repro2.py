import os, sys
sys.path.insert(0, '.')
import numpy as np
os.environ["SN_PREFER_POOL"] = "0"
from avisynth_sangnom2_amd import SangNom2, clip_format
d = np.load(os.path.join(os.path.dirname(__file__), "m.npz"))
src = [d['src0'], d['src1'], d['src2']]; want = [d['want0'], d['want1'], d['want2']]
clip = clip_format("YUV422P8", 512, 240)
kw = dict(order=1, aa=128, aac=128, dh=True, luma=True, chroma=True)
for mode in ("fused", "auto", "pool"):
    for rep in range(2):
        with SangNom2(clip, mode=mode, **kw) as flt:
            outs = [flt.get_frame(src, parity=0) for _ in range(3)]
        for i, got in enumerate(outs):
            bad = [(p, np.argwhere(want[p] != got[p]).tolist()[:4]) for p in range(3) if not np.array_equal(want[p], got[p])]
            print(mode, rep, "call", i, "ok" if not bad else bad, flush=True)
