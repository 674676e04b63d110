import os, sys
sys.path.insert(0, '.')
import numpy as np
os.environ["SN_PREFER_POOL"] = "0"
from avisynth_sangnom2_amd import SangNom2, clip_format, synth
from oracle.sangnom_numpy import NumpySangNom
w, h = 512, 640
clip = clip_format("YUV420P8", w, h)
kw = dict(aa=48, aac=48)
src = synth.frame(clip, "noise", seed=5)
n = NumpySangNom(width=w, height=h, bytes=1, bits=8, planes=3, subw=1, subh=1, **kw)
pools = []
for p in (0, 1):
    dd = np.zeros_like(src[p]); dd[0::2] = src[p][0::2]
    n._plane(dd, 0, p); pools.append(n.pool.copy())
with SangNom2(clip, mode="fused", **kw) as flt:
    flt.get_frame(src)
    g0 = flt.read_coupled_rows(0).astype(np.int64); g1 = flt.read_coupled_rows(1).astype(np.int64)
for r in (1, 2, 3, 5, 6, 7, 11, 12, 40, 100, 160):
    print("row", r, "U got", g1[4, r, 476:512].tolist())
    print("row", r, "U exp", pools[1][4, r, 476:512].tolist())
bad = np.argwhere(g1[:, 1:161, :] != pools[1][:, 1:161, :w])
print("bad count", len(bad), "cols min", bad[:, 2].min() if len(bad) else None, "rows", sorted(set(bad[:, 1].tolist()))[:20])
