import os, sys
sys.path.insert(0, '.')
import numpy as np
os.environ["SN_PREFER_POOL"] = "0"
from avisynth_sangnom2_amd import SangNom2, clip_format
from oracle.sangnom_numpy import NumpySangNom
d = np.load(os.path.join(os.path.dirname(__file__), "m.npz"))
src = [d['src0'], d['src1'], d['src2']]
w, h = 512, 240
clip = clip_format("YUV422P8", w, h)
kw = dict(order=1, aa=128, aac=128, dh=True, luma=True, chroma=True)
n = NumpySangNom(width=w, height=h, bytes=1, bits=8, planes=3, subw=1, subh=0, **kw)
pools = []
for p in (0, 1, 2):
    s = src[p]
    dd = np.zeros((s.shape[0] * 2, s.shape[1]), dtype=s.dtype)
    dd[0::2] = s
    dd[-1] = dd[-2]
    n._plane(dd, 0, p)
    pools.append(n.pool.copy())
with SangNom2(clip, mode="fused", **kw) as flt:
    got = flt.get_frame(src, parity=0)
    rows = flt.info().coupled_rows
    print("coupled rows", rows)
    for which in (0, 1):
        g = flt.read_coupled_rows(which).astype(np.int64)
        e = pools[which][:, :rows, :w]
        nr_c, w_c = 239, 256
        q = np.arange(rows)[:, None]; x = np.arange(w)[None, :]
        extra = 6 if which == 0 else 0
        cone = (q >= 1) & (x < w_c + 3 * (nr_c - q + 2) + extra) & ((x >= w_c) | (q > nr_c))
        bad = np.argwhere((g != e) & cone[None])
        print("hand-off", which, "bad cells in cone:", len(bad), bad[:10].tolist())
        if len(bad):
            b, r, c = bad[0]
            print(" got", g[b, r, c-4:c+5].tolist(), "exp", e[b, r, c-4:c+5].tolist())
    g0 = flt.read_coupled_rows(0).astype(np.int64); g1 = flt.read_coupled_rows(1).astype(np.int64)
    for b in (0, 4):
        for r in (1, 2):
            print("luma  b", b, "row", r, "got", g0[b, r, 484:512].tolist())
            print("luma  b", b, "row", r, "exp", pools[0][b, r, 484:512].tolist())
        print("U     b", b, "row 1 got", g1[b, 1, 484:512].tolist())
        print("U     b", b, "row 1 exp", pools[1][b, 1, 484:512].tolist())
