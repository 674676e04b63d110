import os, sys
sys.path.insert(0, '.')
import numpy as np
os.environ["SN_PREFER_POOL"] = "0"
from avisynth_sangnom2_amd import SangNom2, clip_format, synth
from oracle.sangnom_numpy import NumpySangNom
for fmt, subh in (("YUV420P8", 1), ("YUV422P8", 0)):
  for w in (256, 480, 496, 512, 528, 544, 640, 960, 976, 992, 1008, 1024, 1472, 1920, 2432):
    h = 640 if subh else 320
    clip = clip_format(fmt, w, h)
    kw = dict(aa=48, aac=48)
    src = synth.frame(clip, "noise", seed=5)
    n = NumpySangNom(width=w, height=h, bytes=1, bits=8, planes=3, subw=1, subh=subh, **kw)
    pools = []
    for p in (0, 1):
        dd = np.zeros_like(src[p]); dd[0::2] = src[p][0::2]
        n._plane(dd, 0, p); pools.append(n.pool.copy())
    try:
        flt = SangNom2(clip, mode="fused", **kw)
    except Exception as e:
        print(fmt, w, "n/a", str(e)[:40]); continue
    with flt:
        flt.get_frame(src)
        rows = flt.info().coupled_rows
        res = []
        hc = h >> subh
        nr_c, w_c = hc // 2 - 1, w // 2
        for which in (0, 1):
            g = flt.read_coupled_rows(which).astype(np.int64)
            e = pools[which][:, :rows, :w]
            q = np.arange(rows)[:, None]; x = np.arange(w)[None, :]
            extra = 6 if which == 0 else 0
            last = rows - 1 if which == 0 else min(nr_c + 1, (h + 1) // 2 - 1)
            cone = (q >= 1) & (q <= last) & (x < w_c + 3 * (nr_c - q + 2) + extra) & ((x >= w_c) | (q > nr_c))
            bad = np.argwhere((g != e) & cone[None])
            res.append((len(bad), bad[0].tolist() if len(bad) else None))
        print(fmt, w, "lanes", w // 8, res, flush=True)
