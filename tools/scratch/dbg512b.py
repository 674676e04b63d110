import os, sys
sys.path.insert(0, '.')
import numpy as np, torch
os.environ["SN_PREFER_POOL"] = "0"
from avisynth_sangnom2_amd import SangNom2, clip_format, synth
from oracle.sangnom_numpy import NumpySangNom
w, h = 512, 640
clip = clip_format("YUV420P8", w, h)
kw = dict(aa=48, aac=48)
frames = [synth.frame(clip, "noise", seed=5 + i) for i in range(4)]
n = NumpySangNom(width=w, height=h, bytes=1, bits=8, planes=3, subw=1, subh=1, **kw)
src = frames[0]
pools = []
for p in (0, 1):
    dd = np.zeros_like(src[p]); dd[0::2] = src[p][0::2]
    n._plane(dd, 0, p); pools.append(n.pool.copy())
dev = torch.device("cuda:0")
for N in (1, 2, 4):
    with SangNom2(clip, mode="fused", max_batch=N, **kw) as flt:
        s = [torch.from_numpy(np.stack([frames[f][p] for f in range(N)])).to(dev) for p in range(3)]
        d = [torch.zeros_like(t) for t in s]
        flt.process_batch(s, d, parity=[1] * N); flt.synchronize()
        g1 = flt.read_coupled_rows(1).astype(np.int64)
        print("N", N, "b0", g1[0, 1, 480:512:8].tolist(), "b1", g1[1, 1, 480:512:8].tolist(), "b4", g1[4, 1, 480:512:8].tolist())
print("exp        ", pools[1][4, 1, 488:512].tolist())
