import os, sys, itertools
sys.path.insert(0, '.')
import numpy as np
os.environ["SN_PREFER_POOL"] = "1"
from avisynth_sangnom2_amd import SangNom2, clip_format, synth
from oracle.oracle import Config, Oracle
clip = clip_format("YUV422P8", 512, 240)
kw = dict(order=1, aa=128, aac=128, dh=True, luma=True, chroma=True)
cfg = Config(width=512, height=240, bytes=1, bits=8, planes=3, subw=1, subh=0, **kw)
found = 0
for pattern, seed, bands, warm in itertools.product(["noise", "checker", "edges", "sine"], [1, 2, 3], [0, 2, 3, 5, 9, 16], [0, 1, 6, 12]):
    src = synth.frame(clip, pattern, seed=seed)
    want = Oracle(cfg).process(src, parity=1)
    with SangNom2(clip, host_depth=2, **kw) as flt:
        flt.set_bands(bands, warm)
        got = flt.get_frame(src, parity=1)
        info = flt.info()
    for p in range(3):
        if not np.array_equal(want[p], got[p]):
            d = np.argwhere(want[p] != got[p])
            print("MISMATCH", pattern, seed, bands, warm, "plane", p, "n", len(d), "rows", d[:, 0].min(), d[:, 0].max(), "cols", d[:, 1].min(), d[:, 1].max(),
                  "banded", info.banded_frames, "fallbacks", info.band_fallbacks, flush=True)
            found += 1
    if found >= 6:
        break
print("done, found", found)
