import os, sys
sys.path.insert(0, '.')
import numpy as np
os.environ["SN_PREFER_POOL"] = "0"
from avisynth_sangnom2_amd import SangNom2, clip_format, synth
w, h = 512, 640
clip = clip_format("YUV420P8", w, h)
src = synth.frame(clip, "noise", seed=5)
with SangNom2(clip, mode="fused", aa=48, aac=48) as flt:
    flt.get_frame(src)
    g0 = flt.read_coupled_rows(0).astype(np.int64); g1 = flt.read_coupled_rows(1).astype(np.int64)
    print("luma b4 row 2, cols 496..511 (kind 1, threads 2,3):", g0[4, 2, 496:512].tolist())
    print("luma b4 row 2, cols 16..31  (kind 0, threads 2,3):", g0[4, 2, 16:32].tolist())
    print("U dbg b0 (kind-1 load)      :", g1[0, 1, 496:512].tolist())
    print("U dbg b1 (kind-1 load, glc) :", g1[1, 1, 496:512].tolist())
    print("U dbg b2 (kind-0 load)      :", g1[2, 1, 496:512].tolist())
