import os, sys
sys.path.insert(0, '.')
import numpy as np
os.environ["SN_PREFER_POOL"] = "0"
os.environ["SN_RAW_DUMP"] = "gpurun_out/raw_pools.bin"
from avisynth_sangnom2_amd import SangNom2, clip_format, synth
w, h = 512, 640
clip = clip_format("YUV420P8", w, h)
src = synth.frame(clip, "noise", seed=5)
os.makedirs("gpurun_out", exist_ok=True)
with SangNom2(clip, mode="fused", aa=48, aac=48) as flt:
    flt.get_frame(src)
    rows = flt.info().coupled_rows
    g0 = flt.read_coupled_rows(0); g1 = flt.read_coupled_rows(1)
raw = np.fromfile("gpurun_out/raw_pools.bin", dtype=np.uint8)
per = 9 * rows * 1024
print("rows", rows, "raw bytes", raw.size, "expected", 2 * per)
p0 = raw[:per].reshape(9, rows, 1024)
print("pool0 b4 row 2 kind0 t60..63:", p0[4, 2, 480:512].tolist())
print("pool0 b4 row 2 kind1 t0..5 :", p0[4, 2, 512:560].tolist())
nz = np.argwhere(p0[4, 2] != 0)
print("nonzero byte offsets in that row:", nz.min(), nz.max(), "count", len(nz))
