import sys
sys.path.insert(0, '.')
from avisynth_sangnom2_amd import SangNom2, clip_format, synth
for (w, h, nb) in ((256, 400, 4), (3840, 2160, 16)):
    clip = clip_format("YUV420P8", w, h)
    src = synth.frame(clip, "noise", seed=1)
    for warm in (32, 64, 128):
        with SangNom2(clip, aac=48) as flt:
            flt.set_bands(nb, warm)
            flt.get_frame(src)
            print(w, h, "warm", warm, "fallbacks", flt.info().band_fallbacks, file=sys.stderr)
