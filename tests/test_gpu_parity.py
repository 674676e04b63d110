"""GPU parity: libsangnom_hip.so (through the C ABI) against the CPU oracle, bit-exact.

Reads like the tests the reference never had: construct SangNom2(clip, order, aa, aac, ..., dh,
luma, chroma), request frames, compare with opt=0 semantics
(/root/reference/src/SangNom2.cpp:259-273, :332-397).  Float planes are compared on their bit
patterns (north_star allows 1 ulp; we hold 0).
"""
from contextlib import nullcontext as _nullcontext

import numpy as np
import pytest

from avisynth_sangnom2_amd import ClipFormat, SangNom, SangNom2, SangNomAA, SangNomAAHost, SangNomError, clip_format, synth
from oracle.oracle import Oracle
from oracle.sangnom_numpy import NumpySangNom
from tests.util import describe_diff, make_frames, oracle_cfg, same, to_host


@pytest.fixture(autouse=True)
def _sweeps_unless_asked_otherwise(request, monkeypatch):
    """Auto mode hands launches of a few frames to the pool path; these tests want the fused sweeps wherever a
    configuration is eligible, so that policy is off here except in the tests that are about it."""
    if "small_launch_policy" not in request.keywords:
        from avisynth_sangnom2_amd import capi
        monkeypatch.setitem(capi.POLICY_DEFAULTS, "small_launches", capi.SN_SMALL_SWEEP)

pytestmark = pytest.mark.gpu

PATTERNS = ("noise", "checker", "sine", "edges")

# (pixel_type, width, height, filter kwargs)
CASES = [
    ("Y8", 64, 32, {}),
    ("Y8", 256, 64, dict(order=2)),
    ("Y8", 192, 48, dict(order=0, aa=128)),
    ("Y8", 100, 40, dict(aa=1)),                       # width % 32 != 0: pool keeps state
    ("Y8", 1056, 34, dict(aa=0)),                      # wider than one smoothing pass of 1024 threads
    ("Y10", 96, 36, dict(aa=20)),
    ("Y16", 128, 64, {}),
    ("Y16", 72, 32, dict(order=2, aa=77)),
    ("Y32", 64, 48, {}),
    ("Y32", 40, 32, dict(order=2, aa=10)),
    ("YUV420P8", 128, 64, dict(aac=48)),
    ("YUV420P8", 96, 32, dict(aac=0, order=2)),
    ("YUV420P8", 128, 32, dict(chroma=False)),
    ("YUV420P8", 128, 32, dict(luma=False, aac=33)),   # chroma sees the previous frame's pool
    ("YUV420P16", 128, 64, dict(aac=48)),
    ("YUV422P8", 64, 32, dict(aac=48)),
    ("YUV444P8", 64, 32, dict(aac=100, order=0)),
    ("YUV444PS", 64, 24, dict(dh=True, aac=48)),
    ("YUV420PS", 128, 64, dict(aac=48)),
    ("YUV420P8", 64, 32, dict(dh=True, aac=48)),
    ("YUV420P16", 80, 32, dict(dh=True, luma=False, aac=5)),
]


def _run_case(fmt, w, h, kw, pattern, nframes=3, mode="auto"):
    clip = clip_format(fmt, w, h)
    ora = Oracle(oracle_cfg(clip, **kw))
    with SangNom2(clip, device=0, mode=mode, **kw) as flt:
        for f, src in enumerate(make_frames(clip, pattern, nframes, seed0=7)):
            parity = f & 1
            want = ora.process(src, parity=parity)
            got = flt.get_frame(src, parity=parity)
            for p in range(len(want)):
                assert same(want[p], got[p]), (
                    f"{fmt} {w}x{h} {kw} {pattern} frame {f} plane {p}: " + describe_diff(want[p], got[p]))


@pytest.mark.parametrize("fmt,w,h,kw", CASES, ids=[f"{c[0]}-{c[1]}x{c[2]}-{i}" for i, c in enumerate(CASES)])
@pytest.mark.parametrize("pattern", PATTERNS)
def test_host_frames_match_oracle(hip_lib, fmt, w, h, kw, pattern):
    _run_case(fmt, w, h, kw, pattern)


@pytest.mark.parametrize("fmt,w,h,kw", CASES[:4] + CASES[10:13], ids=lambda v: str(v) if not isinstance(v, dict) else "kw")
def test_pool_mode_matches_oracle(hip_lib, fmt, w, h, kw):
    _run_case(fmt, w, h, kw, "noise", mode="pool")


def test_pool_contents_match_oracle(hip_lib):
    """The smoothed pool itself (stage 1 + 2) is bit-identical, including rows 0 / bh and stale cells."""
    clip = clip_format("YUV420P8", 96, 32)
    ora = Oracle(oracle_cfg(clip, aac=48))
    with SangNom2(clip, aac=48, mode="pool") as flt:
        for f, src in enumerate(make_frames(clip, "noise", 2)):
            ora.process(src)
            flt.get_frame(src)
            assert np.array_equal(ora.pool(), flt.read_pool(0)), f"pool differs after frame {f}"


def test_pitched_host_planes(hip_lib):
    """Arbitrary, differing src/dst pitches (AviSynth frames are pitched); padding is left alone."""
    clip = clip_format("YUV420P8", 96, 32)
    kw = dict(aac=48, order=2)
    ora = Oracle(oracle_cfg(clip, **kw))
    src = synth.frame(clip, "edges", seed=3)
    want = ora.process(src)
    with SangNom2(clip, **kw) as flt:
        psrc, pdst = [], []
        for p, s in enumerate(src):
            big = np.full((s.shape[0], s.shape[1] + 13 + 16 * p), 0xAB, dtype=s.dtype)
            big[:, :s.shape[1]] = s
            psrc.append(big[:, :s.shape[1]])
            d = np.full((flt.plane_shape_out(p)[0], s.shape[1] + 64 - 7 * p), 0xCD, dtype=s.dtype)
            pdst.append(d[:, :s.shape[1]])
        got = flt.get_frame(psrc, dst=pdst)
        for p in range(3):
            assert same(want[p], got[p])
            assert (pdst[p].base[:, want[p].shape[1]:] == 0xCD).all(), "dst padding was overwritten"


def test_device_batch_matches_oracle(hip_lib):
    """sn_process_device_strided on device-resident frames == per-frame oracle (history-free config)."""
    import torch
    clip = clip_format("YUV420P8", 128, 64)
    kw = dict(aac=48)
    N = 5
    frames = make_frames(clip, "noise", N, seed0=11)
    dev = torch.device("cuda:0")
    with SangNom2(clip, max_batch=N, **kw) as flt:
        assert flt.info().history_free == 1
        src = [torch.from_numpy(np.stack([fr[p] for fr in frames])).pin_memory().to(dev) for p in range(3)]
        dst = [torch.zeros((N,) + flt.plane_shape_out(p), dtype=src[p].dtype, device=dev) for p in range(3)]
        torch.cuda.synchronize()
        parity = [1, 0, 1, 1, 0]
        flt.process_batch(src, dst, parity)
        flt.synchronize()
        for f in range(N):
            ora = Oracle(oracle_cfg(clip, **kw))  # fresh instance per frame: history-free
            want = ora.process(frames[f], parity=parity[f])
            for p in range(3):
                assert same(want[p], to_host(dst[p][f])), f"frame {f} plane {p}"


def test_device_batch_history_carrying(hip_lib):
    """width % 32 != 0: a batch must behave like sequential frames on ONE instance (pool state carries)."""
    import torch
    clip = clip_format("Y8", 100, 40)
    N = 4
    frames = make_frames(clip, "noise", N, seed0=5)
    dev = torch.device("cuda:0")
    ora = Oracle(oracle_cfg(clip))
    with SangNom2(clip, max_batch=N) as flt:
        assert flt.info().history_free == 0
        src = [torch.from_numpy(np.stack([fr[0] for fr in frames])).pin_memory().to(dev)]
        dst = [torch.zeros((N,) + flt.plane_shape_out(0), dtype=torch.uint8, device=dev)]
        flt.process_batch(src, dst)
        flt.synchronize()
        for f in range(N):
            want = ora.process(frames[f])
            assert same(want[0], to_host(dst[0][f])), f"frame {f}"


def test_validation_errors_are_the_references(hip_lib):
    """Create_SangNom2's checks and message text, /root/reference/src/SangNom2.cpp:407-422."""
    cases = [
        (clip_format("Y8", 64, 31), {}, "SangNom2: height must be even."),
        (clip_format("YUV420P8", 64, 34), {}, "SangNom2: height must be mod4."),
        (clip_format("Y8", 64, 32), dict(order=3), "SangNom2: order must be between 0..2."),
        (clip_format("Y8", 64, 32), dict(aa=129), "SangNom2: aa must be between 0..128."),
        (clip_format("Y8", 64, 32), dict(aac=-1), "SangNom2: aac must be between 0..128."),
        (clip_format("Y8", 64, 32), dict(opt=2), "SangNom2: opt must be between -1..2."),
    ]
    for clip, kw, text in cases:
        with pytest.raises(SangNomError) as ei:
            SangNom2(clip, **kw)
        assert str(ei.value) == text


def test_flat_and_extreme_inputs(hip_lib):
    """Known answers: a flat plane stays flat; all-max and all-zero planes survive the wrap paths."""
    for fmt, vals in (("Y8", (0, 255, 128)), ("Y16", (0, 65535, 40000)), ("Y32", (0.0, 1.0, 0.25))):
        clip = clip_format(fmt, 64, 32)
        with SangNom2(clip) as flt:
            for v in vals:
                src = [np.full((32, 64), v, dtype=clip.dtype)]
                out = flt.get_frame(src)
                assert (out[0] == src[0]).all()


# ---- the fused sweeps (sn_fused_{u8,u16,f32}_v3.hip) -------------------------------------------------

FUSED_CASES = [
    # (fmt, w, h, kw): widths chosen to hit 1, 2, 3 and 8 waves per workgroup and mid-wave right edges
    ("Y8", 32, 8, {}),
    ("Y8", 512, 20, {}),                 # 64 lanes: one full wave, no ghosts
    ("Y8", 544, 24, dict(order=2)),      # 68 lanes: 2 waves, second almost empty
    ("Y8", 992, 26, {}),                 # 124 lanes
    ("Y8", 1024, 32, dict(aa=100)),      # 128 lanes -> 3 waves
    ("Y8", 1920, 36, dict(order=0)),
    ("Y8", 3840, 28, {}),
    ("Y8", 3840, 44, dict(order=2)),     # widest the fused kernel takes (8 waves); > 8 seam refreshes
    ("Y8", 2048, 64, dict(order=2, aa=90)),
    ("Y8", 4096, 24, {}),                # 5 waves (9 virtual wavefronts, last half-wave dead)
    ("Y8", 5120, 22, dict(order=2)),
    ("Y8", 7680, 26, {}),                # 8 waves, 119 KB of LDS
    ("Y8", 256, 2, {}),                  # nothing to interpolate
    ("Y8", 256, 4, dict(order=2)),       # a single interpolated row
    ("Y8", 256, 6, {}),
    ("Y8", 640, 40, dict(dh=True)),
    ("YUV444P8", 576, 24, dict(aac=48)),
    ("YUV420P8", 576, 24, dict(chroma=False)),  # luma fused, chroma copied
    # subsampled chroma: luma and chroma sweeps coupled through the scratch pools (shared-pool emulation)
    ("YUV420P8", 64, 32, dict(aac=48)),
    ("YUV420P8", 128, 64, dict(aac=30, order=2)),
    ("YUV420P8", 1024, 40, dict(aac=48)),          # chroma region ends inside a strip
    ("YUV420P8", 1984, 24, dict(aac=48, order=0)),
    ("YUV420P8", 3840, 48, dict(aac=48)),
    ("YUV420P8", 96, 8, dict(aac=48)),
    ("YUV422P8", 512, 240, dict(dh=True, aa=128, aac=128)),   # found by tools/fuzz.py: the last 16 columns of a 512-wide pool
    ("YUV420P8", 512, 400, dict(aa=128, aac=128)),             # a single interpolated chroma row
    ("YUV420P8", 96, 4, dict(aac=48)),             # chroma planes too short to interpolate
    ("YUV422P8", 576, 28, dict(aac=48)),           # chroma as tall as luma
    ("YUV420P8", 320, 20, dict(dh=True, aac=48)),
    ("YUV420P8", 320, 36, dict(aac=0)),            # aac = 0: only exact-zero minima leave the vertical average
]


FUSED_CASES += [
    # 9..16-bit samples: sn_fused_u16_v3.hip (one pixel per register, up to 3840 wide)
    ("Y16", 64, 32, {}),
    ("Y16", 512, 20, dict(order=2)),
    ("Y16", 544, 24, dict(aa=100)),
    ("Y10", 1024, 28, dict(aa=20)),
    ("Y16", 1920, 24, dict(order=0)),
    ("Y16", 3840, 26, {}),
    ("Y16", 256, 4, {}),
    ("Y12", 320, 40, dict(dh=True)),
    ("YUV444P16", 576, 24, dict(aac=48)),
    ("YUV420P16", 128, 64, dict(aac=48)),
    ("YUV420P10", 1024, 40, dict(aac=30, order=2)),
    ("YUV420P16", 3840, 48, dict(aac=48)),
    ("YUV422P16", 576, 28, dict(aac=48)),
    ("YUV420P16", 96, 8, dict(aac=48)),
]


FUSED_CASES += [
    # float samples: sn_fused_f32_v3.hip (bit patterns must agree)
    ("Y32", 64, 32, {}),
    ("Y32", 512, 20, dict(order=2)),
    ("Y32", 544, 24, dict(aa=100)),
    ("Y32", 1024, 28, dict(aa=20)),
    ("Y32", 1536, 30, {}),
    ("Y32", 1920, 24, dict(order=0)),
    ("Y32", 2880, 22, {}),
    ("Y32", 3840, 26, {}),
    ("Y32", 3840, 44, dict(order=2, aa=5)),
    ("Y32", 256, 4, {}),
    ("Y32", 256, 2, {}),
    ("Y32", 320, 40, dict(dh=True)),
    ("YUV444PS", 576, 24, dict(aac=48)),
    ("YUV444PS", 64, 24, dict(dh=True, aac=48)),
    ("YUV420PS", 128, 32, dict(chroma=False)),
    # float with subsampled chroma: the hand-off pools carry float bit patterns
    ("YUV420PS", 128, 64, dict(aac=48)),
    ("YUV420PS", 1024, 40, dict(aac=3, order=2)),
    ("YUV420PS", 3840, 48, dict(aac=48)),
    ("YUV422PS", 576, 28, dict(aac=48)),
    ("YUV420PS", 96, 8, dict(aac=48)),
    ("YUV420PS", 64, 32, dict(dh=True, aac=48)),
]


@pytest.mark.parametrize("fmt,w,h,kw", FUSED_CASES, ids=[f"{c[0]}-{c[1]}x{c[2]}" for c in FUSED_CASES])
@pytest.mark.parametrize("pattern", ["noise", "checker", "edges"])
def test_fused_kernel_matches_oracle(hip_lib, fmt, w, h, kw, pattern):
    clip = clip_format(fmt, w, h)
    ora = Oracle(oracle_cfg(clip, **kw))
    with SangNom2(clip, mode="fused", **kw) as flt:
        assert flt.info().fused_eligible == 1
        for f, src in enumerate(make_frames(clip, pattern, 2, seed0=21)):
            want = ora.process(src, parity=f & 1)
            got = flt.get_frame(src, parity=f & 1)
            for p in range(len(want)):
                assert same(want[p], got[p]), f"frame {f} plane {p}: " + describe_diff(want[p], got[p])
        assert flt.info().fused_frames == 2


def test_fused_not_eligible_is_reported(hip_lib):
    for fmt, w, h, kw in (("YUV420PS", 64, 32, dict(luma=False, aac=48)), ("Y32", 3872, 16, {}), ("Y16", 3872, 16, {}), ("Y8", 100, 40, {}), ("YUV420P8", 64, 32, dict(luma=False, aac=1)),
                          ("Y8", 7712, 16, {})):
        with pytest.raises(SangNomError, match="not eligible"):
            SangNom2(clip_format(fmt, w, h), mode="fused", **kw)


def test_fused_equals_pool_at_full_size(hip_lib):
    """2160p Y8 (BASELINE.json's metric configuration): the fused kernel and the pool path agree on
    whole batches of frames, and one frame of the batch is checked against the CPU oracle."""
    import torch
    clip = clip_format("Y8", 3840, 2160)
    N = 3
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(99)
    src = [torch.randint(0, 256, (N, 2160, 3840), device=dev, generator=g, dtype=torch.uint8)]
    src[0][1] = torch.from_numpy(synth.plane(2160, 3840, 1, 8, "checker", 1)).pin_memory().to(dev)
    outs = {}
    for mode in ("fused", "pool"):
        with SangNom2(clip, max_batch=N, mode=mode) as flt:
            dst = [torch.zeros((N, 2160, 3840), device=dev, dtype=torch.uint8)]
            torch.cuda.synchronize()
            flt.process_batch(src, dst, parity=[1, 1, 1])
            flt.synchronize()
            outs[mode] = to_host(dst[0])
    assert np.array_equal(outs["fused"], outs["pool"])
    want = Oracle(oracle_cfg(clip)).process([to_host(src[0][2])])
    assert same(want[0], outs["fused"][2])


# ---- BASELINE.json's configurations at full size ----------------------------------------------------
# The oracle needs ~0.1-0.5 s per full-size frame, so each configuration checks ONE frame against it and
# a small batch through a size-independent property: the fused kernel and the pool path (two independent
# GPU implementations; the pool path mirrors the reference's three stages and its shared pool literally)
# must agree bit for bit on every frame of the batch.

FULL_SIZE = [
    ("1080p Y8", "Y8", 1920, 1080, dict(order=1, aa=48), 3),
    ("2160p YUV420P8", "YUV420P8", 3840, 2160, dict(order=1, aa=48, aac=48), 2),
    ("4320p Y8", "Y8", 7680, 4320, dict(order=1, aa=48), 2),
]


@pytest.mark.parametrize("name,fmt,w,h,kw,N", FULL_SIZE, ids=[c[0] for c in FULL_SIZE])
def test_full_size_fused_equals_pool_and_oracle(hip_lib, name, fmt, w, h, kw, N):
    import torch
    clip = clip_format(fmt, w, h)
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(2026)
    shapes = [(h >> (clip.subh if p else 0), w >> (clip.subw if p else 0)) for p in range(clip.planes)]
    src = [torch.randint(0, 256, (N,) + sh, device=dev, generator=g, dtype=torch.uint8) for sh in shapes]
    for p, sh in enumerate(shapes):  # one hard 0/255 checker frame: maximises the wrap paths of stage 2
        src[p][N - 1] = torch.from_numpy(synth.plane(sh[0], sh[1], 1, 8, "checker", p)).pin_memory().to(dev)
    outs = {}
    for mode in ("fused", "pool"):
        with SangNom2(clip, max_batch=N, mode=mode, **kw) as flt:
            dst = [torch.zeros((N,) + sh, device=dev, dtype=torch.uint8) for sh in shapes]
            torch.cuda.synchronize()
            flt.process_batch(src, dst)
            flt.synchronize()
            outs[mode] = [to_host(d) for d in dst]
    for p in range(clip.planes):
        assert np.array_equal(outs["fused"][p], outs["pool"][p]), f"{name}: plane {p} differs between fused and pool"
    want = Oracle(oracle_cfg(clip, **kw)).process([to_host(s[0]) for s in src])
    for p in range(clip.planes):
        assert same(want[p], outs["fused"][p][0]), f"{name}: plane {p} differs from the oracle"


@pytest.mark.parametrize("fmt,w,h,kw", [("YUV420P16", 3840, 2160, dict(aa=48, aac=48)),
                                        ("YUV420P16", 3840, 1080, dict(aa=48, aac=48, dh=True)),
                                        ("YUV444PS", 3840, 1080, dict(aa=48, aac=48, dh=True)),
                                        ("YUV420PS", 3840, 2160, dict(aa=48, aac=48))],
                         ids=["2160p YUV420P16", "2160p-out YUV420P16 dh", "2160p-out YUV444PS dh", "2160p YUV420PS"])
def test_full_size_16bit_and_float_match_oracle(hip_lib, fmt, w, h, kw):
    """BASELINE configuration 4 at full size, one frame each, against the oracle (float: bit patterns)."""
    clip = clip_format(fmt, w, h)
    src = synth.frame(clip, "noise", seed=77)
    want = Oracle(oracle_cfg(clip, **kw)).process(src)
    with SangNom2(clip, **kw) as flt:
        got = flt.get_frame(src)
    for p in range(3):
        assert same(want[p], got[p]), f"{fmt} plane {p}: " + describe_diff(want[p], got[p])


COUPLED = [("YUV420P8", 256, 64), ("YUV420P8", 1920, 1080), ("YUV420P8", 3840, 2160), ("YUV420P8", 7680, 360),
           ("YUV420P10", 1920, 1080), ("YUV420P16", 640, 48), ("YUV420P16", 3840, 2160),
           ("YUV420PS", 640, 48), ("YUV420PS", 1920, 1080), ("YUV420PS", 3840, 2160),
           # 64 + 60 k lanes: lanes 62, 63 of the last strip own columns (they are ghosts everywhere else), and the frame is
           # tall enough for the dependency cone to reach them
           ("YUV420P8", 512, 640), ("YUV420P8", 992, 720), ("YUV420P16", 512, 640), ("YUV420PS", 512, 640)]


@pytest.mark.parametrize("fmt,w,h", COUPLED, ids=[f"{c[0]}-{c[1]}x{c[2]}" for c in COUPLED])
def test_fused_420_hand_off_rows_match_the_shared_pool(hip_lib, fmt, w, h):
    """What the luma sweep leaves for U and the U sweep for V is, sample for sample, what the reference's
    shared pool holds at those points (src/SangNom2.cpp:322-329): the final planes alone cannot show a wrong
    stale row whose influence stays below the aa threshold."""
    kw = dict(aa=48, aac=48)
    clip = clip_format(fmt, w, h)
    src = synth.frame(clip, "noise", seed=5)
    n = NumpySangNom(width=w, height=h, bytes=clip.bytes, bits=clip.bits, planes=3, subw=1, subh=1, **kw)
    want = []
    for p in (0, 1):
        d = np.zeros_like(src[p])
        d[0::2] = src[p][0::2]     # order=1 keeps the even lines (offset 0, SangNom2.cpp:344-351)
        n._plane(d, 0, p)
        want.append(n.pool.copy())
    # (8-bit: U and V run as one sweep by default, which keeps the U -> V hand-off in registers -- sn_fused_u8_uv.hip; the
    # two-sweep form with its second pool is sn_policy.chroma_sweeps = 1, and is what shows that hand-off here)
    with SangNom2(clip, mode="fused", chroma_sweeps=1, **kw) as flt:
        flt.get_frame(src)
        assert flt.info().uv_sweeps == 0
        rows = flt.info().coupled_rows
        nr_c, bh = h // 4 - 1, (h + 1) // 2
        assert rows == min(nr_c + 2, bh - 1) + 1
        # Only the cells inside the dependency cone are handed on: (row q, column x) can reach a chroma output iff
        # x < w_c + 3 * (nr_c - q + 2) (+6 for what U needs from luma), and inside the chroma region (x < w_c) only
        # the rows below it matter (sn_fused_v3_common.h, Args::cone_*).  What lies outside is never written.
        w_c = w // 2
        for which, last, extra in ((0, rows - 1, 6), (1, min(nr_c + 1, bh - 1), 0)):
            got = flt.read_coupled_rows(which)[:, 1:last + 1]
            exp = want[which][:, 1:last + 1, :w]
            if clip.bytes == 4:  # float: bit patterns
                got, exp = got.view(np.uint32), np.ascontiguousarray(exp, dtype=np.float32).view(np.uint32)
            got, exp = got.astype(np.int64), exp.astype(np.int64)
            q = np.arange(1, last + 1)[:, None]
            x = np.arange(w)[None, :]
            cone = (x < w_c + 3 * (nr_c - q + 2) + extra) & ((x >= w_c) | (q > nr_c))
            assert cone.any()
            bad = np.argwhere((got != exp) & cone[None])
            assert len(bad) == 0, f"hand-off {which}: {len(bad)} samples differ, first (buffer,row-1,x) {bad[:4].tolist()}"
            if h >= 256:  # the trimming is real: far outside the cone nothing was stored (pools start zero-filled)
                far = (x >= w_c + 3 * (nr_c - q + 2) + extra + 16) & (q <= nr_c)
                assert far.any() and not got[:, far].any()


# Stage 2 of the pool path picks its kernel by sample type, pool stride and frames per launch (sn_pool_kernels.hip):
# strips of 60 lanes x 8 columns from 512 columns on (float: launches of up to eight frames), eight columns per thread
# below / otherwise.  Widths: 512 (two lanes in the second strip), 736 (a 720-wide clip's stride is the same: second
# strip partly idle), 1000 (stride 1024: three strips), 1504 (four strips), 480 (below the strips).
STAGE2_WIDTHS = (480, 512, 736, 1000, 1504)


@pytest.mark.parametrize("w", STAGE2_WIDTHS)
@pytest.mark.parametrize("fmt", ["Y8", "Y12", "Y16", "Y32"])
def test_pool_stage2_kernels_match_oracle(hip_lib, fmt, w):
    import torch
    clip = clip_format(fmt, w, 56)
    kw = dict(aa=48)
    history_free = w % 32 == 0
    N = 10
    frames = make_frames(clip, "noise", N - 1, seed0=11) + [synth.frame(clip, "checker", seed=1)]
    dev = torch.device("cuda:0")
    tdt = {1: torch.uint8, 2: torch.int16, 4: torch.float32}[clip.bytes]
    vdt = {1: np.uint8, 2: np.int16, 4: np.float32}[clip.bytes]
    ora = Oracle(oracle_cfg(clip, **kw))
    want = [ora.process(fr, parity=1) for fr in frames]
    with SangNom2(clip, max_batch=N, mode="pool", **kw) as flt:
        assert bool(flt.info().history_free) == history_free
        src = [torch.from_numpy(np.stack([fr[0] for fr in frames]).view(vdt)).pin_memory().to(dev)]
        dst = [torch.zeros((N,) + flt.plane_shape_out(0), dtype=tdt, device=dev)]
        # one launch of three frames and one of ten; a history-carrying clip (1000 wide) runs frame by frame whatever
        # the launch holds, and its frames must come in the oracle's order from a new instance
        for n in ((3, N) if history_free else (N,)):
            dst[0].zero_()
            torch.cuda.synchronize()
            flt.process_batch([src[0][:n]], [dst[0][:n]], parity=[1] * n)
            flt.synchronize()
            for f in range(n):
                got = to_host(dst[0][f]).view(clip.dtype)
                assert same(want[f][0], got), f"{fmt} {w} launch of {n}, frame {f}: " + describe_diff(want[f][0], got)


# History-carrying clips, several frames per launch: the passes run as one chain (run_chain, k_smooth_{u8,u16,f32}_chain).
CHAIN_CASES = [
    # fmt, w, h, kw, frames, sn_policy.scratch_budget_mb
    ("YUV420P8", 720, 96, dict(aac=48), 12, None),              # two strips, eight passes in flight, three planes a frame
    ("Y8", 1000, 56, dict(aa=20), 10, None),                     # three strips
    ("Y8", 40, 200, dict(order=0), 7, None),                     # one strip, sixteen passes in flight; the field follows the parity
    ("YUV420P8", 128, 64, dict(luma=False, aac=33), 20, None),   # chroma only: a multiple of 32 wide and still a chain
    ("YUV420P8", 128, 64, dict(luma=False, aac=33), 20, 1),      # ... in launches of a few passes
    ("YUV422P8", 208, 48, dict(aac=48, dh=True), 9, None),
    ("YUV420P8", 1456, 40, dict(aac=10), 6, None),               # four strips, four passes in flight
    ("YUV420P10", 720, 96, dict(aac=48), 12, None),              # 9..16-bit samples: k_smooth_u16_chain
    ("Y16", 1000, 56, dict(aa=20), 10, None),
    ("YUV420P16", 128, 64, dict(luma=False, aac=33), 20, 1),
    ("Y12", 40, 200, dict(order=0), 7, None),
    ("YUV420PS", 720, 96, dict(aac=48), 12, None),               # float samples: k_smooth_f32_chain
    ("Y32", 1000, 56, dict(aa=20), 10, None),
    ("YUV420PS", 128, 64, dict(luma=False, aac=33), 20, 1),
]


@pytest.mark.parametrize("fmt,w,h,kw,N,budget", CHAIN_CASES, ids=[f"{c[0]}-{c[1]}x{c[2]}-{i}" for i, c in enumerate(CHAIN_CASES)])
@pytest.mark.parametrize("mode", ["auto", "pool"])
def test_history_carrying_chain_matches_oracle(hip_lib, monkeypatch, fmt, w, h, kw, N, budget, mode):
    import torch
    if budget:
        from avisynth_sangnom2_amd import capi
        monkeypatch.setitem(capi.POLICY_DEFAULTS, "scratch_budget_mb", budget)
    clip = clip_format(fmt, w, h)
    frames = make_frames(clip, "noise", N - 1, seed0=5) + [synth.frame(clip, "checker", seed=2)]
    parity = [1] * (N // 2) + [0] + [1] * (N - N // 2 - 1)  # two chains with a single frame of the other field between them
    ora = Oracle(oracle_cfg(clip, **kw))
    dev = torch.device("cuda:0")
    with SangNom2(clip, max_batch=N, mode=mode, **kw) as flt:
        assert not flt.info().history_free
        view = {1: np.uint8, 2: np.int16, 4: np.float32}[clip.bytes]
        src = [torch.from_numpy(np.stack([fr[p] for fr in frames]).view(view)).pin_memory().to(dev) for p in range(clip.planes)]
        dst = [torch.zeros((N,) + flt.plane_shape_out(p), dtype=src[p].dtype, device=dev) for p in range(clip.planes)]
        torch.cuda.synchronize()
        for rnd in range(2):  # the second launch starts from the pool the first one left
            flt.process_batch(src, dst, parity=parity)
            flt.synchronize()
            for f in range(N):
                want = ora.process(frames[f], parity=parity[f])
                for p in range(clip.planes):
                    got = to_host(dst[p][f]).view(clip.dtype)
                    assert same(want[p], got), f"round {rnd} frame {f} plane {p}: " + describe_diff(want[p], got)
        assert 2 * (N - 1) <= flt.info().chained_frames <= 2 * N  # (the field only moves the lines when order = 0)
        # a frame on its own carries on from the chain's last pool, and the pool itself is the reference's
        want = ora.process(frames[0], parity=1)
        got = flt.get_frame(frames[0], parity=1)
        for p in range(clip.planes):
            assert same(want[p], got[p]), f"single frame after the chains, plane {p}: " + describe_diff(want[p], got[p])
        assert np.array_equal(ora.pool(), flt.read_pool(0))


# Chains over several workgroups per cost buffer (sn_policy.chain = 1, 2, 4, 8; k_smooth_{u8,u16,f32}_chain<true>): the hand-off
# between two workgroups goes through memory and round counters -- every count must give the oracle's frames.
GROUP_CASES = [
    # fmt, w, h, kw, frames
    ("YUV420P8", 720, 96, dict(aac=48), 12),     # two strips: 36 passes over sixteen slots, several cycles
    ("YUV420P8", 720, 480, dict(aac=48), 7),     # the bench's history-carrying clip: 49 rounds a pass
    ("Y8", 1000, 56, dict(aa=20), 23),           # three strips: two workgroups at most (pool_chain_groups), the rest of the counts fall back to that
    ("Y8", 40, 200, dict(order=0), 40),          # one strip: thirty-two slots
    ("YUV420P8", 1456, 40, dict(aac=10), 9),     # four strips
    ("YUV444P8", 3000, 24, dict(aac=48), 5),     # seven strips: two workgroups at most
    ("YUV420P10", 720, 96, dict(aac=48), 12),    # 9..16-bit samples (k_smooth_u16_chain<true>): the workgroups add slots
    ("Y16", 1000, 56, dict(aa=20), 23),
    ("Y12", 40, 200, dict(order=0), 40),
    ("YUV420PS", 720, 96, dict(aac=48), 12),     # float samples (k_smooth_f32_chain<true>)
    ("Y32", 1000, 56, dict(aa=20), 23),
    ("YUV444PS", 3000, 24, dict(aac=48), 5),
]


@pytest.mark.parametrize("fmt,w,h,kw,N", GROUP_CASES, ids=[f"{c[0]}-{c[1]}x{c[2]}" for c in GROUP_CASES])
@pytest.mark.parametrize("groups", [1, 2, 4, 8])
def test_chain_over_several_workgroups_per_buffer_matches_oracle(hip_lib, monkeypatch, fmt, w, h, kw, N, groups):
    import torch
    from avisynth_sangnom2_amd import capi
    monkeypatch.setitem(capi.POLICY_DEFAULTS, "chain", groups)
    clip = clip_format(fmt, w, h)
    frames = make_frames(clip, "noise", N - 1, seed0=11) + [synth.frame(clip, "checker", seed=4)]
    ora = Oracle(oracle_cfg(clip, **kw))
    dev = torch.device("cuda:0")
    with SangNom2(clip, max_batch=N, **kw) as flt:
        assert not flt.info().history_free
        assert flt.get_policy().chain == groups
        view = {1: np.uint8, 2: np.int16, 4: np.float32}[clip.bytes]
        src = [torch.from_numpy(np.stack([fr[p] for fr in frames]).view(view)).pin_memory().to(dev) for p in range(clip.planes)]
        dst = [torch.zeros((N,) + flt.plane_shape_out(p), dtype=src[p].dtype, device=dev) for p in range(clip.planes)]
        torch.cuda.synchronize()
        for rnd in range(2):  # the second launch starts from the pool the first one left
            flt.process_batch(src, dst)
            flt.synchronize()
            for f in range(N):
                want = ora.process(frames[f])
                for p in range(clip.planes):
                    got = to_host(dst[p][f]).view(clip.dtype)
                    assert same(want[p], got), f"{groups} workgroups, round {rnd} frame {f} plane {p}: " + describe_diff(want[p], got)
        assert flt.info().chained_frames == 2 * N
        assert np.array_equal(ora.pool(), flt.read_pool(0))


def test_host_ring_groups_of_a_history_carrying_clip_run_as_chains(hip_lib):
    """sn_submit_host / sn_collect_host with a ring deep enough for groups of several frames: a group's launch is a chain."""
    clip = clip_format("YUV420P8", 720, 64)
    kw = dict(aac=48)
    N = 21
    frames = make_frames(clip, "noise", N, seed0=9)
    ora = Oracle(oracle_cfg(clip, **kw))
    with SangNom2(clip, host_depth=12, **kw) as flt:
        slots = flt.host_slots()
        got, inflight = [], []
        for f in range(N):
            if len(inflight) == slots:
                got.append(flt.collect(inflight.pop(0)))
            inflight.append(flt.submit(frames[f]))
        while inflight:
            got.append(flt.collect(inflight.pop(0)))
        assert flt.info().chained_frames >= N // 2
    for f in range(N):
        want = ora.process(frames[f])
        for p in range(clip.planes):
            assert same(want[p], got[f][p]), f"frame {f} plane {p}: " + describe_diff(want[p], got[f][p])


def test_a_chain_launch_that_timed_out_is_redone_on_one_workgroup_per_buffer(hip_lib):
    """A workgroup of a chain over several workgroups per buffer that gives up waiting for the one before it raises the
    launch's fault word; the guarded launches queued behind every such launch -- stage 1 again, the chain on one workgroup per
    buffer -- then redo it before stage 3 runs.  The hook starts the next launch with the word up, so that its waves skip
    every wait (they take rows before they are written: the launch really goes wrong); the frames still equal the oracle's,
    the pool carried into the next launch as well, and sn_info.chain_redone counts the launch."""
    import torch
    from avisynth_sangnom2_amd.filter import SangNomError
    for fmt in ("Y8", "Y16", "Y32", "YUV420P8"):
        clip = clip_format(fmt, 1008 if fmt == "YUV420P8" else 1000, 56)  # (a chain needs planes a multiple of 8 wide: 504-wide chroma)
        N = 24
        frames = make_frames(clip, "noise", 3 * N, seed0=5)
        ora = Oracle(oracle_cfg(clip, aac=48))
        want = [ora.process(fr) for fr in frames]
        dev = torch.device("cuda:0")
        tdt = {1: torch.uint8, 2: torch.int16, 4: torch.float32}[clip.bytes]
        vdt = {1: np.uint8, 2: np.int16, 4: np.float32}[clip.bytes]
        with SangNom2(clip, max_batch=N, aac=48, chain=8) as flt:
            for launch in range(3):
                part = frames[launch * N:(launch + 1) * N]
                src = [torch.from_numpy(np.stack([fr[p] for fr in part]).view(vdt)).pin_memory().to(dev) for p in range(clip.planes)]
                dst = [torch.zeros((N,) + flt.plane_shape_out(p), dtype=tdt, device=dev) for p in range(clip.planes)]
                torch.cuda.synchronize()
                if launch == 1:
                    flt.raise_chain_fault()
                flt.process_batch(src, dst)
                flt.synchronize()
                info = flt.info()
                assert info.chained_frames == (launch + 1) * N
                assert info.chain_redone == (1 if launch >= 1 else 0), (fmt, launch, info.chain_redone)
                for f in range(N):
                    for p in range(clip.planes):
                        got = to_host(dst[p][f]).view(clip.dtype)
                        assert same(want[launch * N + f][p], got), f"{fmt} launch {launch} frame {f} plane {p}: " + describe_diff(want[launch * N + f][p], got)
    with SangNom2(clip_format("Y8", 1000, 56)) as flt:  # no chain yet: nothing to raise
        with pytest.raises(SangNomError, match="has not run a chain"):
            flt.raise_chain_fault()


def capi_code(name):
    from avisynth_sangnom2_amd import capi
    return getattr(capi, name)


def test_chain_can_be_switched_off(hip_lib, monkeypatch):
    import torch
    from avisynth_sangnom2_amd import capi
    monkeypatch.setitem(capi.POLICY_DEFAULTS, "chain", -1)
    clip = clip_format("Y8", 1000, 56)
    frames = make_frames(clip, "noise", 4, seed0=5)
    ora = Oracle(oracle_cfg(clip))
    dev = torch.device("cuda:0")
    with SangNom2(clip, max_batch=4) as flt:
        src = [torch.from_numpy(np.stack([fr[0] for fr in frames])).pin_memory().to(dev)]
        dst = [torch.zeros((4,) + flt.plane_shape_out(0), dtype=torch.uint8, device=dev)]
        torch.cuda.synchronize()
        flt.process_batch(src, dst)
        flt.synchronize()
        assert flt.info().chained_frames == 0
        for f in range(4):
            assert same(ora.process(frames[f])[0], to_host(dst[0][f]))


@pytest.mark.parametrize("fmt,mode", [("YUV420P8", "fused"), ("YUV420P16", "fused"), ("YUV420P8", "pool"), ("Y16", "pool")])
def test_batch_larger_than_the_scratch_budget_runs_in_chunks(hip_lib, monkeypatch, fmt, mode):
    """Scratch is bounded (sn_policy.scratch_budget_mb): a bigger batch is run in chunks on the same slots."""
    import torch
    from avisynth_sangnom2_amd import capi
    monkeypatch.setitem(capi.POLICY_DEFAULTS, "scratch_budget_mb", 1)
    clip = clip_format(fmt, 256, 64)
    kw = dict(aa=48, aac=48)
    N = 31
    frames = make_frames(clip, "noise", N, seed0=3)
    dev = torch.device("cuda:0")
    tdt = {np.uint8: torch.uint8, np.uint16: torch.int16}[clip.dtype]
    with SangNom2(clip, max_batch=N, mode=mode, **kw) as flt:
        src = [torch.from_numpy(np.stack([fr[p] for fr in frames]).view(np.int16 if clip.bytes == 2 else np.uint8)).pin_memory().to(dev)
               for p in range(clip.planes)]
        dst = [torch.zeros((N,) + flt.plane_shape_out(p), dtype=tdt, device=dev) for p in range(clip.planes)]
        torch.cuda.synchronize()
        flt.process_batch(src, dst)
        flt.synchronize()
        want_fused = N if mode == "fused" else 0
        assert flt.info().fused_frames == want_fused
    for f in range(N):
        want = Oracle(oracle_cfg(clip, **kw)).process(frames[f])
        for p in range(clip.planes):
            got = to_host(dst[p][f]).view(clip.dtype)
            assert same(want[p], got), f"frame {f} plane {p}: " + describe_diff(want[p], got)


@pytest.mark.parametrize("fmt,w,h,kw,depth", [
    ("Y8", 1920, 64, {}, 4),
    ("YUV420P8", 256, 64, dict(aac=48), 3),            # fused 4:2:0: one pair of hand-off pools per slot
    ("YUV420P16", 256, 64, dict(aac=48, order=0), 5),
    ("YUV444PS", 128, 48, dict(aac=48, dh=True), 2),
    ("Y8", 100, 40, {}, 4),                            # width % 32 != 0: pool state carries, frames stay in order
    ("YUV420P8", 72, 32, dict(aac=48, order=0), 3),    # history-carrying, parity alternates
])
def test_host_ring_pipelines_frames_like_get_frame(hip_lib, fmt, w, h, kw, depth):
    """sn_submit_host / sn_collect_host with several frames in flight == one oracle instance fed in order."""
    clip = clip_format(fmt, w, h)
    N = 11
    frames = make_frames(clip, "noise", N, seed0=41)
    ora = Oracle(oracle_cfg(clip, **kw))
    want = [ora.process(frames[f], parity=f & 1) for f in range(N)]
    with SangNom2(clip, host_depth=depth, **kw) as flt:
        slots = flt.host_slots()
        assert 1 <= slots <= depth
        inflight, got = [], []
        for f in range(N):
            if len(inflight) == slots:
                got.append(flt.collect(inflight.pop(0)))
            inflight.append(flt.submit(frames[f], parity=f & 1))
        with pytest.raises(SangNomError, match="in flight") if len(inflight) == slots else _nullcontext():
            flt.submit(frames[0])
        while inflight:
            got.append(flt.collect(inflight.pop(0)))
        with pytest.raises(SangNomError, match="holds no frame"):
            flt.collect(0)
    for f in range(N):
        for p in range(len(want[f])):
            assert same(want[f][p], got[f][p]), f"frame {f} plane {p}: " + describe_diff(want[f][p], got[f][p])


ISOLATED = [
    ("YUV420P8", 3840, 64, dict(aa=48, aac=48)),        # chroma 1920 wide: 2-wave planes, two frames per workgroup
    ("YUV420P8", 128, 64, dict(aac=30, order=0)),
    ("YUV420P16", 256, 48, dict(aac=48)),
    ("YUV422P8", 576, 28, dict(aac=48, order=2)),
    ("YUV420PS", 128, 32, dict(aac=48)),
    ("YUV420P8", 200, 40, dict(aac=48)),                # luma 200, chroma 100 wide: every plane history-carrying
    ("YUV420P8", 320, 40, dict(aac=48, dh=True)),       # luma fused, chroma (160 wide) too
    ("YUV420P8", 192, 32, dict(aac=48, luma=False)),    # luma copied
    ("YUV420P8", 192, 32, dict(chroma=False)),
]


@pytest.mark.parametrize("fmt,w,h,kw", ISOLATED, ids=[f"{c[0]}-{c[1]}x{c[2]}-{i}" for i, c in enumerate(ISOLATED)])
def test_isolated_planes_equal_the_filter_applied_to_each_plane_as_a_y_clip(hip_lib, fmt, w, h, kw):
    """sn_config.isolated_planes (extension): plane p of the result == what the reference gives for that plane
    fed in as a Y clip of its own (ExtractY/U/V -> SangNom2 -> CombinePlanes), frame after frame."""
    clip = clip_format(fmt, w, h)
    N = 3
    frames = make_frames(clip, "noise", N, seed0=61)
    oracles = []
    for p in range(3):
        yclip = ClipFormat(width=w >> (clip.subw if p else 0), height=h >> (clip.subh if p else 0), bytes=clip.bytes, bits=clip.bits)
        enabled = kw.get("luma", True) if p == 0 else kw.get("chroma", True)
        oracles.append(Oracle(oracle_cfg(yclip, order=kw.get("order", 1), aa=kw.get("aa", 48) if p == 0 else kw.get("aac", 0),
                                         dh=kw.get("dh", False), luma=enabled)))
    with SangNom2(clip, isolated_planes=True, **kw) as flt:
        for f in range(N):
            got = flt.get_frame(frames[f], parity=f & 1)
            for p in range(3):
                want = oracles[p].process([frames[f][p]], parity=f & 1)[0]
                assert same(want, got[p]), f"frame {f} plane {p}: " + describe_diff(want, got[p])
        info = flt.info()
        all_mod32 = all((w >> (clip.subw if p else 0)) % 32 == 0 for p in range(3)
                        if kw.get("dh") or (kw.get("luma", True) if p == 0 else kw.get("chroma", True)))
        assert info.history_free == int(all_mod32)
        assert info.fused_eligible == int(all_mod32)


FRESH = [
    ("YUV420P8", 720, 480, dict(aa=48, aac=48)),      # SD: luma 720 (pool stride 736), chroma 360 (384): padded sweeps
    ("Y8", 2160, 64, dict(order=2)),                  # a turned 2160p plane: stride 2176
    ("Y16", 1080, 48, {}),
    ("Y8", 100, 40, dict(aa=20)),                     # 100 % 8 != 0: pool path over a pool zeroed per frame
    ("Y32", 720, 32, {}),                             # float, padded sweep
    ("Y32", 100, 24, {}),                             # float, pool path
    ("YUV420P8", 128, 64, dict(aac=48, order=0)),     # widths that need no padding: plain sweeps
    ("YUV444P16", 200, 32, dict(aac=48, dh=True)),
    ("Y8", 7672, 16, {}),                             # stride 7680: the widest padded sweep
]


@pytest.mark.parametrize("fmt,w,h,kw", FRESH, ids=[f"{c[0]}-{c[1]}x{c[2]}" for c in FRESH])
def test_fresh_pool_equals_a_new_instance_per_frame_and_plane(hip_lib, fmt, w, h, kw):
    """sn_config.fresh_pool (extension): plane p of frame f == the reference's FIRST output frame for that plane
    fed in as a Y clip; frames run as a batch (no dependence on earlier frames, whatever the width)."""
    import torch
    clip = clip_format(fmt, w, h)
    N = 4
    frames = make_frames(clip, "noise", N, seed0=81)
    dev = torch.device("cuda:0")
    tdt = {np.uint8: torch.uint8, np.uint16: torch.int16, np.float32: torch.float32}[clip.dtype]
    vt = {1: np.uint8, 2: np.int16, 4: np.float32}[clip.bytes]
    with SangNom2(clip, fresh_pool=True, max_batch=N, **kw) as flt:
        assert flt.info().history_free == 1
        src = [torch.from_numpy(np.stack([fr[p] for fr in frames]).view(vt)).pin_memory().to(dev) for p in range(clip.planes)]
        dst = [torch.zeros((N,) + flt.plane_shape_out(p), dtype=tdt, device=dev) for p in range(clip.planes)]
        torch.cuda.synchronize()
        parity = [f & 1 for f in range(N)]
        flt.process_batch(src, dst, parity)
        flt.synchronize()
        host_way = flt.get_frame(frames[1], parity=1)
    for f in range(N):
        for p in range(clip.planes):
            yclip = ClipFormat(width=w >> (clip.subw if p else 0), height=h >> (clip.subh if p else 0), bytes=clip.bytes, bits=clip.bits)
            ora = Oracle(oracle_cfg(yclip, order=kw.get("order", 1), aa=kw.get("aa", 48) if p == 0 else kw.get("aac", 0), dh=kw.get("dh", False)))
            want = ora.process([frames[f][p]], parity=parity[f])[0]
            got = to_host(dst[p][f]).view(clip.dtype)
            assert same(want, got), f"frame {f} plane {p}: " + describe_diff(want, got)
            if f == 1:
                assert same(want, host_way[p]), f"host path, plane {p}"


@pytest.mark.parametrize("fmt,w,h", [("Y8", 200, 136), ("Y8", 64, 64), ("Y16", 70, 33), ("Y32", 129, 66), ("Y8", 3840, 2160)])
def test_turn_device_is_avisynths_turnright_turnleft(hip_lib, fmt, w, h):
    import torch
    clip = clip_format(fmt, w, h)
    dev = torch.device("cuda:0")
    vt = {1: np.uint8, 2: np.int16, 4: np.float32}[clip.bytes]
    a = np.stack([synth.frame(clip, "noise", seed=s)[0] for s in range(2)])
    with SangNom2(ClipFormat(width=64, height=32, bytes=clip.bytes, bits=clip.bits)) as flt:
        src = torch.from_numpy(a.view(vt)).pin_memory().to(dev)
        for direction, k in ((+1, -1), (-1, 1)):  # TurnRight = clockwise = rot90(k=-1)
            dst = torch.zeros((2, w, h), dtype=src.dtype, device=dev)
            torch.cuda.synchronize()
            flt.turn(src, dst, direction)
            flt.synchronize()
            assert np.array_equal(to_host(dst).view(clip.dtype), np.rot90(a, k=k, axes=(1, 2)))


AA_CASES = [("Y8", 128, 64, dict(aa=48)), ("YUV420P8", 128, 64, dict(aa=48, aac=48)), ("Y16", 96, 64, dict(aa=30, order=2)),
            ("Y32", 64, 32, {}), ("Y8", 96, 80, dict(aa=48)), ("YUV422P8", 128, 64, dict(aac=48)),
            ("Y8", 96, 80, dict(aa=48, fresh_pool=True))]


@pytest.mark.parametrize("fmt,w,h,kw", AA_CASES, ids=[f"{c[0]}-{c[1]}x{c[2]}-{i}" for i, c in enumerate(AA_CASES)])
def test_anti_aliasing_idiom_on_the_device(hip_lib, fmt, w, h, kw):
    """TurnLeft().SangNom2().TurnRight().SangNom2() with the frames staying on the device == the same script with
    two oracle instances and numpy turns, frame after frame (96x80: the turned clip is 80 wide, so the first
    instance carries pool state from frame to frame)."""
    import torch
    clip = clip_format(fmt, w, h)
    turned = ClipFormat(width=h, height=w, bytes=clip.bytes, bits=clip.bits, planes=clip.planes, subw=clip.subh, subh=clip.subw)
    N = 3
    frames = make_frames(clip, "noise", N, seed0=91)
    fresh = kw.get("fresh_pool", False)
    okw = {k: v for k, v in kw.items() if k != "fresh_pool"}

    def oracle_pass(c, planes, state):
        if not fresh:
            return state.setdefault(id(c), Oracle(oracle_cfg(c, **okw))).process(planes)
        out = []
        for p, pl in enumerate(planes):  # every plane of every frame on a new instance
            y = ClipFormat(width=pl.shape[1], height=pl.shape[0], bytes=c.bytes, bits=c.bits)
            out.append(Oracle(oracle_cfg(y, order=okw.get("order", 1), aa=okw.get("aa", 48) if p == 0 else okw.get("aac", 0))).process([pl])[0])
        return out

    state = {}
    want = []
    for fr in frames:
        a = oracle_pass(turned, [np.ascontiguousarray(np.rot90(pl, k=1)) for pl in fr], state)
        want.append(oracle_pass(clip, [np.ascontiguousarray(np.rot90(pl, k=-1)) for pl in a], state))
    dev = torch.device("cuda:0")
    vt = {1: np.uint8, 2: np.int16, 4: np.float32}[clip.bytes]
    with SangNomAA(clip, max_batch=N, **kw) as aa:
        src = [torch.from_numpy(np.stack([fr[p] for fr in frames]).view(vt)).pin_memory().to(dev) for p in range(clip.planes)]
        dst = [torch.zeros_like(s) for s in src]
        torch.cuda.synchronize()
        aa.process_batch(src, dst)
        aa.synchronize()
    for f in range(N):
        for p in range(clip.planes):
            got = to_host(dst[p][f]).view(clip.dtype)
            assert same(want[f][p], got), f"frame {f} plane {p}: " + describe_diff(want[f][p], got)


import glob as _glob
import json as _json
import os as _os

_GOLDEN = sorted(_glob.glob(_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "golden", "*.npz")))


@pytest.mark.parametrize("path", _GOLDEN, ids=_os.path.basename)
@pytest.mark.parametrize("mode", ["auto", "pool"])
def test_committed_golden_vectors(hip_lib, path, mode):
    """The committed input / output vectors (tests/golden/*.npz, written by make_golden.py from the oracle) through
    the C ABI: no oracle involved on the box, frames in order on one instance."""
    z = np.load(path)
    meta = _json.loads(bytes(z["meta"]).decode())
    clip = clip_format(meta["fmt"], meta["width"], meta["height"])
    with SangNom2(clip, mode=mode, **meta["kw"]) as flt:
        for f in range(meta["nframes"]):
            src = [z[f"in_f{f}_p{p}"] for p in range(clip.planes)]
            got = flt.get_frame(src, parity=(f + 1) & 1)
            for p in range(clip.planes):
                want = z[f"out_f{f}_p{p}"]
                assert same(want, got[p]), f"{_os.path.basename(path)} frame {f} plane {p}: " + describe_diff(want, got[p])


with open(_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "golden", "full_size_sha256.json")) as _f:
    _FULL_HASHES = _json.load(_f)


@pytest.mark.parametrize("case", _FULL_HASHES, ids=[c["name"] for c in _FULL_HASHES])
def test_full_size_outputs_have_the_committed_sha256(hip_lib, case):
    """BASELINE.json's configurations at full size: every output plane hashes to the committed value
    (tests/golden/full_size_sha256.json, from the oracle via make_golden.py; inputs from the portable generator)."""
    import hashlib
    clip = clip_format(case["fmt"], case["width"], case["height"])
    with SangNom2(clip, **case["kw"]) as flt:
        for f, want in enumerate(case["sha256"]):
            got = flt.get_frame(synth.frame(clip, case["pattern"], seed=case["seed0"] + f), parity=1)
            for p, digest in enumerate(want):
                assert hashlib.sha256(np.ascontiguousarray(got[p]).tobytes()).hexdigest() == digest, f"frame {f} plane {p}"


@pytest.mark.small_launch_policy
@pytest.mark.parametrize("case", _FULL_HASHES, ids=[c["name"] for c in _FULL_HASHES])
def test_default_single_frame_path_at_full_size_has_the_committed_sha256(hip_lib, case, record_property):
    """What a plugin user gets by default -- ONE frame per call, mode "auto", the library's own small-launch policy:
    row bands for planes on their own, luma bands + pool kernels for the chroma of 4:2:0 -- at BASELINE's
    sizes against the committed hashes (src/SangNom2.cpp:332-397 is what both sides compute).  The bands must actually
    have run; frames redone by the fallback are reported, not forbidden (the result is exact either way)."""
    import hashlib
    clip = clip_format(case["fmt"], case["width"], case["height"])
    with SangNom2(clip, mode="auto", **case["kw"]) as flt:
        for f, want in enumerate(case["sha256"]):
            got = flt.get_frame(synth.frame(clip, case["pattern"], seed=case["seed0"] + f), parity=1)
            for p, digest in enumerate(want):
                assert hashlib.sha256(np.ascontiguousarray(got[p]).tobytes()).hexdigest() == digest, f"frame {f} plane {p}"
        info = flt.info()
        record_property("banded_frames", int(info.banded_frames))
        record_property("band_fallbacks", int(info.band_fallbacks))
        assert info.banded_frames == len(case["sha256"]), (info.banded_frames, info.band_fallbacks)
        assert info.band_fallbacks <= info.banded_frames


def test_legacy_sangnom_wrapper_and_single_frame_device_entry(hip_lib):
    """SangNom(clip, order, aa): order 0/1/2 = bottom/top/double-rate is remapped to SangNom2's 2/1/0
    (src/SangNom2.cpp:441,463), aac = 0; and sn_process_device (one device-resident frame) on the context's stream."""
    import torch
    clip = clip_format("YUV420P8", 128, 64)
    frames = make_frames(clip, "noise", 2, seed0=5)
    for legacy_order, order in ((0, 2), (1, 1), (2, 0)):
        ora = Oracle(oracle_cfg(clip, order=order, aa=30, aac=0))
        with SangNom(clip, order=legacy_order, aa=30) as flt:
            assert flt.stream_handle() not in (None, 0)
            for f, fr in enumerate(frames):
                want = ora.process(fr, parity=f)
                got = flt.get_frame(fr, parity=f)
                dev = torch.device("cuda:0")
                src = [torch.from_numpy(pl).pin_memory().to(dev) for pl in fr]
                dst = [torch.zeros_like(s) for s in src]
                torch.cuda.synchronize()
                for p in range(3):
                    assert same(want[p], got[p]), f"legacy order {legacy_order} frame {f} plane {p}"
    with pytest.raises(SangNomError, match="order must be between 0..2"):
        SangNom(clip, order=3)
    # the single-frame device entry point: same frame, same result (history-free clip)
    with SangNom2(clip, aa=30) as flt:
        want = Oracle(oracle_cfg(clip, aa=30)).process(frames[0])
        src = [torch.from_numpy(pl).pin_memory().to(dev) for pl in frames[0]]
        dst = [torch.zeros_like(s) for s in src]
        torch.cuda.synchronize()
        flt.get_frame_device(src, dst)
        flt.synchronize()
        for p in range(3):
            assert same(want[p], to_host(dst[p])), f"device frame plane {p}"


@pytest.mark.small_launch_policy
@pytest.mark.parametrize("fmt,w,h,kw", [("Y8", 3840, 2160, {}), ("YUV420P8", 1920, 1080, dict(aac=48)), ("Y16", 1920, 1080, {}),
                                        ("YUV420P8", 720, 480, dict(aac=48, fresh_pool=True))])
def test_small_launches_take_the_pool_path_or_row_bands_and_large_ones_the_sweeps(hip_lib, fmt, w, h, kw):
    """SN_MODE_AUTO: one frame goes through the sweep cut into row bands (planes on their own) or through the pool
    kernels (coupled 4:2:0, isolated planes), a launch of a whole round of workgroups through the whole-plane sweeps;
    same bytes either way."""
    import torch
    clip = clip_format(fmt, w, h)
    dev = torch.device("cuda:0")
    N = 1024 if w <= 720 else 256
    tdt = {1: torch.uint8, 2: torch.int16}[clip.bytes]
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    src = []
    for p in range(clip.planes):
        hp, wp = h >> (clip.subh if p else 0), w >> (clip.subw if p else 0)
        hi = 256 if clip.bytes == 1 else 1 << clip.bits
        src.append(torch.randint(0, hi, (N, hp, wp), device=dev, generator=g, dtype=torch.int32).to(tdt))
    with SangNom2(clip, max_batch=N, **kw) as flt:
        one = [torch.zeros_like(s[:1]) for s in src]
        torch.cuda.synchronize()
        flt.process_batch([s[:1] for s in src], one)
        flt.synchronize()
        # a plane on its own: row bands; coupled 4:2:0: the luma plane in bands, chroma by the pool kernels (banded_frames
        # is a subset of fused_frames, sangnom_hip.h); isolated planes of a width the sweeps only serve padded: the pool path
        banded = 0 if kw.get("fresh_pool") else 1
        fused1 = banded
        info = flt.info()
        assert (info.fused_frames, info.banded_frames) == (fused1, banded)
        out = [torch.zeros_like(s) for s in src]
        flt.process_batch(src, out)
        flt.synchronize()
        info = flt.info()
        assert (info.fused_frames, info.banded_frames) == (N + fused1, banded), "a full launch should have taken the whole-plane sweeps"
        for p in range(clip.planes):
            assert torch.equal(one[p][0], out[p][0]), f"plane {p}: the two paths disagree"


# ---- one frame stream over G replicas (MT_MULTI_INSTANCE, SangNom2.h:63-66) ---------------
@pytest.mark.parametrize("world", [2, 4])
@pytest.mark.parametrize("mode", ["round_robin", "block"])
@pytest.mark.parametrize("fmt,w,h,kw", [("Y8", 512, 96, {}), ("YUV420P8", 256, 64, dict(aac=48))], ids=["Y8", "YUV420P8"])
def test_a_stream_sharded_over_logical_ranks_equals_one_context(hip_lib, world, mode, fmt, w, h, kw):
    """G logical ranks mapped to one device, each with its own context (as bench.py --gpus G gives every GPU its
    own): rank r sweeps the frames shard.frames_for_rank hands it; reassembled in stream order the outputs are
    byte-for-byte what ONE context produces for the whole stream, and one frame per rank is checked against the oracle."""
    import torch
    from avisynth_sangnom2_amd import shard
    clip = clip_format(fmt, w, h)
    n_frames = 67  # not a multiple of the world size
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(5150)
    shapes = [(h >> (clip.subh if p else 0), w >> (clip.subw if p else 0)) for p in range(clip.planes)]
    src = [torch.randint(0, 256, (n_frames,) + sh, device=dev, generator=g, dtype=torch.uint8) for sh in shapes]
    parity = [(f * 7) & 1 for f in range(n_frames)]
    kw = dict(kw, order=0)  # order 0: the field follows the frame's parity, so a shard must carry its own parities
    torch.cuda.synchronize()
    with SangNom2(clip, max_batch=n_frames, mode="fused", **kw) as one:
        want = [torch.zeros_like(s) for s in src]
        one.process_batch(src, want, parity)
        one.synchronize()
    got = [torch.zeros_like(s) for s in src]
    ranks = [SangNom2(clip, max_batch=n_frames, mode="fused", **kw) for _ in range(world)]
    try:
        covered = []
        for r, flt in enumerate(ranks):
            mine = shard.frames_for_rank(n_frames, r, world, mode)
            covered += mine
            idx = torch.tensor(mine, device=dev, dtype=torch.long)
            s_r = [s.index_select(0, idx) for s in src]       # the rank's private batch
            d_r = [torch.zeros_like(x) for x in s_r]
            torch.cuda.synchronize()
            flt.process_batch(s_r, d_r, [parity[f] for f in mine])
            flt.synchronize()
            assert flt.info().fused_frames == len(mine)
            for p in range(clip.planes):
                got[p].index_copy_(0, idx, d_r[p])
        assert sorted(covered) == list(range(n_frames))
        torch.cuda.synchronize()
        for p in range(clip.planes):
            assert torch.equal(got[p], want[p]), f"plane {p}: sharded stream differs from the single context"
        for r in range(world):
            f = shard.frames_for_rank(n_frames, r, world, mode)[-1]
            ref = Oracle(oracle_cfg(clip, **kw)).process([to_host(s[f]) for s in src], parity=parity[f])
            for p in range(clip.planes):
                assert same(ref[p], to_host(got[p][f]))
    finally:
        for flt in ranks:
            flt.close()


# ---- round-2 additions: layout fallback, field orders at full height, non-finite floats ---------------------
@pytest.mark.parametrize("fmt", ["Y8", "Y16", "Y32", "YUV420P8"])
def test_misaligned_device_planes_fall_back_to_the_pool_path(hip_lib, fmt):
    """The fused sweeps need 8-byte aligned plane pointers and pitches (sn_fused_select.hip, fused_layout_ok); anything
    else is served by the pool path -- same result, `fused_frames` stays 0."""
    import torch
    w, h, N = 256, 64, 3
    clip = clip_format(fmt, w, h)
    kw = dict(aac=48) if clip.planes == 3 else {}
    dev = torch.device("cuda:0")
    frames = make_frames(clip, "noise", N, seed0=31)
    tdt = {1: torch.uint8, 2: torch.int16, 4: torch.float32}[clip.bytes]
    B = clip.bytes
    with SangNom2(clip, max_batch=N, **kw) as flt:
        assert flt.info().fused_eligible == 1
        src, dst = [], []
        for p in range(flt.nplanes):
            hi, wi = flt.plane_shape_in(p)
            ho, wo = flt.plane_shape_out(p)
            # odd pitch (in samples) and a base shifted by one sample: rows start on no 8-byte boundary
            big = torch.zeros((N, hi, wi + 5), dtype=tdt, device=dev)
            s = big[:, :, 1:1 + wi]
            s.copy_(torch.from_numpy(np.stack([fr[p] for fr in frames]).view({1: np.uint8, 2: np.int16, 4: np.float32}[B])))
            dbig = torch.zeros((N, ho, wo + 3), dtype=tdt, device=dev)
            src.append(s)
            dst.append(dbig[:, :, 1:1 + wo])
            assert (s.data_ptr() % 8) or ((s.stride(1) * B) % 8)
        torch.cuda.synchronize()
        flt.process_batch(src, dst)
        flt.synchronize()
        assert flt.info().fused_frames == 0 and flt.info().frames == N
        for f in range(N):
            want = Oracle(oracle_cfg(clip, **kw)).process(frames[f])
            for p in range(flt.nplanes):
                got = to_host(dst[p][f]).view(clip.dtype)
                assert same(want[p], got), f"{fmt} frame {f} plane {p}"


@pytest.mark.parametrize("fmt,w,h,kw", [("Y8", 3840, 2160, dict(order=0)), ("Y8", 3840, 2160, dict(order=2, aa=20)),
                                        ("Y8", 3840, 1080, dict(order=1, dh=True)), ("Y8", 3840, 1080, dict(order=0, dh=True)),
                                        ("Y16", 3840, 2160, dict(order=2)), ("Y32", 3840, 1080, dict(order=0, dh=True))],
                         ids=["Y8-order0", "Y8-order2", "Y8-dh", "Y8-dh-order0", "Y16-order2", "Y32-dh-order0"])
def test_field_orders_and_double_height_at_2160_rows_through_the_sweeps(hip_lib, fmt, w, h, kw):
    """order 0 / 2 and dh at BASELINE's frame height, fused sweeps: every frame of a small batch (alternating parity for
    order 0) against the oracle."""
    import torch
    clip = clip_format(fmt, w, h)
    N = 2
    frames = make_frames(clip, "noise", N, seed0=53)
    frames[1] = synth.frame(clip, "checker", seed=3)
    parity = [0, 1]
    dev = torch.device("cuda:0")
    npdt = {1: np.uint8, 2: np.int16, 4: np.float32}[clip.bytes]
    with SangNom2(clip, max_batch=N, mode="fused", **kw) as flt:
        src = [torch.from_numpy(np.stack([fr[0] for fr in frames]).view(npdt)).pin_memory().to(dev)]
        dst = [torch.zeros((N,) + flt.plane_shape_out(0), dtype=src[0].dtype, device=dev)]
        torch.cuda.synchronize()
        flt.process_batch(src, dst, parity)
        flt.synchronize()
        assert flt.info().fused_frames == N
        for f in range(N):
            want = Oracle(oracle_cfg(clip, **kw)).process(frames[f], parity=parity[f])
            assert same(want[0], to_host(dst[0][f]).view(clip.dtype)), f"{fmt} {kw} frame {f}"


@pytest.mark.parametrize("mode", ["fused", "pool"])
def test_non_finite_float_samples_follow_the_reference_ladder(hip_lib, mode):
    """Float clips may carry infinities and NaNs.  Where a NaN reaches the minimum of the nine buffers the reference takes
    NO arm of its ladder (every `==` and the `>` are false, src/SangNom2.cpp:204-249) and leaves the pixel of the new
    frame unwritten -- undefined content, nothing to match.  Everywhere else its result is defined by the `std::min`
    chain and the ladder, which the oracle restates operation for operation, and the HIP path must agree: bit patterns
    for numbers, NaN for NaN (x86 and gfx950 give the default NaN different signs)."""
    clip = clip_format("Y32", 256, 64)
    rng = np.random.default_rng(4)
    src = synth.frame(clip, "noise", seed=9)
    plane = src[0]
    ys, xs = rng.integers(0, 64, 40), rng.integers(0, 256, 40)
    vals = np.array([np.inf, -np.inf, np.nan, 3.0e38, -3.0e38], dtype=np.float32)
    plane[ys, xs] = vals[rng.integers(0, len(vals), 40)]
    # pixels the reference writes: the same whatever the new frame held before
    a = Oracle(oracle_cfg(clip)).process(src, dst=[np.zeros((64, 256), np.float32)])[0]
    b = Oracle(oracle_cfg(clip)).process(src, dst=[np.full((64, 256), 7.0, np.float32)])[0]
    written = (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))
    assert written.mean() > 0.7 and not written.all()
    with SangNom2(clip, mode=mode) as flt:
        got = flt.get_frame(src)[0]
    ok = (a.view(np.uint32) == got.view(np.uint32)) | (np.isnan(a) & np.isnan(got))
    bad = np.argwhere(written & ~ok)
    assert len(bad) == 0, f"{len(bad)} defined samples differ, first {bad[:4].tolist()}: {a[written & ~ok][:4]} vs {got[written & ~ok][:4]}"


@pytest.mark.parametrize("fmt,w,h,kw", AA_CASES[:6], ids=[f"{c[0]}-{c[1]}x{c[2]}-{i}" for i, c in enumerate(AA_CASES[:6])])
def test_anti_aliasing_entry_point_of_the_c_abi(hip_lib, fmt, w, h, kw):
    """sn_aa_process_host (what the plugin function SangNomAA binds): host frames through
    TurnLeft -> SangNom2 -> TurnRight -> SangNom2 in one call == the script with two oracle instances and numpy turns."""
    clip = clip_format(fmt, w, h)
    turned = ClipFormat(width=h, height=w, bytes=clip.bytes, bits=clip.bits, planes=clip.planes, subw=clip.subh, subh=clip.subw)
    first, second = Oracle(oracle_cfg(turned, **kw)), Oracle(oracle_cfg(clip, **kw))
    with SangNomAAHost(clip, **kw) as aa:
        for f, fr in enumerate(make_frames(clip, "noise", 3, seed0=17)):
            a = first.process([np.ascontiguousarray(np.rot90(pl, k=1)) for pl in fr])
            want = second.process([np.ascontiguousarray(np.rot90(pl, k=-1)) for pl in a])
            got = aa.get_frame(fr)
            for p in range(clip.planes):
                assert same(want[p], got[p]), f"frame {f} plane {p}: " + describe_diff(want[p], got[p])
    with pytest.raises(SangNomError, match="height must be even"):  # the TURNED clip fails the reference's check
        SangNomAAHost(clip_format("Y8", 63, 64))


# ---- row bands: the small-launch path of SN_MODE_AUTO (sn_fused_v3_common.h kBand, sn_band.hip) -------------------

BAND_CASES = [
    # (pixel_type, width, height, kwargs, bands, warm_rows)
    ("Y8", 256, 400, {}, 0, 0),                       # automatic cut
    ("Y8", 1056, 300, dict(order=2), 4, 0),
    ("Y8", 3840, 360, dict(aa=20), 7, 24),            # bands of unequal size
    ("Y8", 64, 200, dict(order=0), 12, 0),            # two frames per workgroup (one wave per plane)
    ("Y8", 960, 270, dict(dh=True), 5, 0),
    ("YUV444P8", 320, 240, dict(aac=30), 6, 0),
    ("YUV420P8", 256, 256, dict(chroma=False), 6, 0),  # chroma copied
    # 9..16-bit and float sweeps
    ("Y16", 256, 400, {}, 0, 0),
    ("Y10", 1056, 300, dict(order=2, aa=20), 4, 0),
    ("Y16", 3840, 360, {}, 7, 0),
    ("Y16", 64, 200, dict(dh=True), 6, 0),
    ("YUV444P16", 320, 240, dict(aac=30), 6, 0),
    ("Y32", 256, 400, {}, 0, 0),
    ("Y32", 1056, 300, dict(order=2, aa=20), 4, 0),
    ("Y32", 3840, 360, {}, 7, 0),
    ("YUV444PS", 320, 240, dict(aac=30, dh=True), 6, 0),
    # subsampled chroma: the luma plane in bands (its smoothed rows go where the pool path's luma pass leaves them), the
    # chroma planes through the pool kernels
    ("YUV420P8", 256, 400, dict(aac=48), 0, 0),
    ("YUV420P8", 1056, 288, dict(aac=20, order=2), 5, 0),
    ("YUV420P8", 3840, 400, dict(aac=48), 9, 0),
    ("YUV420P8", 7680, 264, dict(aac=48), 4, 0),
    ("YUV422P8", 576, 300, dict(aac=48), 6, 0),
    ("YUV420P8", 640, 200, dict(aac=48, dh=True), 7, 0),
    ("YUV420P8", 320, 240, dict(aac=48), 10, 1),       # a run-up of one row: luma redone by the pool path
    ("YUV420P16", 512, 320, dict(aac=48), 6, 0),
    ("YUV420PS", 256, 256, dict(aac=48, order=0), 5, 0),
]


# (round 2 kept this test last in the file because of a GPU page fault in later tests; DESIGN.md 7 has what became of it)
@pytest.mark.parametrize("fmt,w,h,kw", [("Y8", 512, 128, {}), ("YUV420P8", 256, 64, dict(aac=48)), ("Y16", 256, 64, dict(order=0)),
                                        ("YUV420P8", 256, 320, dict(aac=48, order=2)), ("YUV444P8", 128, 96, dict(aac=20, dh=True)),
                                        ("YUV422P8", 192, 64, dict(chroma=False, order=0)), ("YUV420P16", 128, 64, dict(luma=False, aac=30)),
                                        ("Y32", 320, 240, dict(order=2))])
def test_pinned_host_frames_skip_the_staging_copies(hip_lib, fmt, w, h, kw):
    """sn_pin_host_buffer + sn_submit_host_to: frames that live in pinned memory go over PCIe as they lie, the output
    straight into the announced planes; pinned and pageable planes mix freely and every route gives get_frame's result."""
    from avisynth_sangnom2_amd import pin_host_array, unpin_host_array
    clip = clip_format(fmt, w, h)
    N = 7
    frames = make_frames(clip, "noise", N, seed0=23)
    parity = [(f * 5) & 1 for f in range(N)]
    with SangNom2(clip, host_depth=4, **kw) as flt:
        ora = Oracle(oracle_cfg(clip, **kw))  # ONE instance: chroma-only processing of subsampled clips carries pool state from frame to frame
        want = [ora.process(frames[f], parity=parity[f]) for f in range(N)]
        nplanes = flt.nplanes
        # one pinned arena holding pitched source and destination planes for all frames
        def arena(shape_of, pad):
            planes, total = [], 0
            for f in range(N):
                for p in range(nplanes):
                    hh, ww = shape_of(p)
                    planes.append((total, hh, ww + pad))
                    total += hh * (ww + pad)
            return np.zeros(total, dtype=clip.dtype), planes
        sbuf, smap = arena(flt.plane_shape_in, 9)
        dbuf, dmap = arena(flt.plane_shape_out, 5)
        def view(buf, m, f, p, shape_of):
            off, hh, pw = m[f * nplanes + p]
            return buf[off:off + hh * pw].reshape(hh, pw)[:, :shape_of(p)[1]]
        pin_host_array(sbuf)
        pin_host_array(dbuf)
        try:
            src_v = [[view(sbuf, smap, f, p, flt.plane_shape_in) for p in range(nplanes)] for f in range(N)]
            dst_v = [[view(dbuf, dmap, f, p, flt.plane_shape_out) for p in range(nplanes)] for f in range(N)]
            for f in range(N):
                for p in range(nplanes):
                    src_v[f][p][...] = frames[f][p]
            # frame 3: pageable source; frame 5: pageable destination, announced at submission all the same
            pageable = [np.zeros(flt.plane_shape_out(p), dtype=clip.dtype) for p in range(nplanes)]
            inflight = []
            for f in range(N):
                if len(inflight) == flt.host_slots():
                    slot, d = inflight.pop(0)
                    flt.collect(slot, d, announced=True)
                d = pageable if f == 5 else dst_v[f]
                inflight.append((flt.submit(frames[f] if f == 3 else src_v[f], parity[f], dst=d), d))
            for slot, d in inflight:
                flt.collect(slot, d, announced=True)
            for f in range(N):
                for p in range(nplanes):
                    got = pageable[p] if f == 5 else np.ascontiguousarray(dst_v[f][p])
                    assert same(want[f][p], got), f"frame {f} plane {p}"
            assert not dbuf.reshape(-1)[[m[0] + m[2] - 1 for m in dmap]].any(), "row padding of the pinned planes was written"
            # the synchronous entry point takes pinned planes as well
            got = flt.get_frame(src_v[0], parity[0], dst=dst_v[1])
            again = ora.process(frames[0], parity=parity[0])  # the eighth frame of that instance
            for p in range(nplanes):
                assert same(again[p], np.ascontiguousarray(got[p]))
        finally:
            unpin_host_array(sbuf)
            unpin_host_array(dbuf)
    with pytest.raises(SangNomError):
        unpin_host_array(sbuf)


@pytest.mark.parametrize("fmt,w,h,kw,bands,warm", BAND_CASES, ids=[f"{c[0]}-{c[1]}x{c[2]}-b{c[4]}" for c in BAND_CASES])
@pytest.mark.parametrize("pattern", ["noise", "checker", "edges", "sine"])
def test_row_bands_match_oracle(hip_lib, monkeypatch, fmt, w, h, kw, bands, warm, pattern):
    """A frame cut into row bands equals the oracle whatever the content: where the run-up has not forgotten the
    guessed state the check sends the frame to the pool path."""
    from avisynth_sangnom2_amd import capi as _capi
    monkeypatch.setitem(_capi.POLICY_DEFAULTS, "small_launches", _capi.SN_SMALL_AUTO)
    clip = clip_format(fmt, w, h)
    ora = Oracle(oracle_cfg(clip, **kw))
    with SangNom2(clip, **kw) as flt:
        flt.set_bands(bands, warm)
        for f, src in enumerate(make_frames(clip, pattern, 3, seed0=11)):
            want = ora.process(src, parity=f & 1)
            got = flt.get_frame(src, parity=f & 1)
            for p in range(len(want)):
                assert same(want[p], got[p]), f"{pattern} frame {f} plane {p}: " + describe_diff(want[p], got[p])
        info = flt.info()
        if bands > 0:
            assert info.banded_frames == 3
        else:  # automatic: after a failed check the following launches skip the bands
            assert 1 <= info.banded_frames <= 3 and (info.banded_frames == 3 or info.band_fallbacks > 0)
        if pattern == "noise" and warm == 0:
            assert info.band_fallbacks == 0  # the default run-up is long enough for noise


@pytest.mark.parametrize("fmt,w,h", [("Y8", 512, 300), ("Y8", 3840, 2160), ("Y16", 512, 300), ("Y32", 512, 300)])
def test_row_bands_that_fail_the_check_are_redone(hip_lib, monkeypatch, fmt, w, h):
    """A run-up of one row leaves nearly every band with a wrong state: the check must notice and the pool path must
    repair every such frame."""
    from avisynth_sangnom2_amd import capi as _capi
    monkeypatch.setitem(_capi.POLICY_DEFAULTS, "small_launches", _capi.SN_SMALL_AUTO)
    clip = clip_format(fmt, w, h)
    ora = Oracle(oracle_cfg(clip))
    with SangNom2(clip) as flt:
        flt.set_bands(16, 1)
        for f, src in enumerate(make_frames(clip, "noise", 2, seed0=3)):
            want = ora.process(src)
            got = flt.get_frame(src)
            assert same(want[0], got[0]), f"frame {f}: " + describe_diff(want[0], got[0])
        info = flt.info()
        assert info.banded_frames == 2 and info.band_fallbacks == 2
        flt.set_bands(-1, 0)  # bands off: the same frames through the pool path / plain sweep
        got = flt.get_frame(src)
        assert same(want[0], got[0])
        assert flt.info().banded_frames == 2


def test_row_bands_pause_after_failed_checks(hip_lib, monkeypatch):
    """Content on which the run-up does not forget the guessed state (a checkerboard) would pay for the bands and for the
    pool path every time: after a failed check the next launches skip the bands."""
    from avisynth_sangnom2_amd import capi as _capi
    monkeypatch.setitem(_capi.POLICY_DEFAULTS, "small_launches", _capi.SN_SMALL_AUTO)
    clip = clip_format("Y8", 512, 400)
    ora = Oracle(oracle_cfg(clip))
    src = synth.frame(clip, "checker", seed=2)
    want = ora.process(src)
    with SangNom2(clip) as flt:
        banded = []
        for f in range(12):
            got = flt.get_frame(src)
            assert same(want[0], got[0]), f"frame {f}"
            banded.append(flt.info().banded_frames)
        info = flt.info()
        if info.band_fallbacks:  # the first frame failed its check: the next eight launches skip the bands, then one more try
            assert banded[0] == banded[8] == 1 and info.banded_frames <= 2


@pytest.mark.parametrize("fmt,w,h,kw", [("YUV420P8", 512, 400, dict(aac=48)), ("YUV420P16", 256, 320, dict(aac=48, fresh_pool=True)),
                                        ("YUV422PS", 256, 256, dict(aac=48))])
def test_row_bands_serve_isolated_planes(hip_lib, monkeypatch, fmt, w, h, kw):
    """isolated_planes: every plane is a plane on its own, so single frames are cut into row bands too (4:2:0 included);
    same bytes as the whole-plane sweeps."""
    from avisynth_sangnom2_amd import capi as _capi
    monkeypatch.setitem(_capi.POLICY_DEFAULTS, "small_launches", _capi.SN_SMALL_AUTO)
    clip = clip_format(fmt, w, h)
    frames = make_frames(clip, "noise", 3, seed0=17)
    with SangNom2(clip, isolated_planes=True, **kw) as flt:
        got = [flt.get_frame(src, parity=f & 1) for f, src in enumerate(frames)]
        info = flt.info()
        assert (info.banded_frames, info.band_fallbacks) == (3, 0)
    with SangNom2(clip, isolated_planes=True, mode="fused", **kw) as flt:
        want = [flt.get_frame(src, parity=f & 1) for f, src in enumerate(frames)]
        assert flt.info().banded_frames == 0
    for f in range(3):
        for p in range(3):
            assert same(want[f][p], got[f][p]), f"frame {f} plane {p}: " + describe_diff(want[f][p], got[f][p])


@pytest.mark.parametrize("fmt,w,h,kw,n", [("Y8", 512, 300, {}, 5), ("Y8", 64, 200, dict(order=0), 7), ("Y16", 1056, 240, dict(aa=20), 3),
                                          ("YUV444PS", 256, 200, dict(aac=48), 4), ("YUV420P8", 512, 320, dict(aac=48), 6),
                                          # 4:2:0: luma in bands, U and V as one chain of two passes per frame (run_group)
                                          ("YUV420P16", 512, 320, dict(aac=48), 4), ("YUV420PS", 256, 200, dict(aac=48), 3),
                                          ("YUV420P8", 1024, 128, dict(aac=20, order=0), 5)])
def test_row_bands_with_several_frames_per_launch(hip_lib, monkeypatch, fmt, w, h, kw, n):
    """A launch of a few frames (a host ring group, a short device batch) is cut into bands as well: frames x bands
    workgroups, one flag per frame.  One of the frames is a checkerboard, which may fail its check on its own."""
    import torch
    from avisynth_sangnom2_amd import capi as _capi
    monkeypatch.setitem(_capi.POLICY_DEFAULTS, "small_launches", _capi.SN_SMALL_AUTO)
    clip = clip_format(fmt, w, h)
    frames = make_frames(clip, "noise", n, seed0=41)
    frames[n // 2] = synth.frame(clip, "checker", seed=9)
    parity = [(f * 3) & 1 for f in range(n)]
    ora = Oracle(oracle_cfg(clip, **kw))
    want = [ora.process(frames[f], parity=parity[f]) for f in range(n)]
    dev = torch.device("cuda:0")
    tdt = {1: torch.uint8, 2: torch.int16, 4: torch.float32}[clip.bytes]
    with SangNom2(clip, max_batch=n, **kw) as flt:
        flt.set_bands(6, 0)
        src = [torch.from_numpy(np.stack([frames[f][p] for f in range(n)]).view({1: np.uint8, 2: np.int16, 4: np.float32}[clip.bytes])).pin_memory().to(dev)
               for p in range(clip.planes)]
        dst = [torch.zeros((n,) + flt.plane_shape_out(p), dtype=tdt, device=dev) for p in range(clip.planes)]
        torch.cuda.synchronize()
        flt.process_batch(src, dst, parity=parity)
        flt.synchronize()
        assert flt.info().banded_frames == n
        for f in range(n):
            for p in range(clip.planes):
                got = to_host(dst[p][f]).view(clip.dtype)
                assert same(want[f][p], got), f"frame {f} plane {p}: " + describe_diff(want[f][p], got)
    # the host ring: groups of one or two frames per launch
    with SangNom2(clip, host_depth=4, **kw) as flt:
        flt.set_bands(5, 0)
        inflight, got = [], []
        for f in range(n):
            if len(inflight) == flt.host_slots():
                got.append(flt.collect(inflight.pop(0)))
            inflight.append(flt.submit(frames[f], parity[f]))
        while inflight:
            got.append(flt.collect(inflight.pop(0)))
        assert flt.info().banded_frames == n
        for f in range(n):
            for p in range(clip.planes):
                assert same(want[f][p], got[f][p]), f"ring frame {f} plane {p}: " + describe_diff(want[f][p], got[f][p])


@pytest.mark.parametrize("fmt,w,h,kw", [("Y8", 512, 300, {}), ("YUV420P8", 256, 320, dict(aac=48)), ("Y16", 256, 240, {})])
def test_filter_instances_on_their_own_threads(hip_lib, monkeypatch, fmt, w, h, kw):
    """MT_MULTI_INSTANCE (SangNom2.h:63-66): the host runs several instances of the filter at once, each on its own thread
    with its own context.  Four threads, each with a context of its own, call the synchronous entry point on their own
    frames (the latency path: bands, checks, guarded pool kernels, each context on its own stream and scratch)."""
    import threading
    from avisynth_sangnom2_amd import capi as _capi
    monkeypatch.setitem(_capi.POLICY_DEFAULTS, "small_launches", _capi.SN_SMALL_AUTO)
    clip = clip_format(fmt, w, h)
    T, N = 4, 6
    frames = [[synth.frame(clip, "noise" if (t + f) % 3 else "checker", seed=100 * t + f) for f in range(N)] for t in range(T)]
    want = []
    for t in range(T):
        ora = Oracle(oracle_cfg(clip, **kw))
        want.append([ora.process(frames[t][f], parity=f & 1) for f in range(N)])
    got = [[None] * N for _ in range(T)]
    errors = []

    def work(t):
        try:
            with SangNom2(clip, **kw) as flt:
                for f in range(N):
                    got[t][f] = flt.get_frame(frames[t][f], parity=f & 1)
        except Exception as e:  # noqa: BLE001
            errors.append((t, repr(e)))

    ths = [threading.Thread(target=work, args=(t,)) for t in range(T)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    assert not errors, errors
    for t in range(T):
        for f in range(N):
            for p in range(len(want[t][f])):
                assert same(want[t][f][p], got[t][f][p]), f"thread {t} frame {f} plane {p}"


def test_launches_of_hundreds_of_small_frames_stay_on_the_whole_plane_sweeps(hip_lib, monkeypatch):
    """More than 512 frames in one launch of an isolated-planes context (which has a scratch slot for each of them): the band
    count works out as zero there -- once a division by it -- and the launch must take the whole-plane sweeps."""
    import torch
    from avisynth_sangnom2_amd import capi as _capi
    monkeypatch.setitem(_capi.POLICY_DEFAULTS, "small_launches", _capi.SN_SMALL_AUTO)
    clip = clip_format("YUV420P8", 128, 80)
    N = 600
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(77)
    src = [torch.randint(0, 256, (N, 80 >> (1 if p else 0), 128 >> (1 if p else 0)), device=dev, generator=g, dtype=torch.uint8) for p in range(3)]
    outs = {}
    for mode in ("auto", "fused"):
        with SangNom2(clip, max_batch=N, mode=mode, isolated_planes=True, aac=48) as flt:
            dst = [torch.zeros_like(s) for s in src]
            torch.cuda.synchronize()
            flt.process_batch(src, dst)
            flt.synchronize()
            info = flt.info()
            assert (info.fused_frames, info.banded_frames) == (N, 0)
            outs[mode] = [to_host(d) for d in dst]
    for p in range(3):
        assert np.array_equal(outs["auto"][p], outs["fused"][p])


@pytest.mark.parametrize("fmt", ["YUV420P8", "YUV422P8", "YUV420P16", "YUV420PS"])
@pytest.mark.parametrize("w,h", [(512, 400), (992, 720), (1472, 1000), (544, 400), (1024, 720), (960, 400), (1920, 300), (2912, 200)])
def test_coupled_sweeps_equal_the_pool_path_where_the_cone_reaches_the_last_columns(hip_lib, fmt, w, h):
    """Tall frames of widths whose last strip ends on lanes 62 / 63 (64 + 60 k lanes: 512, 992, 1472) and of their
    neighbours: the hand-off's dependency cone then includes the last columns of the pool, which short test frames
    never reach (the 8-bit sweeps once read zeros there).  Whole-plane sweeps against the pool path, byte for byte.
    960, 1920 (and 3840, in the full-size tests): the chroma region ends two real lanes before a wave's ghosts -- the 16-bit
    sweep's fetch ring and the float sweep's packed fetch / packed store (kPacked) with the other region waves in their loops
    without a fetch; 992: only the ghosts take stale values; 512, 1472, 2912: the region ends in the middle of a wave (kFetch)."""
    hh = h if fmt != "YUV422P8" else h // 2
    clip = clip_format(fmt, w, hh)
    kw = dict(aa=128, aac=128)
    for pattern in ("edges", "noise"):
        src = synth.frame(clip, pattern, seed=5)
        outs = {}
        for mode in ("fused", "pool"):
            with SangNom2(clip, mode=mode, **kw) as flt:
                outs[mode] = flt.get_frame(src)
        for p in range(3):
            assert same(outs["pool"][p], outs["fused"][p]), f"{pattern} plane {p}: " + describe_diff(outs["pool"][p], outs["fused"][p])


# The U and V passes of an 8-bit 4:2:0 frame as ONE sweep (sn_fused_u8_uv.hip).  Geometries by where the chroma region's
# right edge falls among the strips of the luma-wide pool (62 lanes of 8 columns in the first strip, 60 in every later one):
# inside a single strip (256, 512), exactly on a seam so that TWO waves hold both chroma and stale lanes (992), two lanes
# into a strip (1024, 3840), in the middle of one (1280, 2560), with one to eight strips, and frames just tall enough.
UV_GEOMETRIES = [(256, 64), (512, 640), (640, 32), (992, 720), (1024, 360), (1280, 720), (1472, 200), (1920, 1080), (2560, 96), (3200, 128),
                 (3840, 32), (3840, 1080)]


@pytest.mark.gpu
@pytest.mark.parametrize("w,h", UV_GEOMETRIES, ids=[f"{w}x{h}" for w, h in UV_GEOMETRIES])
def test_u_and_v_as_one_sweep_match_oracle_and_the_two_sweep_form(hip_lib, w, h):
    """Three frames per launch (every field order and parity), against one oracle instance per configuration and against the
    two-sweep form (sn_policy.chroma_sweeps = 1); the one-sweep path is asserted to have run."""
    import torch
    clip = clip_format("YUV420P8", w, h)
    dev = torch.device("cuda:0")
    for kw, parity, pattern in ((dict(order=1, aa=48, aac=48), [1, 1, 1], "noise"), (dict(order=0, aa=128, aac=128), [1, 0, 1], "edges"),
                                (dict(order=2, aa=0, aac=10), [1, 1, 1], "noise"), (dict(order=1, aa=48, aac=48, dh=True), [1, 1, 1], "checker")):
        frames = [synth.frame(clip, pattern, seed=3 * w + h + i) for i in range(3)]
        ora = Oracle(oracle_cfg(clip, **kw))
        want = [ora.process(fr, parity=parity[i]) for i, fr in enumerate(frames)]
        src = [torch.from_numpy(np.stack([fr[p] for fr in frames])).pin_memory().to(dev) for p in range(3)]
        got = {}
        for name, extra in (("one", {}), ("two", dict(chroma_sweeps=1))):
            with SangNom2(clip, max_batch=3, mode="fused", **kw, **extra) as flt:
                dst = [torch.zeros((3,) + flt.plane_shape_out(p), dtype=torch.uint8, device=dev) for p in range(3)]
                torch.cuda.synchronize()
                flt.process_batch(src, dst, parity=parity)
                flt.synchronize()
                assert flt.info().uv_sweeps == (1 if name == "one" else 0)
                got[name] = [to_host(d) for d in dst]
        for f in range(3):
            for p in range(3):
                assert same(want[f][p], got["one"][p][f]), f"{kw} frame {f} plane {p}: " + describe_diff(want[f][p], got["one"][p][f])
                assert np.array_equal(got["one"][p][f], got["two"][p][f])


@pytest.mark.gpu
def test_geometries_the_one_sweep_form_does_not_take_keep_the_two_chroma_sweeps(hip_lib):
    """Frames of a few lines and pools of more than eight strips fall back to a sweep per chroma plane; still exact."""
    for fmt, w, h in (("YUV420P8", 512, 24), ("YUV420P8", 4096, 64), ("YUV422P8", 512, 12)):
        clip = clip_format(fmt, w, h)
        src = synth.frame(clip, "noise", seed=9)
        want = Oracle(oracle_cfg(clip, aa=48, aac=48)).process(src)
        with SangNom2(clip, mode="fused", aa=48, aac=48) as flt:
            got = flt.get_frame(src)
            assert flt.info().uv_sweeps == 0 and flt.info().fused_frames == 1
        for p in range(3):
            assert same(want[p], got[p])


UV_422 = [(256, 32), (512, 320), (992, 360), (1024, 180), (1920, 540), (3840, 64), (3840, 1080)]


@pytest.mark.gpu
@pytest.mark.parametrize("w,h", UV_422, ids=[f"{w}x{h}" for w, h in UV_422])
def test_u_and_v_as_one_sweep_for_422(hip_lib, w, h):
    """4:2:2: the chroma planes are as tall as luma, so the shared pool has no row below their last one -- the U pass has no
    extra row and V's last row takes nothing from it -- and the dependency cone covers most of the stale region for most of
    the sweep (the waves right of the region stay to the end).  Against the oracle and the two-sweep form."""
    import torch
    clip = clip_format("YUV422P8", w, h)
    dev = torch.device("cuda:0")
    for kw, parity, pattern in ((dict(order=1, aa=48, aac=48), [1, 1], "noise"), (dict(order=0, aa=128, aac=128), [0, 1], "edges"),
                                (dict(order=2, aa=10, aac=20, dh=True), [1, 1], "checker")):
        frames = [synth.frame(clip, pattern, seed=5 * w + h + i) for i in range(2)]
        ora = Oracle(oracle_cfg(clip, **kw))
        want = [ora.process(fr, parity=parity[i]) for i, fr in enumerate(frames)]
        src = [torch.from_numpy(np.stack([fr[p] for fr in frames])).pin_memory().to(dev) for p in range(3)]
        got = {}
        for name, extra in (("one", {}), ("two", dict(chroma_sweeps=1))):
            with SangNom2(clip, max_batch=2, mode="fused", **kw, **extra) as flt:
                dst = [torch.zeros((2,) + flt.plane_shape_out(p), dtype=torch.uint8, device=dev) for p in range(3)]
                torch.cuda.synchronize()
                flt.process_batch(src, dst, parity=parity)
                flt.synchronize()
                assert flt.info().uv_sweeps == (1 if name == "one" else 0)
                got[name] = [to_host(d) for d in dst]
        for f in range(2):
            for p in range(3):
                assert same(want[f][p], got["one"][p][f]), f"{kw} frame {f} plane {p}: " + describe_diff(want[f][p], got["one"][p][f])
                assert np.array_equal(got["one"][p][f], got["two"][p][f])


@pytest.mark.gpu
def test_chroma_sweeps_can_change_during_a_contexts_life(hip_lib):
    """sn_set_policy(chroma_sweeps): a context that starts with U and V as one sweep holds the luma -> U pool only; switching to a
    sweep per plane allocates the U -> V pool on first use, switching back leaves it idle.  Same bytes every time."""
    import torch
    clip = clip_format("YUV420P8", 1024, 360)
    dev = torch.device("cuda:0")
    kw = dict(aa=48, aac=48)
    frames = [synth.frame(clip, "noise", seed=40 + i) for i in range(4)]
    ora = Oracle(oracle_cfg(clip, **kw))
    want = [ora.process(fr) for fr in frames]
    src = [torch.from_numpy(np.stack([fr[p] for fr in frames])).pin_memory().to(dev) for p in range(3)]
    with SangNom2(clip, max_batch=4, mode="fused", **kw) as flt:
        for sweeps, uv_before, uv_after in ((1, 0, 0), (0, 0, 1), (1, 1, 1), (0, 1, 1)):
            flt.set_policy(chroma_sweeps=sweeps)
            assert flt.get_policy().chroma_sweeps == sweeps
            assert flt.info().uv_sweeps == uv_before
            dst = [torch.zeros((4,) + flt.plane_shape_out(p), dtype=torch.uint8, device=dev) for p in range(3)]
            torch.cuda.synchronize()
            flt.process_batch(src, dst)
            flt.synchronize()
            assert flt.info().uv_sweeps == uv_after
            for f in range(4):
                for p in range(3):
                    assert same(want[f][p], to_host(dst[p][f])), f"chroma_sweeps={sweeps} frame {f} plane {p}"
        if True:  # the U -> V hand-off exists now (the two-sweep form ran): readable, and not all zero inside the cone
            rows = flt.read_coupled_rows(1)
            assert rows.any()
