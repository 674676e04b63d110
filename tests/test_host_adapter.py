"""The C++ host-side adapter (host/sangnom2_filter.hpp) driven like a script engine drives the
reference plugin: SangNom2(clip, order, aa, aac, threads, dh, luma, chroma, opt) then GetFrame(n)
(/root/reference/src/SangNom2.cpp:399-435, :332-397).  Frames go through pitched host frames, the
C ABI and the HIP kernels; results are compared with the oracle."""
import os
import struct
import subprocess

import numpy as np
import pytest

from avisynth_sangnom2_amd import clip_format, synth
from oracle.oracle import Oracle
from tests.util import oracle_cfg, same

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "host", "sn_host_test")


def _build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "host"), "sn_host_test"])


def _run(tmp_path, clip, kw, frames, parities, extra=()):
    _build()
    hdr = [clip.width, clip.height, clip.bytes, clip.bits, clip.planes, clip.subw, clip.subh,
           kw.get("order", 1), kw.get("aa", 48), kw.get("aac", 0), int(kw.get("dh", False)),
           int(kw.get("luma", True)), int(kw.get("chroma", True)), len(frames)]
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(fin, "wb") as f:
        f.write(struct.pack("<14i", *hdr))
        for fr, par in zip(frames, parities):
            f.write(struct.pack("<i", par))
            for pl in fr:
                f.write(np.ascontiguousarray(pl).tobytes())
    env = dict(os.environ, SN_HOST_TEST_SWEEPS="1")  # small clips: reach the fused sweeps, not auto mode's small-launch paths
    r = subprocess.run([BIN, fin, fout, *[str(x) for x in extra]], capture_output=True, text=True, timeout=300, env=env)
    return r, fout


def test_constructor_errors_carry_the_reference_text(tmp_path):
    """Validation happens before any device is touched, so this runs without a GPU."""
    for fmt, w, h, kw, text in (
            ("Y8", 64, 31, {}, "SangNom2: height must be even."),
            ("YUV420P8", 64, 34, {}, "SangNom2: height must be mod4."),
            ("Y8", 64, 32, dict(order=5), "SangNom2: order must be between 0..2."),
            ("Y8", 64, 32, dict(aa=200), "SangNom2: aa must be between 0..128."),
            ("Y8", 64, 32, dict(aac=129), "SangNom2: aac must be between 0..128.")):
        r, _ = _run(tmp_path, clip_format(fmt, w, h), kw, [], [])
        assert r.returncode == 3 and r.stdout.strip() == text, (r.returncode, r.stdout, r.stderr)


@pytest.mark.gpu
@pytest.mark.parametrize("fmt,w,h,kw", [
    ("Y8", 256, 64, dict(order=1, aa=48)),
    ("Y8", 100, 40, dict(order=0, aa=20)),
    ("YUV420P8", 128, 64, dict(aac=48, order=2)),
    ("YUV420P16", 96, 32, dict(aac=48)),
    ("YUV444PS", 64, 24, dict(dh=True, aac=48)),
    ("YUV420P8", 128, 32, dict(chroma=False)),
])
def test_getframe_matches_oracle(tmp_path, fmt, w, h, kw):
    clip = clip_format(fmt, w, h)
    frames = [synth.frame(clip, "noise" if i % 2 == 0 else "edges", seed=40 + i) for i in range(3)]
    parities = [1, 0, 1]
    r, fout = _run(tmp_path, clip, kw, frames, parities)
    assert r.returncode == 0, (r.stdout, r.stderr)
    ora = Oracle(oracle_cfg(clip, **kw))
    raw = np.fromfile(fout, dtype=np.uint8)
    pos = 0
    for f, (fr, par) in enumerate(zip(frames, parities)):
        want = ora.process(fr, parity=par)
        for p, wpl in enumerate(want):
            got = raw[pos:pos + wpl.nbytes].view(wpl.dtype).reshape(wpl.shape)
            pos += wpl.nbytes
            assert same(wpl, got), f"{fmt} frame {f} plane {p}"
    assert pos == raw.size


@pytest.mark.gpu
@pytest.mark.parametrize("fmt,w,h,kw", [
    ("Y8", 256, 64, dict(order=0, aa=48)),
    ("YUV420P8", 128, 64, dict(aac=48)),
    ("Y8", 100, 40, dict(aa=20)),              # history-carrying: GetFrame stays synchronous
])
def test_getframe_with_lookahead_and_a_seek(tmp_path, fmt, w, h, kw):
    """Args::lookahead = 4: frames are requested ahead of the caller over the host ring; the request order
    0 1 2 6 7 8 3 4 (two seeks) must give the frames a synchronous filter gives (history-free clips: every
    frame on its own; the history-carrying clip is checked against one oracle fed in request order)."""
    clip = clip_format(fmt, w, h)
    N = 9
    frames = [synth.frame(clip, "noise", seed=70 + i) for i in range(N)]
    parities = [i & 1 for i in range(N)]
    order = [0, 1, 2, 6, 7, 8, 3, 4]
    r, fout = _run(tmp_path, clip, kw, frames, parities, extra=[4] + order)
    assert r.returncode == 0, (r.stdout, r.stderr)
    raw = np.fromfile(fout, dtype=np.uint8)
    ora = Oracle(oracle_cfg(clip, **kw))
    pos = 0
    for n in order:
        want = ora.process(frames[n], parity=parities[n])
        for p, wpl in enumerate(want):
            got = raw[pos:pos + wpl.nbytes].view(wpl.dtype).reshape(wpl.shape)
            pos += wpl.nbytes
            assert same(wpl, got), f"{fmt} request {n} plane {p}"
    assert pos == raw.size


@pytest.mark.gpu
@pytest.mark.parametrize("fmt,w,h,kw", [("Y8", 128, 64, dict(aa=48)), ("YUV420P8", 128, 64, dict(aa=48, aac=48)),
                                        ("Y16", 96, 80, dict(aa=30, order=2))])
def test_plugin_level_anti_aliasing_function(tmp_path, fmt, w, h, kw):
    """SangNomAA(clip, order, aa, aac) of the plugin (sangnom::AAFilter over sn_aa_process_host) == the script
    TurnLeft().SangNom2(...).TurnRight().SangNom2(...) run with two oracle instances and numpy turns."""
    from avisynth_sangnom2_amd import ClipFormat
    clip = clip_format(fmt, w, h)
    turned = ClipFormat(width=h, height=w, bytes=clip.bytes, bits=clip.bits, planes=clip.planes, subw=clip.subh, subh=clip.subw)
    frames = [synth.frame(clip, "noise", seed=11 + i) for i in range(3)]
    r, fout = _run(tmp_path, clip, kw, frames, [1, 1, 1], extra=["aa"])
    assert r.returncode == 0, (r.stdout, r.stderr)
    first, second = Oracle(oracle_cfg(turned, **kw)), Oracle(oracle_cfg(clip, **kw))
    raw = np.fromfile(fout, dtype=np.uint8)
    pos = 0
    for f, fr in enumerate(frames):
        a = first.process([np.ascontiguousarray(np.rot90(pl, k=1)) for pl in fr])
        want = second.process([np.ascontiguousarray(np.rot90(pl, k=-1)) for pl in a])
        for p, wpl in enumerate(want):
            got = raw[pos:pos + wpl.nbytes].view(wpl.dtype).reshape(wpl.shape)
            pos += wpl.nbytes
            assert same(wpl, got), f"{fmt} frame {f} plane {p}"
    assert pos == raw.size


def test_anti_aliasing_function_checks_the_turned_clip_too(tmp_path):
    r, _ = _run(tmp_path, clip_format("Y8", 63, 64), {}, [], [], extra=["aa"])
    assert r.returncode == 3 and "height must be even" in r.stdout, (r.returncode, r.stdout)


@pytest.mark.gpu
@pytest.mark.parametrize("dh,lookahead", [(False, 1), (True, 1), (False, 4)])
def test_alpha_plane_is_passed_through(tmp_path, dh, lookahead):
    """A YUVA clip: the reference leaves the fourth plane of the new frame unwritten (src/SangNom2.cpp:346-348); the
    adapter copies it (dh: every source line twice).  Y, U, V are the oracle's."""
    w, h = 128, 64
    clip = clip_format("YUV420P8", w, h)
    kw = dict(aac=48, dh=dh)
    rng = np.random.default_rng(12)
    frames, alphas = [], []
    for i in range(5):
        fr = synth.frame(clip, "noise", seed=60 + i)
        alphas.append(rng.integers(0, 256, (h, w), dtype=np.uint8))
        frames.append(list(fr) + [alphas[-1]])
    yuva = clip_format("YUV420P8", w, h)
    yuva.planes = 4
    r, fout = _run(tmp_path, yuva, kw, frames, [1] * 5, extra=[lookahead])
    assert r.returncode == 0, (r.stdout, r.stderr)
    raw = np.fromfile(fout, dtype=np.uint8)
    pos = 0
    for f in range(5):
        want = Oracle(oracle_cfg(clip, **kw)).process(frames[f][:3])
        want.append(np.repeat(alphas[f], 2, axis=0) if dh else alphas[f])
        for p, wpl in enumerate(want):
            got = raw[pos:pos + wpl.nbytes].view(wpl.dtype).reshape(wpl.shape)
            pos += wpl.nbytes
            assert same(wpl, got), f"frame {f} plane {p}"
    assert pos == raw.size
