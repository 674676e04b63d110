"""C-ABI checks that need no GPU: the library loads, exports every symbol include/sangnom_hip.h
declares, validates arguments like Create_SangNom2 (/root/reference/src/SangNom2.cpp:407-422), and
refuses to compute without a device (no CPU fallback)."""
import ctypes
import os
import re

import pytest

from avisynth_sangnom2_amd import ClipFormat, SangNom2, SangNomError, capi, clip_format

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cfg(**kw):
    base = dict(struct_size=ctypes.sizeof(capi.SnConfig), width=64, height=32, bytes_per_sample=1,
                bits_per_sample=8, num_planes=1, sub_w=0, sub_h=0, order=1, aa=48, aac=0, dh=0, luma=1,
                chroma=1, device=0, max_batch=1, mode=0, host_depth=0, isolated_planes=0, fresh_pool=0, stream=None)
    base.update(kw)
    return capi.SnConfig(**base)


def test_library_exports_every_declared_symbol(hip_lib):
    header = open(os.path.join(ROOT, "include", "sangnom_hip.h")).read()
    declared = set(re.findall(r"\b(sn_[a-z_]+)\s*\(", header))
    assert declared == set(capi.EXPORTS), declared ^ set(capi.EXPORTS)
    for name in declared:
        assert hasattr(hip_lib, name), f"libsangnom_hip.so does not export {name}"
    assert hip_lib.sn_abi_version() == 4  # 2: + sn_aa_* (round 2); 3: + sn_policy (round 3); 4: sn_info.chain_redone, sn_policy.chroma_sweeps (round 4)


def test_the_library_reads_no_environment_variable(hip_lib):
    """Scheduling knobs are part of the ABI (sn_policy); the shipped library has no getenv -- only a -DSN_TEST_HOOKS
    build does.  And a policy is checked before any device is touched."""
    import subprocess
    syms = subprocess.run(["nm", "-D", "--undefined-only", capi.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "getenv" not in syms, "libsangnom_hip.so imports getenv"
    h = ctypes.c_void_p()
    cfg = _cfg()
    for bad in (dict(small_launches=7), dict(chain=3), dict(chain=-2), dict(copy_threads=99), dict(scratch_budget_mb=-1)):
        pol = capi.SnPolicy(struct_size=ctypes.sizeof(capi.SnPolicy), **bad)
        assert hip_lib.sn_create_with_policy(ctypes.byref(cfg), ctypes.byref(pol), ctypes.byref(h)) == capi.SN_ERR_INVALID_ARG
        assert b"sn_policy" in hip_lib.sn_last_error(None)
    pol = capi.SnPolicy(struct_size=4)
    assert hip_lib.sn_create_with_policy(ctypes.byref(cfg), ctypes.byref(pol), ctypes.byref(h)) == capi.SN_ERR_INVALID_ARG


def test_validate_matches_reference_messages(hip_lib):
    def v(**kw):
        msg = ctypes.create_string_buffer(256)
        c = _cfg(**kw)
        return hip_lib.sn_validate(ctypes.byref(c), msg, 256), msg.value.decode()

    assert v() == (capi.SN_OK, "")
    assert v(height=31) == (capi.SN_ERR_CONFIG, "SangNom2: height must be even.")
    assert v(height=34, num_planes=3, sub_w=1, sub_h=1) == (capi.SN_ERR_CONFIG, "SangNom2: height must be mod4.")
    assert v(order=-1) == (capi.SN_ERR_CONFIG, "SangNom2: order must be between 0..2.")
    assert v(aa=129) == (capi.SN_ERR_CONFIG, "SangNom2: aa must be between 0..128.")
    assert v(aac=500) == (capi.SN_ERR_CONFIG, "SangNom2: aac must be between 0..128.")
    assert v(struct_size=4)[0] == capi.SN_ERR_INVALID_ARG
    assert v(bytes_per_sample=3)[0] == capi.SN_ERR_INVALID_ARG
    assert v(bytes_per_sample=2, bits_per_sample=8)[0] == capi.SN_ERR_INVALID_ARG


def test_python_mirror_raises_reference_text(hip_lib):
    with pytest.raises(SangNomError, match="height must be even"):
        SangNom2(clip_format("Y8", 64, 33))
    with pytest.raises(SangNomError, match=r"opt must be between -1\.\.2"):
        SangNom2(clip_format("Y8", 64, 32), opt=5)


def test_no_device_means_loud_failure_not_fallback(hip_lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present; the no-device path cannot be exercised")
    h = ctypes.c_void_p()
    c = _cfg()
    rc = hip_lib.sn_create(ctypes.byref(c), ctypes.byref(h))
    assert rc == capi.SN_ERR_NO_DEVICE and not h.value
    assert b"no HIP device" in hip_lib.sn_last_error(None)
    with pytest.raises(SangNomError, match="no HIP device"):
        SangNom2(ClipFormat(64, 32))


def test_null_arguments_are_rejected(hip_lib):
    assert hip_lib.sn_create(None, None) == capi.SN_ERR_INVALID_ARG
    assert hip_lib.sn_synchronize(None) == capi.SN_ERR_INVALID_ARG
    assert hip_lib.sn_get_stream(None) is None
    hip_lib.sn_destroy(None)  # no-op


def test_product_never_imports_the_oracle():
    """The shipped package must not reach into oracle/ (the oracle is test infrastructure)."""
    pkg = os.path.join(ROOT, "avisynth_sangnom2_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp")) or f == "Makefile":
                text = open(os.path.join(dirpath, f)).read()
                for needle in ("import oracle", "from oracle", "sangnom_oracle", "oracle/"):
                    assert needle not in text, f"{f} refers to the oracle ({needle!r})"


def test_staging_copy_pool_is_race_free_under_tsan():
    """sn::Copier (the host ring's staging copies on worker threads) under ThreadSanitizer, every byte checked."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "host"), "copier_test"])
    for workers in (0, 3, 7):
        r = subprocess.run([os.path.join(root, "host", "copier_test"), str(workers), "40"], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and "ThreadSanitizer" not in r.stderr, (r.stdout[-500:], r.stderr[-2000:])


def test_header_is_plain_c_and_every_entry_point_links(tmp_path):
    """include/sangnom_hip.h compiled as C99 -pedantic; the program links against libsangnom_hip.so and runs the
    entry points that need no GPU."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    capi.build()
    exe = str(tmp_path / "abi_check")
    libdir = os.path.join(root, "avisynth_sangnom2_amd")
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(root, "include"),
                           os.path.join(root, "tests", "c", "abi_check.c"), "-o", exe, "-L", libdir, "-lsangnom_hip",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath-link,/opt/rocm/lib"])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "ok" in r.stdout, (r.returncode, r.stdout, r.stderr)


def test_chain_hand_off_between_workgroups_drains_its_stores_before_the_barrier():
    """Release side of the chains over several workgroups per buffer (sn_pool_kernels.hip, chain_release_barrier): in
    the gfx950 ISA of every k_smooth_*_chain<true> kernel an `s_waitcnt vmcnt(0)` stands in front of the round loop's
    s_barrier, so that a workgroup's rows have landed before its round counter is published.  Checked in the ISA because
    it is the compiler that places (or drops) the wait: round 3's build had it in the 8-bit kernel by accident only."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "check_chain_release.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-2000:])
    assert r.stdout.count("0 without a preceding s_waitcnt vmcnt(0)") == 3, r.stdout
