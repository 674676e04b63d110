"""ctypes binding of libsangnom_hip.so (include/sangnom_hip.h).

This module is plumbing: it loads the in-tree shared library and declares the C ABI.  It fails
loudly (ImportError at load()) if the library has not been built -- there is no Python or CPU
fallback for the kernels.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsangnom_hip.so")

SN_OK, SN_ERR_INVALID_ARG, SN_ERR_CONFIG, SN_ERR_HIP, SN_ERR_NO_DEVICE, SN_ERR_UNSUPPORTED, SN_ERR_BUSY = range(7)
SN_MODE_AUTO, SN_MODE_POOL, SN_MODE_FUSED = range(3)
MODES = {"auto": SN_MODE_AUTO, "pool": SN_MODE_POOL, "fused": SN_MODE_FUSED}

# every symbol include/sangnom_hip.h declares
EXPORTS = (
    "sn_abi_version", "sn_validate", "sn_create", "sn_destroy", "sn_last_error",
    "sn_process_host", "sn_process_device", "sn_process_device_strided", "sn_synchronize",
    "sn_get_stream", "sn_get_info", "sn_debug_read_pool", "sn_debug_read_coupled_rows",
    "sn_host_slots", "sn_submit_host", "sn_collect_host", "sn_turn_device",
    "sn_aa_create", "sn_aa_process_host", "sn_aa_last_error", "sn_aa_destroy",
    "sn_pin_host_buffer", "sn_unpin_host_buffer", "sn_submit_host_to", "sn_debug_set_bands",
    "sn_create_with_policy", "sn_get_policy", "sn_set_policy", "sn_aa_create_with_policy",
    "sn_debug_raise_chain_fault",
)

SN_SMALL_AUTO, SN_SMALL_SWEEP = 0, 1
# What a filter object asks for when its caller says nothing (all zeros = the library's defaults).  The test suite
# changes entries here (small clips would otherwise never reach the whole-plane sweeps); the library itself reads no
# environment variable.
POLICY_DEFAULTS = {"small_launches": SN_SMALL_AUTO, "chain": 0, "copy_threads": 0, "scratch_budget_mb": 0, "chroma_sweeps": 0}


class SnConfig(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in (
        "struct_size", "width", "height", "bytes_per_sample", "bits_per_sample", "num_planes",
        "sub_w", "sub_h", "order", "aa", "aac", "dh", "luma", "chroma", "device", "max_batch",
        "mode", "host_depth", "isolated_planes", "fresh_pool")] + [("stream", ctypes.c_void_p)]


class SnPolicy(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("struct_size", "small_launches", "chain", "copy_threads", "scratch_budget_mb", "chroma_sweeps")] + [
        ("reserved", ctypes.c_int32 * 2)]


def policy(**over) -> "SnPolicy":
    """sn_policy from POLICY_DEFAULTS with the given fields replaced (None = keep the default)."""
    v = dict(POLICY_DEFAULTS)
    v.update({k: x for k, x in over.items() if x is not None})
    return SnPolicy(struct_size=ctypes.sizeof(SnPolicy), **{k: int(x) for k, x in v.items()})


class SnInfo(ctypes.Structure):
    _fields_ = [("struct_size", ctypes.c_int32), ("out_height", ctypes.c_int32),
                ("pool_stride", ctypes.c_int32), ("pool_rows", ctypes.c_int32),
                ("fused_eligible", ctypes.c_int32), ("history_free", ctypes.c_int32),
                ("frames", ctypes.c_int64), ("fused_frames", ctypes.c_int64),
                ("coupled_rows", ctypes.c_int32), ("uv_sweeps", ctypes.c_int32),
                ("threshold", ctypes.c_double * 3),
                ("banded_frames", ctypes.c_int64), ("band_fallbacks", ctypes.c_int64),
                ("chained_frames", ctypes.c_int64), ("chain_redone", ctypes.c_int64)]


def build(force: bool = False) -> str:
    """Compile libsangnom_hip.so for gfx950 with hipcc (csrc/Makefile)."""
    args = ["make", "-s", "-C", os.path.join(_HERE, "csrc")]
    if force:
        subprocess.check_call(args + ["clean"])
    subprocess.check_call(args + ["-j4"])
    return LIB_PATH


_lib = None


def _share_torch_hip_runtime():
    """One process must hold ONE HIP runtime.  PyTorch-ROCm wheels bundle their own libamdhip64.so
    (SONAME libamdhip64.so.7, the same SONAME libsangnom_hip.so needs); if the system copy were
    loaded first, a later `import torch` would load a second runtime and find no GPU.  So when torch
    is installed its runtime is loaded first and libsangnom_hip.so binds to it by SONAME.  Without
    torch (a plain C/C++ host such as the AviSynth adapter) the RUNPATH copy under /opt/rocm is used."""
    try:
        import torch  # noqa: F401
    except Exception:
        return
    cand = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)


def load():
    """Load the library and declare prototypes.  Raises ImportError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    _share_torch_hip_runtime()
    L = ctypes.CDLL(LIB_PATH)
    vp, i32, i64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64
    p3v, p3i, p3l = ctypes.POINTER(vp), ctypes.POINTER(i32), ctypes.POINTER(i64)
    L.sn_abi_version.restype = ctypes.c_int
    L.sn_validate.argtypes = [ctypes.POINTER(SnConfig), ctypes.c_char_p, ctypes.c_size_t]
    L.sn_create.argtypes = [ctypes.POINTER(SnConfig), ctypes.POINTER(vp)]
    L.sn_create_with_policy.argtypes = [ctypes.POINTER(SnConfig), ctypes.POINTER(SnPolicy), ctypes.POINTER(vp)]
    L.sn_get_policy.argtypes = [vp, ctypes.POINTER(SnPolicy)]
    L.sn_set_policy.argtypes = [vp, ctypes.POINTER(SnPolicy)]
    L.sn_aa_create_with_policy.argtypes = [ctypes.POINTER(SnConfig), ctypes.POINTER(SnPolicy), ctypes.POINTER(vp)]
    L.sn_destroy.argtypes = [vp]
    L.sn_destroy.restype = None
    L.sn_last_error.argtypes = [vp]
    L.sn_last_error.restype = ctypes.c_char_p
    L.sn_process_host.argtypes = [vp, p3v, p3i, p3v, p3i, i32]
    L.sn_process_device.argtypes = [vp, p3v, p3i, p3v, p3i, i32]
    L.sn_host_slots.argtypes = [vp]
    L.sn_turn_device.argtypes = [vp, i32, i32, vp, ctypes.c_int64, i32, i32, i32, vp, ctypes.c_int64, i32]
    L.sn_submit_host.argtypes = [vp, p3v, p3i, i32, ctypes.POINTER(i32)]
    L.sn_collect_host.argtypes = [vp, i32, p3v, p3i]
    L.sn_process_device_strided.argtypes = [vp, i32, p3v, p3l, p3i, p3v, p3l, p3i, p3i]
    L.sn_synchronize.argtypes = [vp]
    L.sn_get_stream.argtypes = [vp]
    L.sn_get_stream.restype = vp
    L.sn_get_info.argtypes = [vp, ctypes.POINTER(SnInfo)]
    L.sn_debug_read_pool.argtypes = [vp, i32, vp, ctypes.c_size_t]
    L.sn_debug_read_coupled_rows.argtypes = [vp, i32, vp, ctypes.c_size_t]
    L.sn_debug_set_bands.argtypes = [vp, i32, i32]
    L.sn_debug_raise_chain_fault.argtypes = [vp]
    L.sn_pin_host_buffer.argtypes = [vp, ctypes.c_size_t]
    L.sn_unpin_host_buffer.argtypes = [vp]
    L.sn_submit_host_to.argtypes = [vp, p3v, p3i, p3v, p3i, i32, ctypes.POINTER(i32)]
    L.sn_aa_create.argtypes = [ctypes.POINTER(SnConfig), ctypes.POINTER(vp)]
    L.sn_aa_process_host.argtypes = [vp, p3v, p3i, p3v, p3i, i32]
    L.sn_aa_last_error.argtypes = [vp]
    L.sn_aa_last_error.restype = ctypes.c_char_p
    L.sn_aa_destroy.argtypes = [vp]
    L.sn_aa_destroy.restype = None
    _lib = L
    return L
