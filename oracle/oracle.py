"""ctypes front-end of the C oracle (oracle/sangnom_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; the
product package (avisynth_sangnom2_amd) never does.  See sangnom_oracle.h: PARITY UNPINNED.

Reference being restated: /root/reference/src/SangNom2.cpp:25-397 (opt=0 path).
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# SN_ORACLE_LIB: another build of the same sources, e.g. libsangnom_oracle_asan.so (tests/test_oracle.py runs the
# restatement under AddressSanitizer / UBSan that way, in a child process that preloads the sanitizer runtime)
_LIB_PATH = os.environ.get("SN_ORACLE_LIB") or os.path.join(_HERE, "libsangnom_oracle.so")


def build(force: bool = False) -> str:
    """Compile the C oracle with gcc (oracle/Makefile)."""
    srcs = [os.path.join(_HERE, f) for f in ("sangnom_oracle.c", "sangnom_oracle_impl.inc", "sangnom_oracle.h")]
    stale = force or not os.path.exists(_LIB_PATH) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-s", "-C", _HERE, os.path.basename(_LIB_PATH)])
    return _LIB_PATH


class _Cfg(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in (
        "width", "height", "bytes", "bits", "planes", "subw", "subh",
        "order", "aa", "aac", "dh", "luma", "chroma")]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        L.sno_create.restype = ctypes.c_void_p
        L.sno_create.argtypes = [ctypes.POINTER(_Cfg)]
        L.sno_destroy.argtypes = [ctypes.c_void_p]
        L.sno_validate.argtypes = [ctypes.POINTER(_Cfg), ctypes.c_char_p, ctypes.c_size_t]
        L.sno_process.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int),
                                  ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int), ctypes.c_int]
        for f in ("sno_out_height", "sno_pool_stride", "sno_pool_rows"):
            getattr(L, f).argtypes = [ctypes.c_void_p]
        for f in ("sno_plane_width", "sno_plane_height_in", "sno_plane_height_out"):
            getattr(L, f).argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.sno_pool.restype = ctypes.c_void_p
        L.sno_pool.argtypes = [ctypes.c_void_p]
        L.sno_threshold.restype = ctypes.c_double
        L.sno_threshold.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.sno_plane.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                ctypes.c_int, ctypes.c_int]
        _lib = L
    return _lib


@dataclass
class Config:
    """Script-level arguments of SangNom2(...) plus the clip format (SangNom2.cpp:399-435)."""
    width: int
    height: int
    bytes: int = 1
    bits: int = 8
    planes: int = 1
    subw: int = 0
    subh: int = 0
    order: int = 1
    aa: int = 48
    aac: int = 0
    dh: bool = False
    luma: bool = True
    chroma: bool = True

    def c(self) -> _Cfg:
        return _Cfg(self.width, self.height, self.bytes, self.bits, self.planes, self.subw, self.subh,
                    self.order, self.aa, self.aac, int(self.dh), int(self.luma), int(self.chroma))

    @property
    def dtype(self):
        return {1: np.uint8, 2: np.uint16, 4: np.float32}[self.bytes]

    @property
    def out_height(self) -> int:
        return self.height * 2 if self.dh else self.height

    def plane_shape_in(self, p: int):
        return (self.height >> (self.subh if p else 0), self.width >> (self.subw if p else 0))

    def plane_shape_out(self, p: int):
        return (self.out_height >> (self.subh if p else 0), self.width >> (self.subw if p else 0))


def validate(cfg: Config) -> str:
    """'' if accepted, else the reference's error text (SangNom2.cpp:407-422)."""
    msg = ctypes.create_string_buffer(256)
    c = cfg.c()
    return msg.value.decode() if lib().sno_validate(ctypes.byref(c), msg, 256) else ""


class Oracle:
    """One filter instance: zeroed shared pool, frames in call order."""

    def __init__(self, cfg: Config):
        self.cfg = cfg
        c = cfg.c()
        self._h = lib().sno_create(ctypes.byref(c))
        if not self._h:
            raise ValueError("sno_create rejected the configuration")

    def close(self):
        if self._h:
            lib().sno_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def threshold(self, plane: int) -> float:
        return lib().sno_threshold(self._h, plane)

    def pool(self) -> np.ndarray:
        """View of the scratch pool as [9, bh + 1, stride_e]."""
        rows, stride = lib().sno_pool_rows(self._h), lib().sno_pool_stride(self._h)
        n = 9 * rows * stride
        buf = (ctypes.c_uint8 * (n * self.cfg.bytes)).from_address(lib().sno_pool(self._h))
        return np.frombuffer(buf, dtype=self.cfg.dtype).reshape(9, rows, stride)

    def process(self, src, parity: int = 1, dst=None):
        """src: list of 2-D arrays (one per plane, any row pitch).  Returns list of output planes.
        If dst is given its arrays are written in place (padding untouched)."""
        cfg = self.cfg
        n = min(cfg.planes, 3)
        assert len(src) >= n
        if dst is None:
            dst = [np.zeros(cfg.plane_shape_out(p), dtype=cfg.dtype) for p in range(n)]
        sp = (ctypes.c_void_p * 3)()
        dp = (ctypes.c_void_p * 3)()
        spitch = (ctypes.c_int * 3)()
        dpitch = (ctypes.c_int * 3)()
        for p in range(n):
            s, d = src[p], dst[p]
            assert s.dtype == cfg.dtype and d.dtype == cfg.dtype
            assert s.shape == cfg.plane_shape_in(p), (s.shape, cfg.plane_shape_in(p))
            assert d.shape == cfg.plane_shape_out(p)
            assert s.strides[1] == cfg.bytes and d.strides[1] == cfg.bytes
            sp[p], dp[p] = s.ctypes.data, d.ctypes.data
            spitch[p], dpitch[p] = s.strides[0], d.strides[0]
        rc = lib().sno_process(self._h, sp, spitch, dp, dpitch, int(parity))
        if rc:
            raise RuntimeError(f"sno_process failed: {rc}")
        return dst


# ---- the vectorisable 8-bit port (oracle/sangnom_vec.c): CPU timing baseline, bit-identical to the oracle ----
_VEC_PATH = os.path.join(_HERE, "libsangnom_vec.so")
_vec = None


def cpu_has_avx2() -> bool:
    try:
        with open("/proc/cpuinfo") as f:
            return " avx2" in f.read()
    except OSError:
        return False


def vec_lib():
    """libsangnom_vec.so, or None where it cannot run (no AVX2) or be built."""
    global _vec
    if _vec is None:
        if not cpu_has_avx2():
            return None
        src = os.path.join(_HERE, "sangnom_vec.c")
        if not os.path.exists(_VEC_PATH) or os.path.getmtime(src) > os.path.getmtime(_VEC_PATH):
            subprocess.check_call(["make", "-s", "-C", _HERE, "libsangnom_vec.so"])
        L = ctypes.CDLL(_VEC_PATH)
        L.snv_create_y8.restype = ctypes.c_void_p
        L.snv_create_y8.argtypes = [ctypes.c_int] * 4
        L.snv_destroy.argtypes = [ctypes.c_void_p]
        L.snv_pool.restype = ctypes.c_void_p
        L.snv_pool.argtypes = [ctypes.c_void_p]
        L.snv_process.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        _vec = L
    return _vec


class VecOracleY8:
    """One filter instance for an 8-bit Y clip (no dh), same conventions as Oracle."""

    def __init__(self, width: int, height: int, order: int = 1, aa: int = 48):
        self._lib = vec_lib()
        if self._lib is None:
            raise RuntimeError("libsangnom_vec.so needs AVX2")
        self.w, self.h = width, height
        self._h = self._lib.snv_create_y8(width, height, order, aa)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.snv_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def pool(self) -> np.ndarray:
        se, bh = (self.w + 31) // 32 * 32, (self.h + 1) >> 1
        buf = (ctypes.c_uint8 * (9 * (bh + 1) * se)).from_address(self._lib.snv_pool(self._h))
        return np.frombuffer(buf, dtype=np.uint8).reshape(9, bh + 1, se).copy()

    def process(self, src: np.ndarray, parity: int = 1, dst: np.ndarray | None = None) -> np.ndarray:
        if dst is None:
            dst = np.zeros_like(src)
        assert src.dtype == np.uint8 and src.shape == (self.h, self.w) and src.strides[1] == 1 and dst.strides[1] == 1
        self._lib.snv_process(self._h, src.ctypes.data, src.strides[0], dst.ctypes.data, dst.strides[0], int(parity))
        return dst
