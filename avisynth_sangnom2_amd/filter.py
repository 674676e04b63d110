"""Host-side mirror of the reference's filter object for tests and the benchmark.

`SangNom2(clip_format, order, aa, aac, threads, dh, luma, chroma, opt)` keeps the reference's
argument names, defaults and error text (/root/reference/src/SangNom2.cpp:399-435; README.md:20-57)
and `get_frame` plays the role of `SangNom2::GetFrame` (src/SangNom2.cpp:332-397).  All pixel work is
done by libsangnom_hip.so through the C ABI; torch is used only to hold device memory.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass

import numpy as np

from . import capi


class SangNomError(RuntimeError):
    """Raised where the reference would call env->ThrowError, with the same message."""

    def __init__(self, code: int, message: str):
        super().__init__(message)
        self.code = code


@dataclass
class ClipFormat:
    """The part of AviSynth's VideoInfo the filter looks at (planar Y / YUV only)."""
    width: int
    height: int
    bytes: int = 1       # vi.ComponentSize()
    bits: int = 8        # vi.BitsPerComponent()
    planes: int = 1      # vi.NumComponents() clipped to 3
    subw: int = 0        # log2 chroma subsampling
    subh: int = 0

    @property
    def dtype(self):
        return {1: np.uint8, 2: np.uint16, 4: np.float32}[self.bytes]


_FORMATS = {
    "Y8": dict(bytes=1, bits=8, planes=1), "Y10": dict(bytes=2, bits=10, planes=1),
    "Y12": dict(bytes=2, bits=12, planes=1), "Y14": dict(bytes=2, bits=14, planes=1),
    "Y16": dict(bytes=2, bits=16, planes=1), "Y32": dict(bytes=4, bits=32, planes=1),
    "YUV422P16": dict(bytes=2, bits=16, planes=3, subw=1, subh=0),
    "YUV420P8": dict(bytes=1, bits=8, planes=3, subw=1, subh=1),
    "YUV420P10": dict(bytes=2, bits=10, planes=3, subw=1, subh=1),
    "YUV420P16": dict(bytes=2, bits=16, planes=3, subw=1, subh=1),
    "YUV422P8": dict(bytes=1, bits=8, planes=3, subw=1, subh=0),
    "YUV444P8": dict(bytes=1, bits=8, planes=3), "YUV444P16": dict(bytes=2, bits=16, planes=3),
    "YUV444PS": dict(bytes=4, bits=32, planes=3), "YUV420PS": dict(bytes=4, bits=32, planes=3, subw=1, subh=1),
    "YUV422PS": dict(bytes=4, bits=32, planes=3, subw=1, subh=0),
}


def clip_format(name: str, width: int, height: int) -> ClipFormat:
    """ClipFormat from an AviSynth+ pixel_type name such as "YUV420P8" or "Y16"."""
    return ClipFormat(width=width, height=height, **_FORMATS[name])


class SangNom2:
    """One filter instance == one sn_context (its own stream and device pool)."""

    def __init__(self, clip: ClipFormat, order: int = 1, aa: int = 48, aac: int = 0, threads: int = 0,
                 dh: bool = False, luma: bool = True, chroma: bool = True, opt: int = -1,
                 device: int = 0, max_batch: int = 1, mode: str = "auto", stream: int | None = None,
                 host_depth: int = 0, isolated_planes: bool = False, fresh_pool: bool = False,
                 small_launches: int | None = None, chain: int | None = None, copy_threads: int | None = None,
                 scratch_budget_mb: int | None = None, chroma_sweeps: int | None = None):
        # `threads` is a dummy in the reference (README.md:40-41); `opt` picks its CPU code path.
        if opt < -1 or opt > 1:
            raise SangNomError(capi.SN_ERR_CONFIG, "SangNom2: opt must be between -1..2.")  # sic, SangNom2.cpp:420
        self.clip = clip
        self._lib = capi.load()
        cfg = capi.SnConfig(
            struct_size=ctypes.sizeof(capi.SnConfig), width=clip.width, height=clip.height,
            bytes_per_sample=clip.bytes, bits_per_sample=clip.bits, num_planes=clip.planes,
            sub_w=clip.subw, sub_h=clip.subh, order=order, aa=aa, aac=aac, dh=int(dh), luma=int(luma),
            chroma=int(chroma), device=device, max_batch=max_batch, mode=capi.MODES[mode], host_depth=host_depth, isolated_planes=int(isolated_planes), fresh_pool=int(fresh_pool),
            stream=stream)
        self._cfg = cfg
        self.max_batch = max_batch
        self.device = device
        self._h = ctypes.c_void_p()
        # scheduling only (sn_policy, sangnom_hip.h); None = capi.POLICY_DEFAULTS.  chain: 0 on, -1 off, 1 / 2 / 4 / 8 = on with
        # at most that many workgroups per cost buffer
        pol = capi.policy(small_launches=small_launches, chain=chain, copy_threads=copy_threads, scratch_budget_mb=scratch_budget_mb,
                          chroma_sweeps=chroma_sweeps)
        rc = self._lib.sn_create_with_policy(ctypes.byref(cfg), ctypes.byref(pol), ctypes.byref(self._h))
        if rc != capi.SN_OK:
            self._h = None
            raise SangNomError(rc, self._lib.sn_last_error(None).decode())
        self.out_height = clip.height * 2 if dh else clip.height

    # -- geometry -------------------------------------------------------------------------------
    @property
    def nplanes(self) -> int:
        return min(self.clip.planes, 3)

    def plane_shape_in(self, p: int):
        return (self.clip.height >> (self.clip.subh if p else 0), self.clip.width >> (self.clip.subw if p else 0))

    def plane_shape_out(self, p: int):
        return (self.out_height >> (self.clip.subh if p else 0), self.clip.width >> (self.clip.subw if p else 0))

    # -- life cycle -----------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._lib.sn_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc: int):
        if rc != capi.SN_OK:
            raise SangNomError(rc, self._lib.sn_last_error(self._h).decode())

    def info(self) -> capi.SnInfo:
        i = capi.SnInfo(struct_size=ctypes.sizeof(capi.SnInfo))
        self._check(self._lib.sn_get_info(self._h, ctypes.byref(i)))
        return i

    def stream_handle(self) -> int:
        return self._lib.sn_get_stream(self._h)

    def synchronize(self):
        self._check(self._lib.sn_synchronize(self._h))

    def read_pool(self, slot: int = 0) -> np.ndarray:
        i = self.info()
        out = np.empty((9, i.pool_rows, i.pool_stride), dtype=self.clip.dtype)
        self._check(self._lib.sn_debug_read_pool(self._h, slot, out.ctypes.data, out.nbytes))
        return out

    def set_policy(self, **fields) -> None:
        """Change the scheduling policy of the live context (sn_set_policy): small_launches, chain, copy_threads, chroma_sweeps."""
        cur = capi.SnPolicy(struct_size=ctypes.sizeof(capi.SnPolicy))
        self._check(self._lib.sn_get_policy(self._h, ctypes.byref(cur)))
        for k, v in fields.items():
            setattr(cur, k, int(v))
        self._check(self._lib.sn_set_policy(self._h, ctypes.byref(cur)))

    def get_policy(self) -> capi.SnPolicy:
        cur = capi.SnPolicy(struct_size=ctypes.sizeof(capi.SnPolicy))
        self._check(self._lib.sn_get_policy(self._h, ctypes.byref(cur)))
        return cur

    def raise_chain_fault(self) -> None:
        """Test hook: as if a chain over several workgroups had timed out (sn_debug_raise_chain_fault)."""
        self._check(self._lib.sn_debug_raise_chain_fault(self._h))

    def set_bands(self, bands: int = 0, warm_rows: int = 0) -> None:
        """Test hook: row bands of the small-launch path (sn_debug_set_bands)."""
        self._check(self._lib.sn_debug_set_bands(self._h, bands, warm_rows))

    def read_coupled_rows(self, which: int) -> np.ndarray:
        """Rows the fused 4:2:0 sweep of plane `which` left for the next plane: [9, rows, width]."""
        rows = self.info().coupled_rows
        out = np.empty((9, rows, self.clip.width), dtype=self.clip.dtype)
        self._check(self._lib.sn_debug_read_coupled_rows(self._h, which, out.ctypes.data, out.nbytes))
        return out

    # -- GetFrame -------------------------------------------------------------------------------
    def get_frame(self, src, parity: int = 1, dst=None):
        """Host planes (numpy 2-D arrays, any row pitch) in, host planes out (synchronous)."""
        n = self.nplanes
        if dst is None:
            dst = [np.zeros(self.plane_shape_out(p), dtype=self.clip.dtype) for p in range(n)]
        sp, dp = (ctypes.c_void_p * 3)(), (ctypes.c_void_p * 3)()
        spi, dpi = (ctypes.c_int32 * 3)(), (ctypes.c_int32 * 3)()
        for p in range(n):
            s, d = src[p], dst[p]
            if s.dtype != self.clip.dtype or d.dtype != self.clip.dtype:
                raise TypeError("plane dtype does not match the clip format")
            if s.shape != self.plane_shape_in(p) or d.shape != self.plane_shape_out(p):
                raise ValueError(f"plane {p}: shape {s.shape}->{d.shape}, expected "
                                 f"{self.plane_shape_in(p)}->{self.plane_shape_out(p)}")
            if s.strides[1] != self.clip.bytes or d.strides[1] != self.clip.bytes:
                raise ValueError("planes must be contiguous along x")
            sp[p], dp[p], spi[p], dpi[p] = s.ctypes.data, d.ctypes.data, s.strides[0], d.strides[0]
        self._check(self._lib.sn_process_host(self._h, sp, spi, dp, dpi, int(parity)))
        return dst

    # -- GetFrame with look-ahead: the host ring ------------------------------------------------------
    def host_slots(self) -> int:
        return self._lib.sn_host_slots(self._h)

    def _plane_args(self, planes, shape_of):
        ptr, pitch = (ctypes.c_void_p * 3)(), (ctypes.c_int32 * 3)()
        for p in range(self.nplanes):
            a = planes[p]
            if a.dtype != self.clip.dtype or a.shape != shape_of(p) or a.strides[1] != self.clip.bytes:
                raise ValueError(f"plane {p}: expected {shape_of(p)} {self.clip.dtype}, x-contiguous")
            ptr[p], pitch[p] = a.ctypes.data, a.strides[0]
        return ptr, pitch

    def submit(self, src, parity: int = 1, dst=None) -> int:
        """Queue one host frame (H2D, sweeps, D2H on a ring slot's own stream); returns the slot.  With `dst` the
        output planes are named now (sn_submit_host_to): pinned ones are written straight from the device."""
        sp, spi = self._plane_args(src, self.plane_shape_in)
        slot = ctypes.c_int32(-1)
        if dst is None:
            self._check(self._lib.sn_submit_host(self._h, sp, spi, int(parity), ctypes.byref(slot)))
        else:
            dp, dpi = self._plane_args(dst, self.plane_shape_out)
            self._check(self._lib.sn_submit_host_to(self._h, sp, spi, dp, dpi, int(parity), ctypes.byref(slot)))
        return slot.value

    def collect(self, slot: int, dst=None, announced: bool = False):
        """Wait for the frame in `slot` and copy it into host planes (announced: they were named at submission)."""
        if announced:
            self._check(self._lib.sn_collect_host(self._h, int(slot), None, None))
            return dst
        if dst is None:
            dst = [np.zeros(self.plane_shape_out(p), dtype=self.clip.dtype) for p in range(self.nplanes)]
        dp, dpi = self._plane_args(dst, self.plane_shape_out)
        self._check(self._lib.sn_collect_host(self._h, int(slot), dp, dpi))
        return dst

    def turn(self, src, dst, direction: int):
        """TurnRight (direction > 0) / TurnLeft (< 0) of device planes: src [N, H, W] -> dst [N, W, H] (torch tensors)."""
        B = self.clip.bytes
        N, H, W = src.shape
        if tuple(dst.shape) != (N, W, H) or src.stride(2) != 1 or dst.stride(2) != 1 or src.element_size() != B or dst.element_size() != B:
            raise ValueError("turn: dst must be [N, W, H] of the clip's sample type, both x-contiguous")
        self._check(self._lib.sn_turn_device(self._h, int(direction), N, src.data_ptr(), src.stride(0) * B, src.stride(1) * B, W, H,
                                             dst.data_ptr(), dst.stride(0) * B, dst.stride(1) * B))
        return dst

    def get_frame_device(self, src, dst, parity: int = 1):
        """Device planes: torch tensors [H, W] on this context's GPU.  Asynchronous."""
        return self.process_batch([s.unsqueeze(0) for s in src], [d.unsqueeze(0) for d in dst], [parity])

    def process_batch(self, src, dst, parity=None):
        """src[p] / dst[p]: torch tensors [N, H_p, W_p] (device-resident, x-contiguous).
        Frame f of plane p lives at tensor[f].  Asynchronous on the context's stream."""
        n = self.nplanes
        N = src[0].shape[0]
        sp, dp = (ctypes.c_void_p * 3)(), (ctypes.c_void_p * 3)()
        spi, dpi = (ctypes.c_int32 * 3)(), (ctypes.c_int32 * 3)()
        sfs, dfs = (ctypes.c_int64 * 3)(), (ctypes.c_int64 * 3)()
        B = self.clip.bytes
        for p in range(n):
            s, d = src[p], dst[p]
            if tuple(s.shape[1:]) != self.plane_shape_in(p) or tuple(d.shape[1:]) != self.plane_shape_out(p):
                raise ValueError(f"plane {p}: bad shape {tuple(s.shape)} -> {tuple(d.shape)}")
            if s.shape[0] != N or d.shape[0] != N:
                raise ValueError("all planes must carry the same number of frames")
            if s.stride(2) != 1 or d.stride(2) != 1 or s.element_size() != B or d.element_size() != B:
                raise ValueError("planes must be x-contiguous tensors of the clip's sample type")
            if not s.is_cuda or not d.is_cuda:
                raise ValueError("process_batch needs device-resident tensors (use get_frame for host planes)")
            sp[p], dp[p] = s.data_ptr(), d.data_ptr()
            spi[p], dpi[p] = s.stride(1) * B, d.stride(1) * B
            sfs[p], dfs[p] = s.stride(0) * B, d.stride(0) * B
        par = None
        if parity is not None:
            par = (ctypes.c_int32 * N)(*[int(x) for x in parity])
        self._check(self._lib.sn_process_device_strided(self._h, N, sp, sfs, spi, dp, dfs, dpi, par))
        return dst


def pin_host_array(a: np.ndarray) -> None:
    """sn_pin_host_buffer on a numpy array's memory: frames inside it then move over PCIe without staging copies.
    Keep the array alive and call unpin_host_array before dropping it."""
    rc = capi.load().sn_pin_host_buffer(a.ctypes.data, a.nbytes)
    if rc != capi.SN_OK:
        raise SangNomError(rc, capi.load().sn_last_error(None).decode())


def unpin_host_array(a: np.ndarray) -> None:
    rc = capi.load().sn_unpin_host_buffer(a.ctypes.data)
    if rc != capi.SN_OK:
        raise SangNomError(rc, capi.load().sn_last_error(None).decode())


def SangNom(clip: ClipFormat, order: int = 1, aa: int = 48, opt: int = -1, **kw) -> SangNom2:
    """Legacy entry point (src/SangNom2.cpp:437-472): order 0/1/2 means bottom/top/double-rate and
    is remapped to SangNom2's 2/1/0.  The reference reads args[3] (`opt`) as aac because of an
    out-of-range argument read; that quirk is NOT reproduced: aac is 0 here."""
    if order < 0 or order > 2:
        raise SangNomError(capi.SN_ERR_CONFIG, "SangNom: order must be between 0..2.")
    return SangNom2(clip, order=(2, 1, 0)[order], aa=aa, aac=0, opt=opt, **kw)


class SangNomAA:
    """The anti-aliasing idiom TurnLeft().SangNom2(...).TurnRight().SangNom2(...) with the frames kept on the
    device between the two passes: two filter instances -- one for the turned clip, one for the
    clip itself -- on one stream, and the library's turn kernel in between.  Not a function of the reference; the
    result is what that script gives with it."""

    def __init__(self, clip: ClipFormat, max_batch: int = 1, device: int = 0, **kw):
        self.clip = clip
        turned = ClipFormat(width=clip.height, height=clip.width, bytes=clip.bytes, bits=clip.bits, planes=clip.planes,
                            subw=clip.subh, subh=clip.subw)
        self.first = SangNom2(turned, max_batch=max_batch, device=device, **kw)
        self.second = SangNom2(clip, max_batch=max_batch, device=device, stream=self.first.stream_handle(), **kw)
        self._tmp = None

    def close(self):
        self.second.close()
        self.first.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def process_batch(self, src, dst, parity=None):
        """src[p], dst[p]: torch tensors [N, H_p, W_p] on the device.  Asynchronous on the instances' stream."""
        import torch
        n = self.first.nplanes
        N = src[0].shape[0]
        if self._tmp is None or self._tmp[0][0].shape[0] != N:
            mk = lambda shape, like: torch.empty((N,) + tuple(shape), dtype=like.dtype, device=like.device)
            self._tmp = ([mk(self.first.plane_shape_in(p), src[p]) for p in range(n)],
                         [mk(self.first.plane_shape_out(p), src[p]) for p in range(n)],
                         [mk(self.second.plane_shape_in(p), src[p]) for p in range(n)])
            # the buffers were allocated on torch's current stream; the library runs on its own
            torch.cuda.current_stream().synchronize()
        t1, u1, t2 = self._tmp
        for p in range(n):
            self.first.turn(src[p], t1[p], -1)
        self.first.process_batch(t1, u1, parity)
        for p in range(n):
            self.first.turn(u1[p], t2[p], +1)
        return self.second.process_batch(t2, dst, parity)

    def synchronize(self):
        self.second.synchronize()


class SangNomAAHost:
    """The same idiom through the C ABI's one-call entry point (sn_aa_create / sn_aa_process_host): host planes in,
    host planes out, the frame stays on the device between the two passes.  This is what the plugin function
    SangNomAA (host/sangnom2_avs_plugin.cpp) binds."""

    def __init__(self, clip: ClipFormat, order: int = 1, aa: int = 48, aac: int = 0, luma: bool = True, chroma: bool = True,
                 device: int = 0, isolated_planes: bool = False, fresh_pool: bool = False, **policy_kw):
        self.clip = clip
        self._lib = capi.load()
        cfg = capi.SnConfig(
            struct_size=ctypes.sizeof(capi.SnConfig), width=clip.width, height=clip.height, bytes_per_sample=clip.bytes,
            bits_per_sample=clip.bits, num_planes=clip.planes, sub_w=clip.subw, sub_h=clip.subh, order=order, aa=aa, aac=aac,
            dh=0, luma=int(luma), chroma=int(chroma), device=device, max_batch=1, mode=capi.SN_MODE_AUTO, host_depth=0,
            isolated_planes=int(isolated_planes), fresh_pool=int(fresh_pool), stream=None)
        self._h = ctypes.c_void_p()
        pol = capi.policy(**{k: policy_kw.get(k) for k in ("small_launches", "chain", "copy_threads", "scratch_budget_mb")})
        rc = self._lib.sn_aa_create_with_policy(ctypes.byref(cfg), ctypes.byref(pol), ctypes.byref(self._h))
        if rc != capi.SN_OK:
            self._h = None
            raise SangNomError(rc, self._lib.sn_aa_last_error(None).decode())

    def close(self):
        if getattr(self, "_h", None):
            self._lib.sn_aa_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def get_frame(self, src, parity: int = 1):
        n = min(self.clip.planes, 3)
        dst = [np.zeros_like(src[p]) for p in range(n)]
        sp, dp = (ctypes.c_void_p * 3)(), (ctypes.c_void_p * 3)()
        spi, dpi = (ctypes.c_int32 * 3)(), (ctypes.c_int32 * 3)()
        for p in range(n):
            if src[p].dtype != self.clip.dtype or src[p].strides[1] != self.clip.bytes:
                raise ValueError("planes must be x-contiguous arrays of the clip's sample type")
            sp[p], dp[p], spi[p], dpi[p] = src[p].ctypes.data, dst[p].ctypes.data, src[p].strides[0], dst[p].strides[0]
        rc = self._lib.sn_aa_process_host(self._h, sp, spi, dp, dpi, int(parity))
        if rc != capi.SN_OK:
            raise SangNomError(rc, self._lib.sn_aa_last_error(self._h).decode())
        return dst
