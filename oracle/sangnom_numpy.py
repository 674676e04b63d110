"""Independent numpy restatement of the SangNom2 opt=0 path.  TEST INFRASTRUCTURE ONLY.

Written from SURVEY.md Appendix A and /root/reference/src/SangNom2.cpp:25-397 in a deliberately
different style from oracle/sangnom_oracle.c (whole-row vector operations on edge-padded arrays,
explicit modular arithmetic instead of C narrowing) so that the two restatements cross-check each
other.  PARITY UNPINNED, like the C oracle: the reference has no fixtures and cannot be built here.

Not used by the product; imported only by tests/.
"""
from __future__ import annotations

import numpy as np

F32 = np.float32


class NumpySangNom:
    """One filter instance with the reference's shared, luma-sized scratch pool (zero-filled)."""

    def __init__(self, width, height, bytes=1, bits=8, planes=1, subw=0, subh=0,
                 order=1, aa=48, aac=0, dh=False, luma=True, chroma=True):
        self.w, self.h_in = width, height
        self.bytes, self.bits, self.planes = bytes, bits, planes
        self.subw, self.subh = subw, subh
        self.order, self.dh = order, dh
        self.process = [luma, chroma, chroma]
        self.is_float = bytes == 4
        self.dtype = {1: np.uint8, 2: np.uint16, 4: np.float32}[bytes]
        self.M = 1 << (8 * bytes)                      # container modulus (Appendix A)
        self.h_out = height * 2 if dh else height      # SangNom2.cpp:284-285
        self.stride_e = (width + 31) // 32 * 32        # SangNom2.cpp:287
        self.bh = (self.h_out + 1) >> 1                # SangNom2.cpp:288
        # thresholds, SangNom2.cpp:280-282 (float32 arithmetic, then truncation to T for integers)
        self.thr = []
        for a in (aa, aac, aac):
            if self.is_float:
                self.thr.append(F32(F32(F32(a) * F32(21.0) / F32(16.0)) / F32(256.0)))
            else:
                v = F32(F32(a) * F32(21.0) / F32(16.0)) * F32(1 << (bits - 8))
                self.thr.append(int(v) % self.M)
        wt = np.float32 if self.is_float else np.int64
        self.pool = np.zeros((9, self.bh + 1, self.stride_e), dtype=wt)

    # ---- helpers -------------------------------------------------------------------------
    def _taps(self, row):
        """row[clamp(x+k, 0, w-1)] for k=-3..3 as a dict k -> array."""
        w = row.shape[0]
        p = np.pad(row, 3, mode="edge")
        return {k: p[3 + k:3 + k + w] for k in range(-3, 4)}

    def _sg(self, p1, p2, p3):
        if self.is_float:
            s = (p1 * F32(4) + p2 * F32(5)) - p3      # one rounding per operation
            return s * F32(0.125)
        return ((4 * p1 + 5 * p2 - p3) >> 3) % self.M  # arithmetic shift, then wrap into T

    def _avg(self, a, b):
        if self.is_float:
            return (a + b) * F32(0.5)
        return (a + b + 1) >> 1

    def _candidates(self, c, n):
        tc, tn = self._taps(c), self._taps(n)
        f1 = self._sg(tc[-1], tc[0], tc[1])
        f2 = self._sg(tn[1], tn[0], tn[-1])
        b1 = self._sg(tc[1], tc[0], tc[-1])
        b2 = self._sg(tn[-1], tn[0], tn[1])
        return tc, tn, f1, f2, b1, b2

    # ---- the three stages on one plane -----------------------------------------------------
    def _plane(self, dst, offset, plane):
        h, w = dst.shape
        wt = np.float32 if self.is_float else np.int64
        K = dst[offset::2].astype(wt)                  # kept field, h/2 lines
        nr = h // 2 - 1
        P = self.pool
        # stage 1, SangNom2.cpp:74-124
        for y in range(nr):
            tc, tn, f1, f2, b1, b2 = self._candidates(K[y], K[y + 1])
            d = [tc[-3] - tn[3], tc[-2] - tn[2], tc[-1] - tn[1], f1 - f2, tc[0] - tn[0],
                 b1 - b2, tc[1] - tn[-1], tc[2] - tn[-2], tc[3] - tn[-3]]
            for b in range(9):
                P[b, y + 1, :w] = np.abs(d[b])
        # stage 2, SangNom2.cpp:126-159: in place, top to bottom, whole pool stride
        se = self.stride_e
        for b in range(9):
            for r in range(1, self.bh):
                S = (P[b, r - 1] + P[b, r]) + P[b, r + 1]
                Sp = np.pad(S, 3, mode="edge")
                acc = Sp[0:se] + Sp[1:se + 1]
                for k in range(2, 7):
                    acc = acc + Sp[k:se + k]
                if self.is_float:
                    P[b, r] = acc / F32(16)
                else:
                    P[b, r] = (acc // 16) % self.M
        # stage 3, SangNom2.cpp:161-257
        thr = self.thr[plane]
        out_rows = []
        for y in range(nr):
            tc, tn, f1, f2, b1, b2 = self._candidates(K[y], K[y + 1])
            v = P[:, y + 1, :w]
            m = v.min(axis=0)
            cands = [  # (buffer index, value) in the reference's priority order
                (5, self._avg(b1, b2)), (3, self._avg(f1, f2)),
                (6, self._avg(tc[1], tn[-1])), (2, self._avg(tc[-1], tn[1])),
                (7, self._avg(tc[2], tn[-2])), (1, self._avg(tc[-2], tn[2])),
                (8, self._avg(tc[3], tn[-3])), (0, self._avg(tc[-3], tn[3])),
            ]
            res = cands[-1][1].copy()
            for bidx, val in reversed(cands[:-1]):
                res = np.where(v[bidx] == m, val, res)
            res = np.where((v[4] == m) | (m > thr), self._avg(tc[0], tn[0]), res)
            out_rows.append(res)
        for y in range(nr):
            dst[offset + 2 * y + 1] = out_rows[y].astype(self.dtype)

    # ---- GetFrame, SangNom2.cpp:332-397 ------------------------------------------------------
    def get_frame(self, src, parity=1):
        offset = {0: 0 if parity else 1, 1: 0, 2: 1}[self.order]
        out = []
        for i in range(min(self.planes, 3)):
            s = src[i]
            h_out = self.h_out >> (self.subh if i else 0)
            w = self.w >> (self.subw if i else 0)
            assert s.shape == (self.h_in >> (self.subh if i else 0), w)
            d = np.zeros((h_out, w), dtype=self.dtype)
            if self.dh:
                d[offset::2] = s
            elif not self.process[i]:
                out.append(s.copy())
                continue
            else:
                d[offset::2] = s[offset::2]
            if offset == 0:
                d[h_out - 1] = d[h_out - 2]
            else:
                d[0] = d[1]
            self._plane(d, offset, i)
            out.append(d)
        return out
