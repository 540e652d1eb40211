"""oracle/oracle.py — ctypes front-ends for the two CPU checkers.  TEST INFRASTRUCTURE.

  CpuRef  : oracle/_build/libcpuref.so   our C restatement (any q < 2^62, batched)
  CTRef   : oracle/_ref/libctensor.so    the reference's own lol-cpp C++ (q < ~2^31.5,
                                         one polynomial per call, NOT thread-safe:
                                         global Zq::q, types.h:59)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from . import lolmath as lm

_HERE = os.path.dirname(os.path.abspath(__file__))
CPUREF_SO = os.path.join(_HERE, "_build", "libcpuref.so")
CTREF_SO = os.path.join(_HERE, "_ref", "libctensor.so")

_i64p = C.POINTER(C.c_int64)


class _PP(C.Structure):
    _fields_ = [("prime", C.c_int16), ("exponent", C.c_int16)]


def build(ref: bool = True) -> None:
    """Compile the restatement (always) and the reference library (when
    /root/reference is present).  Building the checker is not using it."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "cpuref"] + (["ref"] if ref else []))


def _pp_array(pps):
    arr = (_PP * max(1, len(pps)))()
    for i, (p, e) in enumerate(pps):
        arr[i].prime, arr[i].exponent = p, e
    return arr


def _ptr(a: np.ndarray):
    assert a.dtype == np.int64 and a.flags.c_contiguous
    return a.ctypes.data_as(_i64p)


def _ru_ptrs(tabs):
    keep = [np.ascontiguousarray(np.array(t, dtype=np.int64)) for t in tabs]
    arr = (_i64p * max(1, len(keep)))(*[_ptr(k) for k in keep])
    return arr, keep


class Params:
    """Everything a transform needs for (pps, qs): twiddles by the Lol rule."""

    def __init__(self, pps, qs):
        self.pps = [tuple(pe) for pe in pps]
        self.qs = [int(q) for q in qs]
        self.T = len(self.qs)
        self.m = lm.value_pps(self.pps)
        self.n = lm.totient_pps(self.pps)
        self.ru = lm.ru_tables(self.pps, self.qs)
        self.ruinv = lm.ru_tables(self.pps, self.qs, inverse=True)
        self.mhatinv = [lm.mhat_inv(self.m, q) for q in self.qs]

    def random(self, rng, B=1):
        cols = [rng.integers(0, q, size=(B, self.n), dtype=np.int64) for q in self.qs]
        return np.ascontiguousarray(np.stack(cols, axis=-1))  # [B, n, T]


class CpuRef:
    def __init__(self, path: str = CPUREF_SO):
        if not os.path.exists(path):
            build(ref=False)
        self.lib = C.CDLL(path)
        L = self.lib
        pp, ipp, i64 = C.POINTER(_PP), C.POINTER(_i64p), C.c_int64
        L.ref_crt.argtypes = [C.c_int, _i64p, i64, i64, pp, C.c_int, ipp, _i64p]
        L.ref_crtinv.argtypes = [C.c_int, _i64p, i64, i64, pp, C.c_int, ipp, _i64p, _i64p]
        L.ref_mul.argtypes = [C.c_int, _i64p, _i64p, i64, i64, _i64p]
        for nm in ("ref_l", "ref_linv", "ref_gpow", "ref_gdec", "ref_ginvpow", "ref_ginvdec"):
            getattr(L, nm).argtypes = [C.c_int, _i64p, i64, i64, pp, C.c_int, _i64p]
        L.ref_ginvpow.restype = C.c_int
        L.ref_ginvdec.restype = C.c_int
        L.ref_crt_naive.argtypes = [_i64p, _i64p, i64, pp, C.c_int, _i64p, i64]
        L.ref_polymul.argtypes = [C.c_int, _i64p, _i64p, _i64p, i64, i64, pp, C.c_int, ipp, ipp, _i64p, _i64p]

    @staticmethod
    def _shape(P: Params, y):
        y = np.ascontiguousarray(y, dtype=np.int64)
        assert y.size % (P.n * P.T) == 0
        return y.copy(), y.size // (P.n * P.T)

    def crt(self, P: Params, y):
        y, B = self._shape(P, y)
        ru, keep = _ru_ptrs(P.ru)
        qs = np.array(P.qs, dtype=np.int64)
        self.lib.ref_crt(P.T, _ptr(y), B, P.n, _pp_array(P.pps), len(P.pps), ru, _ptr(qs))
        return y

    def crtinv(self, P: Params, y):
        y, B = self._shape(P, y)
        ru, keep = _ru_ptrs(P.ruinv)
        qs = np.array(P.qs, dtype=np.int64)
        mh = np.array(P.mhatinv, dtype=np.int64)
        self.lib.ref_crtinv(P.T, _ptr(y), B, P.n, _pp_array(P.pps), len(P.pps), ru, _ptr(mh), _ptr(qs))
        return y

    def mul(self, P: Params, a, b):
        a, B = self._shape(P, a)
        b = np.ascontiguousarray(b, dtype=np.int64)
        qs = np.array(P.qs, dtype=np.int64)
        self.lib.ref_mul(P.T, _ptr(a), _ptr(b), B, P.n, _ptr(qs))
        return a

    def _prime(self, name, P: Params, y):
        y, B = self._shape(P, y)
        qs = np.array(P.qs, dtype=np.int64)
        ret = getattr(self.lib, name)(P.T, _ptr(y), B, P.n, _pp_array(P.pps), len(P.pps), _ptr(qs))
        return y, ret

    def l(self, P, y): return self._prime("ref_l", P, y)[0]
    def linv(self, P, y): return self._prime("ref_linv", P, y)[0]
    def gpow(self, P, y): return self._prime("ref_gpow", P, y)[0]
    def gdec(self, P, y): return self._prime("ref_gdec", P, y)[0]

    def ginvpow(self, P, y):
        y, ok = self._prime("ref_ginvpow", P, y)
        return y if ok else None

    def ginvdec(self, P, y):
        y, ok = self._prime("ref_ginvdec", P, y)
        return y if ok else None

    def crt_naive(self, P: Params, y, t=0):
        """Closed-form definition, one component of one polynomial."""
        y = np.ascontiguousarray(y, dtype=np.int64).reshape(P.n)
        out = np.zeros_like(y)
        om = np.array([P.ru[k][1 * P.T + t] for k in range(len(P.pps))], dtype=np.int64)
        self.lib.ref_crt_naive(_ptr(out), _ptr(y), P.n, _pp_array(P.pps), len(P.pps), _ptr(om), P.qs[t])
        return out

    def polymul(self, P: Params, a, b):
        a, B = self._shape(P, a)
        b = np.ascontiguousarray(b, dtype=np.int64)
        c = np.empty_like(a)
        ru, k1 = _ru_ptrs(P.ru)
        rui, k2 = _ru_ptrs(P.ruinv)
        qs = np.array(P.qs, dtype=np.int64)
        mh = np.array(P.mhatinv, dtype=np.int64)
        self.lib.ref_polymul(P.T, _ptr(c), _ptr(a), _ptr(b), B, P.n, _pp_array(P.pps), len(P.pps),
                             ru, rui, _ptr(mh), _ptr(qs))
        return c

    # twace/embed have no C reference (Haskell only): pure-Python restatement
    def _per_comp(self, P_in: Params, n_out, y, f):
        y = np.ascontiguousarray(y, dtype=np.int64).reshape(-1, P_in.n, P_in.T)
        out = np.zeros((y.shape[0], n_out, P_in.T), dtype=np.int64)
        for b in range(y.shape[0]):
            for t, q in enumerate(P_in.qs):
                out[b, :, t] = f([int(v) for v in y[b, :, t]], q)
        return out

    def embed_pow(self, P, P2, y):
        return self._per_comp(P, P2.n, y, lambda a, q: lm.embed_pow(P.pps, P2.pps, a))

    def embed_dec(self, P, P2, y):
        return self._per_comp(P, P2.n, y, lambda a, q: lm.embed_dec(P.pps, P2.pps, a, q))

    def embed_crt(self, P, P2, y):
        return self._per_comp(P, P2.n, y, lambda a, q: lm.embed_crt(P.pps, P2.pps, a))

    def twace_powdec(self, P, P2, y):
        return self._per_comp(P2, P.n, y, lambda a, q: lm.twace_powdec(P.pps, P2.pps, a))

    def twace_crt(self, P, P2, y):
        return self._per_comp(P2, P.n, y, lambda a, q: lm.twace_crt(P.pps, P2.pps, a, q))

    def coeffs(self, P, P2, y):
        """class Tensor `coeffs` (Tensor.hs:174; CPP/Extension.hs:90-93): [n'/n][B][n][T]."""
        idx = np.array(lm.ext_indices_coeffs(P.pps, P2.pps), dtype=np.int64)       # [rel][n]
        y = np.ascontiguousarray(y, dtype=np.int64).reshape(-1, P2.n, P2.T)
        return np.ascontiguousarray(np.moveaxis(y[:, idx, :], 1, 0))                # y[b, idx[i1,i0], t]


class CTRef:
    """The reference's own C++ (lol-cpp CT), one polynomial per call.
    Correct only for q < ~2^31.5 (types.h:79-84)."""

    def __init__(self, path: str = CTREF_SO):
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        self.lib = C.CDLL(path)
        L = self.lib
        pp, ipp, i16 = C.POINTER(_PP), C.POINTER(_i64p), C.c_int16
        hdim = C.c_int32  # hDim_t (types.h:22)
        L.tensorCRTRq.argtypes = [i16, _i64p, hdim, pp, i16, ipp, _i64p]
        L.tensorCRTInvRq.argtypes = [i16, _i64p, hdim, pp, i16, ipp, _i64p, _i64p]
        L.mulRq.argtypes = [i16, _i64p, _i64p, hdim, _i64p]
        for nm in ("tensorLRq", "tensorLInvRq", "tensorGPowRq", "tensorGDecRq",
                   "tensorGInvPowRq", "tensorGInvDecRq"):
            getattr(L, nm).argtypes = [i16, _i64p, hdim, pp, i16, _i64p]
        L.tensorGInvPowRq.restype = i16
        L.tensorGInvDecRq.restype = i16
        # floating-point members (crt.cpp:583-598, random.cpp:61-64); Complex = {double re, im}
        cp, cpp_ = C.c_void_p, C.POINTER(C.c_void_p)
        L.tensorCRTC.argtypes = [i16, cp, hdim, pp, i16, cpp_]
        L.tensorCRTInvC.argtypes = [i16, cp, hdim, pp, i16, cpp_, cp]
        L.tensorGaussianDec.argtypes = [i16, cp, hdim, pp, i16, cpp_]
        for nm in ("tensorCRTC", "tensorCRTInvC", "tensorGaussianDec"):
            getattr(L, nm).restype = None

    # ---- floating-point members: the reference's own code on complex roots omega = exp(2 pi i / m) ----
    @staticmethod
    def _cplx_ru(pps, inverse=False):
        """ru[k][i] = omega_{pp_k}^{+-i} as complex128 (CPP.hs:422-442 at Complex Double)."""
        keep = []
        for p, e in pps:
            pp = p ** e
            ang = 2.0 * np.pi * np.arange(pp) / pp
            keep.append(np.ascontiguousarray(np.cos(ang) + (-1j if inverse else 1j) * np.sin(ang), dtype=np.complex128))
        arr = (C.c_void_p * max(1, len(keep)))(*[k.ctypes.data for k in keep])
        return arr, keep

    def crtc(self, pps, y):
        n = lm.totient_pps(pps)
        y = np.ascontiguousarray(y, dtype=np.complex128).copy().reshape(-1, n)
        ru, keep = self._cplx_ru(pps)
        pa = _pp_array(pps)
        for b in range(y.shape[0]):
            self.lib.tensorCRTC(1, y[b].ctypes.data, n, pa, len(pps), ru)
        return y

    def crtinvc(self, pps, y):
        n = lm.totient_pps(pps)
        m = lm.value_pps(pps)
        y = np.ascontiguousarray(y, dtype=np.complex128).copy().reshape(-1, n)
        ru, keep = self._cplx_ru(pps, inverse=True)
        mh = np.array([1.0 / (m // 2 if m % 2 == 0 else m)], dtype=np.complex128)
        pa = _pp_array(pps)
        for b in range(y.shape[0]):
            self.lib.tensorCRTInvC(1, y[b].ctypes.data, n, pa, len(pps), ru, mh.ctypes.data)
        return y

    def gaussian_dec(self, pps, y):
        n = lm.totient_pps(pps)
        y = np.ascontiguousarray(y, dtype=np.float64).copy().reshape(-1, n)
        ru, keep = self._cplx_ru(pps)
        pa = _pp_array(pps)
        for b in range(y.shape[0]):
            self.lib.tensorGaussianDec(1, y[b].ctypes.data, n, pa, len(pps), ru)
        return y

    def _each(self, P: Params, y, fn):
        y = np.ascontiguousarray(y, dtype=np.int64).copy().reshape(-1, P.n, P.T)
        for b in range(y.shape[0]):
            fn(y[b])
        return y

    def crt(self, P, y):
        ru, keep = _ru_ptrs(P.ru)
        qs = np.array(P.qs, dtype=np.int64)
        pa = _pp_array(P.pps)
        return self._each(P, y, lambda v: self.lib.tensorCRTRq(P.T, _ptr(v), P.n, pa, len(P.pps), ru, _ptr(qs)))

    def crtinv(self, P, y):
        ru, keep = _ru_ptrs(P.ruinv)
        qs = np.array(P.qs, dtype=np.int64)
        mh = np.array(P.mhatinv, dtype=np.int64)
        pa = _pp_array(P.pps)
        return self._each(P, y, lambda v: self.lib.tensorCRTInvRq(P.T, _ptr(v), P.n, pa, len(P.pps), ru, _ptr(mh), _ptr(qs)))

    def mul(self, P, a, b):
        a = np.ascontiguousarray(a, dtype=np.int64).copy().reshape(-1, P.n, P.T)
        b = np.ascontiguousarray(b, dtype=np.int64).reshape(-1, P.n, P.T)
        qs = np.array(P.qs, dtype=np.int64)
        for i in range(a.shape[0]):
            self.lib.mulRq(P.T, _ptr(a[i]), _ptr(b[i]), P.n, _ptr(qs))
        return a

    def _prime(self, name, P, y):
        qs = np.array(P.qs, dtype=np.int64)
        pa = _pp_array(P.pps)
        rets = []
        out = self._each(P, y, lambda v: rets.append(getattr(self.lib, name)(P.T, _ptr(v), P.n, pa, len(P.pps), _ptr(qs))))
        return out, rets

    def l(self, P, y): return self._prime("tensorLRq", P, y)[0]
    def linv(self, P, y): return self._prime("tensorLInvRq", P, y)[0]
    def gpow(self, P, y): return self._prime("tensorGPowRq", P, y)[0]
    def gdec(self, P, y): return self._prime("tensorGDecRq", P, y)[0]

    def ginvpow(self, P, y):
        out, rets = self._prime("tensorGInvPowRq", P, y)
        return out if all(rets) else None

    def ginvdec(self, P, y):
        out, rets = self._prime("tensorGInvDecRq", P, y)
        return out if all(rets) else None

    def polymul(self, P, a, b):
        return self.crtinv(P, self.mul(P, self.crt(P, a), self.crt(P, b)))
