"""lol_amd/tensor.py — ctypes binding of include/lolhip.h, shaped like Lol's `Tensor` class.

Method names follow the reference's class methods (lol/Crypto/Lol/Cyclotomic/Tensor.hs:86-193):
crt, crtInv, l, lInv, mulGPow, mulGDec, divGPow, divGDec, mulGCRT, divGCRT,
twacePowDec, twaceCRT, embedPow, embedDec, embedCRT, plus zipWithT (*) as `mul` and
the fused ring product `polymul` (Cyc (*), lol/Crypto/Lol/Cyclotomic/Cyc.hs:262-297).

Arrays are int64 with shape [..., n, T] (T = number of RNS moduli; a trailing axis of
length 1 for a single modulus) — the reference's AoS layout (tensor.h:69).
`divG*` return None where the reference returns Nothing (CPP.hs:309-323).

Two calling styles per operation:
  * numpy arrays  -> host round trip through `lolhip_op_host`
  * torch CUDA int64 tensors (or raw device pointers via *_dev) -> in place in HBM
Neither has a CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_i64p = C.POINTER(C.c_int64)
_i32p = C.POINTER(C.c_int32)

OK, ERR_INVALID, ERR_MODULUS, ERR_NO_CRT, ERR_ROOT, ERR_NO_DEVICE, ERR_HIP, ERR_NOT_DIVISIBLE, ERR_DEVICE = 0, -1, -2, -3, -4, -5, -6, -7, -8
_ERRNAMES = {ERR_INVALID: "invalid argument", ERR_MODULUS: "modulus out of range / missing inverse",
             ERR_NO_CRT: "no CRT basis for this modulus", ERR_ROOT: "bad root of unity",
             ERR_NO_DEVICE: "no HIP device (liblolhip has no CPU fallback)", ERR_HIP: "HIP runtime error",
             ERR_NOT_DIVISIBLE: "not divisible by g",
             ERR_DEVICE: "the current HIP device is not the plan's device"}

OP_CRT, OP_CRTINV, OP_MUL, OP_POLYMUL, OP_L, OP_LINV, OP_MULGPOW, OP_MULGDEC, OP_DIVGPOW, OP_DIVGDEC, OP_MULGCRT, OP_DIVGCRT = range(12)
EXT_TWACE_POWDEC, EXT_TWACE_CRT, EXT_EMBED_POW, EXT_EMBED_DEC, EXT_EMBED_CRT, EXT_COEFFS = range(6)


class LolHipError(RuntimeError):
    def __init__(self, code, what=""):
        self.code = code
        super().__init__(f"lolhip: {_ERRNAMES.get(code, code)} ({code}) {what}")


class NoDeviceError(LolHipError):
    pass


class _PP(C.Structure):
    _fields_ = [("prime", C.c_int16), ("exponent", C.c_int16)]


def lib_path() -> str:
    return os.path.join(_HERE, "liblolhip.so")


_lib = None


def lib():
    """Load liblolhip.so (built in-tree by __graft_entry__.build()).  Fails loudly."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm wheels carry their own libamdhip64; if torch is going to be used in this
    # process it must be the one that loads the HIP runtime first, or its later device probe
    # finds the GPU already claimed by a second runtime copy ("No HIP GPUs are available").
    if os.environ.get("LOLHIP_NO_TORCH_PRELOAD") is None:
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    path = lib_path()
    if not os.path.exists(path):
        raise ImportError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(there is no CPU fallback)")
    L = C.CDLL(path)
    vp, i64, ci = C.c_void_p, C.c_int64, C.c_int
    pp = C.POINTER(_PP)
    L.lolhip_plan_create.argtypes = [pp, ci, _i64p, ci, ci, C.POINTER(vp)]
    L.lolhip_plan_create_roots.argtypes = [pp, ci, _i64p, ci, _i64p, _i64p, ci, C.POINTER(vp)]
    L.lolhip_plan_destroy.argtypes = [vp]
    L.lolhip_plan_destroy.restype = None
    for nm in ("lolhip_plan_n", "lolhip_plan_m"):
        getattr(L, nm).argtypes = [vp]
        getattr(L, nm).restype = i64
    L.lolhip_plan_T.argtypes = [vp]
    L.lolhip_plan_has_crt.argtypes = [vp]
    L.lolhip_plan_table.argtypes = [vp, ci, ci, _i64p, i64]
    L.lolhip_plan_table.restype = i64
    L.lolhip_good_q.argtypes = [i64, i64]
    L.lolhip_good_q.restype = i64
    for nm in ("crt", "crtinv", "l", "linv", "mulgpow", "mulgdec", "divgpow", "divgdec", "mulgcrt", "divgcrt"):
        getattr(L, f"lolhip_{nm}_batch").argtypes = [vp, vp, vp, i64]
    L.lolhip_mul_batch.argtypes = [vp, vp, vp, vp, i64]
    for nm in ("crtc", "crtinvc", "gaussian_dec"):
        getattr(L, f"lolhip_{nm}_batch").argtypes = [vp, vp, vp, i64]
    L.lolhip_polymul_batch.argtypes = [vp, vp, vp, vp, vp, i64]
    L.lolhip_ctmul_crt_batch.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, i64]
    L.lolhip_decompose_len.argtypes = [vp, i64]
    L.lolhip_gadget.argtypes = [vp, i64, _i64p, i64]
    L.lolhip_decompose_batch.argtypes = [vp, vp, vp, i64, vp, i64]
    L.lolhip_knapsack_batch.argtypes = [vp, vp, vp, ci, vp, ci, vp, vp, i64]
    L.lolhip_keyswitch_batch.argtypes = [vp, vp, vp, i64, vp, ci, vp, vp, vp, i64]
    L.lolhip_rescale_drop_batch.argtypes = [vp, vp, vp, vp, i64]
    L.lolhip_ext_create.argtypes = [vp, vp, C.POINTER(vp)]
    L.lolhip_ext_destroy.argtypes = [vp]
    L.lolhip_ext_destroy.restype = None
    for nm in ("twace_powdec", "twace_crt", "embed_pow", "embed_dec", "embed_crt", "coeffs"):
        getattr(L, f"lolhip_{nm}_batch").argtypes = [vp, vp, vp, vp, i64]
    L.lolhip_evallin_batch.argtypes = [vp, vp, vp, vp, vp, vp, vp, i64]
    L.lolhip_tunnel_work_len.argtypes = [vp, vp, i64, i64]
    L.lolhip_tunnel_work_len.restype = i64
    L.lolhip_tunnel_batch.argtypes = [vp, vp, vp, vp, vp, vp, vp, i64, vp, vp, i64]
    L.lolhip_ext_table.argtypes = [vp, ci, _i32p, i64]
    L.lolhip_ext_table.restype = i64
    L.lolhip_op_host.argtypes = [vp, ci, _i64p, _i64p, i64]
    L.lolhip_ext_host.argtypes = [vp, ci, _i64p, _i64p, i64]
    L.lolhip_thread_release.argtypes = []
    L.lolhip_thread_release.restype = None
    u8p = C.POINTER(C.c_uint8)
    L.lolhip_rqproduct_read.argtypes = [u8p, i64, C.POINTER(C.c_uint32), _i64p, ci, C.POINTER(ci), _i64p, i64]
    L.lolhip_rqproduct_read.restype = i64
    L.lolhip_rqproduct_write.argtypes = [C.c_uint32, _i64p, ci, _i64p, i64, u8p, i64]
    L.lolhip_rqproduct_write.restype = i64
    L.lolhip_kshint_read.argtypes = [u8p, i64, C.POINTER(C.c_uint32), _i64p, ci, C.POINTER(ci), C.POINTER(ci), C.POINTER(ci), _i64p, i64]
    L.lolhip_kshint_read.restype = i64
    u32p, f64p = C.POINTER(C.c_uint32), C.POINTER(C.c_double)
    L.lolhip_r_read.argtypes = [u8p, i64, u32p, _i64p, i64]
    L.lolhip_secretkey_read.argtypes = [u8p, i64, u32p, f64p, _i64p, i64]
    L.lolhip_kqproduct_read.argtypes = [u8p, i64, u32p, _i64p, ci, C.POINTER(ci), f64p, i64]
    L.lolhip_linearrq_read.argtypes = [u8p, i64, u32p, u32p, C.POINTER(ci), u32p, _i64p, ci, C.POINTER(ci), _i64p, i64]
    L.lolhip_kshint_write.argtypes = [C.c_uint32, _i64p, ci, ci, ci, _i64p, i64, C.c_uint64, C.c_uint64, u8p, i64]
    L.lolhip_tunnelhint_read.argtypes = [u8p, i64, u32p, u32p, u32p, C.POINTER(C.c_uint64), _i64p, _i64p, _i64p, _i64p, ci]
    u8pp = C.POINTER(C.POINTER(C.c_uint8))
    L.lolhip_r_write.argtypes = [C.c_uint32, _i64p, i64, u8p, i64]
    L.lolhip_secretkey_write.argtypes = [C.c_uint32, C.c_double, _i64p, i64, u8p, i64]
    L.lolhip_linearrq_write.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, _i64p, ci, ci, _i64p, i64, u8p, i64]
    L.lolhip_tunnelhint_write.argtypes = [u8p, i64, u8pp, _i64p, ci, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, u8p, i64]
    L.lolhip_chain_read.argtypes = [u8p, i64, _i64p, _i64p, ci]
    L.lolhip_chain_write.argtypes = [u8pp, _i64p, ci, u8p, i64]
    for nm in ("r_write", "secretkey_write", "linearrq_write", "tunnelhint_write", "chain_read", "chain_write"):
        getattr(L, f"lolhip_{nm}").restype = i64
    for nm in ("r_read", "secretkey_read", "kqproduct_read", "linearrq_read", "kshint_write", "tunnelhint_read"):
        getattr(L, f"lolhip_{nm}").restype = i64
    L.lolhip_debug_set.argtypes = [C.c_char_p, ci]
    L.lolhip_copy_slab.argtypes = [vp, vp, vp, i64, ci]
    L.lolhip_device_count.restype = ci
    L.lolhip_version.restype = C.c_char_p
    L.lolhip_last_status.restype = ci
    _lib = L
    return L


def _check(rc, what=""):
    if rc == OK:
        return
    raise (NoDeviceError if rc == ERR_NO_DEVICE else LolHipError)(rc, what)


def device_count() -> int:
    return lib().lolhip_device_count()


def debug_set(name: str, value: bool) -> None:
    """Force (value true) or release a launch path of the library: lolhip_debug_set (tests, A/B runs)."""
    _check(lib().lolhip_debug_set(name.encode(), 1 if value else 0), f"lolhip_debug_set({name})")


def good_q(m: int, lower: int) -> int:
    """Head of `goodQs m lower` (ZqBasic.hs:71-73)."""
    return int(lib().lolhip_good_q(m, lower))


def factor_pps(m: int):
    """ppsFact (FactoredDefs.hs:360-361): [(p, e)] ascending."""
    out, p = [], 2
    while m > 1:
        if m % p == 0:
            e = 0
            while m % p == 0:
                m //= p
                e += 1
            out.append((p, e))
        p += 1 if p == 2 else 2
        if p * p > m and m > 1:
            out.append((m, 1))
            break
    return out


def rqproduct_write(m: int, qs, xs) -> bytes:
    """Serialise decoding-basis residues xs [n][T] as a Lol `RqProduct` message (lol/Lol.proto;
    IZipVector.hs:127-205): one Rq per modulus, centred lifts, unpacked sint64."""
    xs = np.ascontiguousarray(xs, dtype=np.int64)
    qa = np.ascontiguousarray(qs, dtype=np.int64)
    T = len(qa)
    n = xs.size // max(T, 1)
    L = lib()
    need = L.lolhip_rqproduct_write(m, qa.ctypes.data_as(_i64p), T, xs.ctypes.data_as(_i64p), n, None, 0)
    _check(min(need, 0))
    buf = (C.c_uint8 * max(need, 1))()
    wrote = L.lolhip_rqproduct_write(m, qa.ctypes.data_as(_i64p), T, xs.ctypes.data_as(_i64p), n, buf, need)
    _check(min(wrote, 0))
    return bytes(buf[:wrote])


def rqproduct_read(data: bytes):
    """Parse a Lol `RqProduct` message -> (m, [q_t], xs [n][T]) with canonical residues."""
    L = lib()
    raw = (C.c_uint8 * max(len(data), 1)).from_buffer_copy(data if data else b"\0")
    m, T = C.c_uint32(0), C.c_int(0)
    qs = np.zeros(16, dtype=np.int64)
    n = L.lolhip_rqproduct_read(raw, len(data), C.byref(m), qs.ctypes.data_as(_i64p), 16, C.byref(T), None, 0)
    _check(min(n, 0))
    xs = np.zeros((n, T.value), dtype=np.int64)
    n2 = L.lolhip_rqproduct_read(raw, len(data), C.byref(m), qs.ctypes.data_as(_i64p), 16, C.byref(T),
                                 xs.ctypes.data_as(_i64p), xs.size)
    _check(min(n2, 0))
    return int(m.value), [int(q) for q in qs[:T.value]], xs


def kshint_read(data: bytes):
    """Parse a SymmSHE `KSHint` message (lol-apps/SHE.proto) -> (m, [q_t], xs [L][K][n][T]),
    decoding basis; `plan.crt(plan.l(xs.reshape(L*K, n, T)))` is the hint slab of keySwitch."""
    L_ = lib()
    raw = (C.c_uint8 * max(len(data), 1)).from_buffer_copy(data if data else b"\0")
    m, T, Lh, K = C.c_uint32(0), C.c_int(0), C.c_int(0), C.c_int(0)
    qs = np.zeros(16, dtype=np.int64)
    args = (raw, len(data), C.byref(m), qs.ctypes.data_as(_i64p), 16, C.byref(T), C.byref(Lh), C.byref(K))
    n = L_.lolhip_kshint_read(*args, None, 0)
    _check(min(n, 0))
    xs = np.zeros((Lh.value, K.value, n, T.value), dtype=np.int64)
    _check(min(L_.lolhip_kshint_read(*args, xs.ctypes.data_as(_i64p), xs.size), 0))
    return int(m.value), [int(q) for q in qs[:T.value]], xs


def _raw(data: bytes):
    return (C.c_uint8 * max(len(data), 1)).from_buffer_copy(data if data else b"\0")


def r_read(data: bytes):
    """Lol.proto `R` -> (m, xs[n]) integer decoding-basis coefficients."""
    L_ = lib()
    raw, m = _raw(data), C.c_uint32(0)
    n = L_.lolhip_r_read(raw, len(data), C.byref(m), None, 0)
    _check(min(n, 0))
    xs = np.zeros(n, dtype=np.int64)
    _check(min(L_.lolhip_r_read(raw, len(data), C.byref(m), xs.ctypes.data_as(_i64p), n), 0))
    return int(m.value), xs


def secretkey_read(data: bytes):
    """SHE.proto `SecretKey` -> (m, scaled variance v, sk[n])."""
    L_ = lib()
    raw, m, v = _raw(data), C.c_uint32(0), C.c_double(0)
    n = L_.lolhip_secretkey_read(raw, len(data), C.byref(m), C.byref(v), None, 0)
    _check(min(n, 0))
    xs = np.zeros(n, dtype=np.int64)
    _check(min(L_.lolhip_secretkey_read(raw, len(data), C.byref(m), C.byref(v), xs.ctypes.data_as(_i64p), n), 0))
    return int(m.value), float(v.value), xs


def kqproduct_read(data: bytes):
    """Lol.proto `KqProduct` -> (m, [q_t], xs [n][T] float64)."""
    L_ = lib()
    raw, m, T = _raw(data), C.c_uint32(0), C.c_int(0)
    qs = np.zeros(16, dtype=np.int64)
    n = L_.lolhip_kqproduct_read(raw, len(data), C.byref(m), qs.ctypes.data_as(_i64p), 16, C.byref(T), None, 0)
    _check(min(n, 0))
    xs = np.zeros((n, T.value), dtype=np.float64)
    _check(min(L_.lolhip_kqproduct_read(raw, len(data), C.byref(m), qs.ctypes.data_as(_i64p), 16, C.byref(T),
                                        xs.ctypes.data_as(C.POINTER(C.c_double)), xs.size), 0))
    return int(m.value), [int(q) for q in qs[:T.value]], xs


def linearrq_read(data: bytes):
    """Lol.proto `LinearRq` -> (e, r, m_out, [q_t], xs [C][n][T]) decoding basis, canonical residues."""
    L_ = lib()
    raw = _raw(data)
    e, r, m, Cn, T = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0), C.c_int(0), C.c_int(0)
    qs = np.zeros(16, dtype=np.int64)
    args = (raw, len(data), C.byref(e), C.byref(r), C.byref(Cn), C.byref(m), qs.ctypes.data_as(_i64p), 16, C.byref(T))
    n = L_.lolhip_linearrq_read(*args, None, 0)
    _check(min(n, 0))
    xs = np.zeros((Cn.value, n, T.value), dtype=np.int64)
    _check(min(L_.lolhip_linearrq_read(*args, xs.ctypes.data_as(_i64p), xs.size), 0))
    return int(e.value), int(r.value), int(m.value), [int(q) for q in qs[:T.value]], xs


def kshint_write(m: int, qs, xs, gad=(0, 0)) -> bytes:
    """xs [L][K][n][T] decoding-basis residues -> SHE.proto `KSHint` bytes (gad = TypeRep words)."""
    xs = np.ascontiguousarray(xs, dtype=np.int64)
    Lh, K, n, T = xs.shape
    qa = np.ascontiguousarray(qs, dtype=np.int64)
    L_ = lib()
    args = (m, qa.ctypes.data_as(_i64p), T, Lh, K, xs.ctypes.data_as(_i64p), n, int(gad[0]), int(gad[1]))
    need = L_.lolhip_kshint_write(*args, None, 0)
    _check(min(need, 0))
    buf = (C.c_uint8 * max(need, 1))()
    wrote = L_.lolhip_kshint_write(*args, buf, need)
    _check(min(wrote, 0))
    return bytes(buf[:wrote])


def tunnelhint_read(data: bytes):
    """SHE.proto `TunnelHint` -> dict(e, r, s, p, func = linearrq_read(...), hints = [kshint_read(...)])."""
    L_ = lib()
    raw = _raw(data)
    e, r, s, p = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0), C.c_uint64(0)
    fo, fl = C.c_int64(0), C.c_int64(0)
    nh = L_.lolhip_tunnelhint_read(raw, len(data), C.byref(e), C.byref(r), C.byref(s), C.byref(p), C.byref(fo), C.byref(fl), None, None, 0)
    _check(min(nh, 0))
    ho, hl = np.zeros(max(nh, 1), dtype=np.int64), np.zeros(max(nh, 1), dtype=np.int64)
    _check(min(L_.lolhip_tunnelhint_read(raw, len(data), C.byref(e), C.byref(r), C.byref(s), C.byref(p), C.byref(fo), C.byref(fl),
                                         ho.ctypes.data_as(_i64p), hl.ctypes.data_as(_i64p), nh), 0))
    return {"e": int(e.value), "r": int(r.value), "s": int(s.value), "p": int(p.value),
            "func": linearrq_read(data[fo.value: fo.value + fl.value]),
            "hints": [kshint_read(data[int(o): int(o) + int(l)]) for o, l in zip(ho[:nh], hl[:nh])]}


def _two_pass(fn, *args) -> bytes:
    """size query (out = NULL), then the write"""
    need = fn(*args, None, 0)
    _check(min(need, 0))
    buf = (C.c_uint8 * max(need, 1))()
    wrote = fn(*args, buf, need)
    _check(min(wrote, 0))
    return bytes(buf[:wrote])


def _byte_parts(parts):
    """list of bytes objects -> (array of pointers, array of lengths, keep-alive list)"""
    keep = [(C.c_uint8 * max(len(p), 1)).from_buffer_copy(p if p else b"\0") for p in parts]
    ptrs = (C.POINTER(C.c_uint8) * max(len(parts), 1))(*[C.cast(k, C.POINTER(C.c_uint8)) for k in keep])
    lens = np.array([len(p) for p in parts] or [0], dtype=np.int64)
    return ptrs, lens, keep


def r_write(m: int, xs) -> bytes:
    """integer decoding-basis coefficients xs [n] -> Lol.proto `R` bytes."""
    xs = np.ascontiguousarray(xs, dtype=np.int64)
    return _two_pass(lib().lolhip_r_write, m, xs.ctypes.data_as(_i64p), xs.size)


def secretkey_write(m: int, v: float, xs) -> bytes:
    """SHE.proto `SecretKey` (ring element xs [n] as `R`, scaled variance v)."""
    xs = np.ascontiguousarray(xs, dtype=np.int64)
    return _two_pass(lib().lolhip_secretkey_write, m, C.c_double(v), xs.ctypes.data_as(_i64p), xs.size)


def linearrq_write(e: int, r: int, m: int, qs, xs) -> bytes:
    """xs [C][n][T] (values of an E-linear function on the relative decoding basis) -> Lol.proto `LinearRq` bytes."""
    xs = np.ascontiguousarray(xs, dtype=np.int64)
    Cn, n, T = xs.shape
    qa = np.ascontiguousarray(qs, dtype=np.int64)
    return _two_pass(lib().lolhip_linearrq_write, e, r, m, qa.ctypes.data_as(_i64p), T, Cn, xs.ctypes.data_as(_i64p), n)


def tunnelhint_write(func: bytes, hints, e: int, r: int, s: int, p: int) -> bytes:
    """SHE.proto `TunnelHint` from an encoded LinearRq and a list of encoded KSHints."""
    fbuf = (C.c_uint8 * max(len(func), 1)).from_buffer_copy(func if func else b"\0")
    ptrs, lens, keep = _byte_parts(list(hints))
    return _two_pass(lib().lolhip_tunnelhint_write, fbuf, len(func), ptrs, lens.ctypes.data_as(_i64p), len(hints), e, r, s, p)


def chain_write(elems) -> bytes:
    """HomomPRF.proto chain (LinearFuncChain / TunnelHintChain / RoundHintChain) of encoded elements."""
    ptrs, lens, keep = _byte_parts(list(elems))
    return _two_pass(lib().lolhip_chain_write, ptrs, lens.ctypes.data_as(_i64p), len(elems))


def chain_read(data: bytes):
    """HomomPRF.proto chain -> list of the encoded elements (hand them to linearrq_read / tunnelhint_read / kshint_read)."""
    L_ = lib()
    raw = _raw(data)
    cnt = L_.lolhip_chain_read(raw, len(data), None, None, 0)
    _check(min(cnt, 0))
    off, ln = np.zeros(max(cnt, 1), dtype=np.int64), np.zeros(max(cnt, 1), dtype=np.int64)
    _check(min(L_.lolhip_chain_read(raw, len(data), off.ctypes.data_as(_i64p), ln.ctypes.data_as(_i64p), cnt), 0))
    return [data[int(o): int(o) + int(l)] for o, l in zip(off[:cnt], ln[:cnt])]


def _np(a):
    a = np.ascontiguousarray(a, dtype=np.int64)
    return a, a.ctypes.data_as(_i64p)


def _devptr(x):
    """raw device address of a torch CUDA int64 tensor, or an int address"""
    if isinstance(x, int):
        return x
    if not (x.is_cuda and x.is_contiguous() and x.dtype.__str__() == "torch.int64"):
        raise TypeError("expected a contiguous int64 CUDA tensor")
    return x.data_ptr()


def _stream(stream):
    if stream is None:
        try:
            import torch
            if torch.cuda.is_available():
                return torch.cuda.current_stream().cuda_stream
        except ImportError:
            pass
        return 0
    return int(stream)


class Plan:
    """Twiddles, g vectors, stage programs and moduli for one (m, moduli), resident in HBM.

    Mirrors what lol-cpp's shim marshals per call: `ru`/`ruInv` (CPP.hs:422-442),
    `mhatInv` (ZqBasic.hs:167-171), `gCRT`/`gInvCRT` (CPP.hs:444-454), the prime-power
    list (CPP.hs:325-337) and the moduli (Backend.hs:195-199)."""

    def __init__(self, pps, qs, host_only=False, omega_pp=None, mhatinv=None):
        self.pps = [(int(p), int(e)) for p, e in pps]
        self.qs = [int(q) for q in qs]
        arr = (_PP * max(1, len(self.pps)))()
        for i, (p, e) in enumerate(self.pps):
            arr[i].prime, arr[i].exponent = p, e
        qa = (C.c_int64 * len(self.qs))(*self.qs)
        h = C.c_void_p()
        L = lib()
        if omega_pp is None and mhatinv is None:
            rc = L.lolhip_plan_create(arr, len(self.pps), qa, len(self.qs), int(host_only), C.byref(h))
        else:
            om = None if omega_pp is None else (C.c_int64 * len(omega_pp))(*[int(v) for v in omega_pp])
            mh = None if mhatinv is None else (C.c_int64 * len(mhatinv))(*[int(v) for v in mhatinv])
            rc = L.lolhip_plan_create_roots(arr, len(self.pps), qa, len(self.qs), om, mh, int(host_only), C.byref(h))
        _check(rc, f"plan_create(pps={self.pps}, qs={self.qs})")
        self._h = h
        self.n = int(L.lolhip_plan_n(h))
        self.m = int(L.lolhip_plan_m(h))
        self.T = int(L.lolhip_plan_T(h))
        self.has_crt = bool(L.lolhip_plan_has_crt(h))

    @classmethod
    def for_index(cls, m, qs, **kw):
        return cls(factor_pps(m), qs, **kw)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None:
            _lib.lolhip_plan_destroy(h)

    # ---- tables -------------------------------------------------------------------
    def _table(self, which, k=0):
        L = lib()
        cnt = L.lolhip_plan_table(self._h, which, k, None, 0)
        out = np.zeros(cnt, dtype=np.int64)
        if cnt:
            L.lolhip_plan_table(self._h, which, k, out.ctypes.data_as(_i64p), cnt)
        return out

    def ru(self, k): return self._table(0, k)
    def ruInv(self, k): return self._table(1, k)
    def mhatInv(self): return self._table(2)
    def gCRT(self): return self._table(3).reshape(self.n, self.T)
    def gInvCRT(self): return self._table(4).reshape(self.n, self.T)

    def program(self, inverse=False):
        """Stage program of a lone crt / crtInv (inspection): rows (kind, prime or first level, length or levels, stride)."""
        return self._table(11 if inverse else 10).reshape(-1, 4)

    # ---- helpers ------------------------------------------------------------------
    def _batch(self, a):
        per = self.n * self.T
        if a.size % per:
            raise ValueError(f"array of {a.size} residues is not a whole number of [n={self.n}, T={self.T}] polynomials")
        return a.size // per

    def _host(self, op, y, b=None):
        y, yp = _np(y)
        y = y.copy()
        yp = y.ctypes.data_as(_i64p)
        B = self._batch(y)
        bp = None
        if b is not None:
            b, bp = _np(b)
            if b.size != y.size:
                raise ValueError("operand shapes differ")
        rc = lib().lolhip_op_host(self._h, op, yp, bp, B)
        if rc == ERR_NOT_DIVISIBLE:
            return None
        _check(rc)
        return y

    def _dev(self, name, y, stream):
        B = self._batch_t(y)
        rc = getattr(lib(), f"lolhip_{name}_batch")(self._h, _stream(stream), _devptr(y), B)
        if rc == ERR_NOT_DIVISIBLE:
            return None
        _check(rc)
        return y

    def _batch_t(self, t):
        per = self.n * self.T
        numel = t.numel()
        if numel % per:
            raise ValueError("tensor is not a whole number of polynomials")
        return numel // per

    def _op(self, op, name, y, stream=None):
        if isinstance(y, np.ndarray) or isinstance(y, (list, tuple)):
            return self._host(op, y)
        return self._dev(name, y, stream)

    # ---- the Tensor interface (in place for device tensors, copies for numpy) -----
    def crt(self, y, stream=None): return self._op(OP_CRT, "crt", y, stream)
    def crtInv(self, y, stream=None): return self._op(OP_CRTINV, "crtinv", y, stream)
    def l(self, y, stream=None): return self._op(OP_L, "l", y, stream)
    def lInv(self, y, stream=None): return self._op(OP_LINV, "linv", y, stream)
    def mulGPow(self, y, stream=None): return self._op(OP_MULGPOW, "mulgpow", y, stream)
    def mulGDec(self, y, stream=None): return self._op(OP_MULGDEC, "mulgdec", y, stream)
    def divGPow(self, y, stream=None): return self._op(OP_DIVGPOW, "divgpow", y, stream)
    def divGDec(self, y, stream=None): return self._op(OP_DIVGDEC, "divgdec", y, stream)
    def mulGCRT(self, y, stream=None): return self._op(OP_MULGCRT, "mulgcrt", y, stream)
    def divGCRT(self, y, stream=None): return self._op(OP_DIVGCRT, "divgcrt", y, stream)

    def mul(self, a, b, stream=None):
        """zipWithT (*): a * b pointwise (a is overwritten when it is a device tensor)."""
        if isinstance(a, np.ndarray):
            return self._host(OP_MUL, a, b)
        _check(lib().lolhip_mul_batch(self._h, _stream(stream), _devptr(a), _devptr(b), self._batch_t(a)))
        return a

    def polymul(self, a, b, out=None, stream=None):
        """crtInv(crt a * crt b): one ring product per batch item, powerful basis in and out."""
        if isinstance(a, np.ndarray):
            return self._host(OP_POLYMUL, a, b)
        out = a if out is None else out
        _check(lib().lolhip_polymul_batch(self._h, _stream(stream), _devptr(out), _devptr(a), _devptr(b), self._batch_t(a)))
        return out


    # ---- floating-point members: float64, tolerance contract -----------
    def _float_op(self, name, y, complex_):
        """numpy in -> numpy out (staged through HBM); torch CUDA tensor -> in place."""
        import torch
        host = isinstance(y, np.ndarray)
        want = torch.complex128 if complex_ else torch.float64
        t = torch.from_numpy(np.ascontiguousarray(y, dtype=np.complex128 if complex_ else np.float64)).cuda() if host else y
        if not (t.is_cuda and t.is_contiguous() and t.dtype == want):
            raise TypeError(f"expected a contiguous {want} CUDA tensor")
        if t.numel() % self.n:
            raise ValueError("tensor is not a whole number of polynomials")
        _check(getattr(lib(), f"lolhip_{name}_batch")(self._h, _stream(None), t.data_ptr(), t.numel() // self.n))
        return t.cpu().numpy() if host else t

    def crtC(self, y):
        """CRT over C (tensorCRTC, crt.cpp:583-586): complex128 [..., n]."""
        return self._float_op("crtc", y, True)

    def crtInvC(self, y):
        """inverse CRT over C including mhat^-1 (tensorCRTInvC, crt.cpp:589-598)."""
        return self._float_op("crtinvc", y, True)

    def gaussianDec(self, y):
        """iid real Gaussians [..., n] -> decoding-basis sample (tensorGaussianDec, random.cpp:61-64)."""
        return self._float_op("gaussian_dec", y, False)

    # ---- ring-level pipelines of SymmSHE (include/lolhip.h) -----------
    # numpy in -> numpy out (staged through HBM with torch); CUDA tensors in -> CUDA tensors out.
    @staticmethod
    def _stage(*arrays):
        import torch
        host = isinstance(arrays[0], np.ndarray)
        dev = [None if a is None else (torch.from_numpy(np.ascontiguousarray(a, dtype=np.int64)).cuda() if host else a)
               for a in arrays]
        return host, dev

    @staticmethod
    def _unstage(host, *tensors):
        out = tuple(t.cpu().numpy() if host else t for t in tensors)
        return out if len(out) > 1 else out[0]

    def ctMulCRT(self, c0, c1, d0, d1, stream=None):
        """(g c0 d0, g (c0 d1 + c1 d0), g c1 d1): mulG <$> c*d for two linear ciphertexts, every
        operand in the CRT basis (SymmSHE.hs:444-449)."""
        import torch
        host, (c0, c1, d0, d1) = self._stage(c0, c1, d0, d1)
        e = [torch.empty_like(c0) for _ in range(3)]
        _check(lib().lolhip_ctmul_crt_batch(self._h, _stream(stream), *map(_devptr, (c0, c1, d0, d1, *e)), self._batch_t(c0)))
        return self._unstage(host, *e)

    def decomposeLen(self, base):
        L = lib().lolhip_decompose_len(self._h, int(base))
        _check(min(L, 0))
        return L

    def gadget(self, base):
        """gadget vector [L][T] (Gadget.hs:92-94 over ZqBasic.hs:227-229,248-253)."""
        L = self.decomposeLen(base)
        out = np.zeros((L, self.T), dtype=np.int64)
        _check(min(lib().lolhip_gadget(self._h, int(base), out.ctypes.data_as(_i64p), out.size), 0))
        return out

    def decompose(self, c, base, stream=None):
        """powerful-basis c [B][n][T] -> reduced digit polynomials [L][B][n][T] (Cyc.hs:592-604)."""
        import torch
        host, (c,) = self._stage(c)
        B, L = self._batch_t(c), self.decomposeLen(base)
        out = torch.empty((L, B, self.n, self.T), dtype=torch.int64, device=c.device)
        _check(lib().lolhip_decompose_batch(self._h, _stream(stream), _devptr(c), int(base), _devptr(out), B))
        return self._unstage(host, out)

    def knapsack(self, xs_crt, hint, addend=None, stream=None):
        """sum_j xs_j *>> hint_j (+ addend), CRT basis (SymmSHE.hs:302-304): xs [L][B][n][T],
        hint [L][K][n][T] -> [K][B][n][T]."""
        import torch
        host, (xs_crt, hint, addend) = self._stage(xs_crt, hint, addend)
        L, K = xs_crt.shape[0], hint.shape[1]
        B = self._batch_t(xs_crt) // max(L, 1)
        out = torch.empty((K, B, self.n, self.T), dtype=torch.int64, device=hint.device)
        _check(lib().lolhip_knapsack_batch(self._h, _stream(stream), _devptr(xs_crt), L, _devptr(hint), K,
                                           None if addend is None else _devptr(addend), _devptr(out), B))
        return self._unstage(host, out)

    def keySwitch(self, c2_pow, base, hint, addend=None, stream=None):
        """addend + knapsack hint (crt (reduce <$> decompose c2)) — `switch`, SymmSHE.hs:312-314."""
        import torch
        host, (c2_pow, hint, addend) = self._stage(c2_pow, hint, addend)
        B, L, K = self._batch_t(c2_pow), self.decomposeLen(base), hint.shape[1]
        work = torch.empty((L, B, self.n, self.T), dtype=torch.int64, device=c2_pow.device)
        out = torch.empty((K, B, self.n, self.T), dtype=torch.int64, device=c2_pow.device)
        _check(lib().lolhip_keyswitch_batch(self._h, _stream(stream), _devptr(c2_pow), int(base), _devptr(hint), K,
                                            None if addend is None else _devptr(addend), _devptr(out), _devptr(work), B))
        return self._unstage(host, out)

    def rescaleDropFirst(self, c, stream=None):
        """RescaleCyc (a,b) -> b (Cyc.hs:529-542): [B][n][T] -> [B][n][T-1]."""
        import torch
        host, (c,) = self._stage(c)
        B = self._batch_t(c)
        out = torch.empty((B, self.n, self.T - 1), dtype=torch.int64, device=c.device)
        _check(lib().lolhip_rescale_drop_batch(self._h, _stream(stream), _devptr(c), _devptr(out), B))
        return self._unstage(host, out)


class Ext:
    """Index tables and kernels for a ring extension m | m' (Tensor.hs:390-509, Extension.hs:54-129)."""

    def __init__(self, lo: Plan, hi: Plan):
        self.lo, self.hi = lo, hi
        h = C.c_void_p()
        _check(lib().lolhip_ext_create(lo._h, hi._h, C.byref(h)), "ext_create")
        self._h = h

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None:
            _lib.lolhip_ext_destroy(h)

    def table(self, which):
        L = lib()
        cnt = L.lolhip_ext_table(self._h, which, None, 0)
        out = np.zeros(cnt, dtype=np.int32)
        if cnt:
            L.lolhip_ext_table(self._h, which, out.ctypes.data_as(_i32p), cnt)
        return out

    def _run(self, op, name, x, to_hi, out, stream):
        src, dst = (self.lo, self.hi) if to_hi else (self.hi, self.lo)
        if isinstance(x, np.ndarray):
            x, xp = _np(x)
            B = src._batch(x)
            res = np.zeros((B, dst.n, dst.T), dtype=np.int64)
            _check(lib().lolhip_ext_host(self._h, op, res.ctypes.data_as(_i64p), xp, B))
            return res
        B = src._batch_t(x)
        if out is None:
            import torch
            out = torch.empty((B, dst.n, dst.T), dtype=torch.int64, device=x.device)
        _check(getattr(lib(), f"lolhip_{name}_batch")(self._h, _stream(stream), _devptr(out), _devptr(x), B))
        return out

    def coeffs(self, x, out=None, stream=None):
        """[n'/n][B][n][T] coefficient vectors over the relative powerful/decoding basis (Tensor.hs:174)."""
        rel = self.hi.n // self.lo.n
        if isinstance(x, np.ndarray):
            x, xp = _np(x)
            B = self.hi._batch(x)
            res = np.zeros((rel, B, self.lo.n, self.lo.T), dtype=np.int64)
            _check(lib().lolhip_ext_host(self._h, EXT_COEFFS, res.ctypes.data_as(_i64p), xp, B))
            return res
        B = self.hi._batch_t(x)
        if out is None:
            import torch
            out = torch.empty((rel, B, self.lo.n, self.lo.T), dtype=torch.int64, device=x.device)
        _check(lib().lolhip_coeffs_batch(self._h, _stream(stream), _devptr(out), _devptr(x), B))
        return out

    def evalLin(self, es: "Ext", r_dec, ys_crt, stream=None):
        """sum_i ys_i * embed(coeffsDec_i r)  (Linear.hs:75-79): self = E in R, es = E in S;
        r_dec [B][n_R][T] decoding basis, ys_crt [n_R/n_E][n_S][T] CRT basis -> [B][n_S][T] CRT basis."""
        import torch
        host, (r_dec, ys_crt) = Plan._stage(r_dec, ys_crt)
        B, rel = self.hi._batch_t(r_dec), self.hi.n // self.lo.n
        S = es.hi
        work = torch.empty((rel * B * (self.lo.n + S.n) * S.T,), dtype=torch.int64, device=r_dec.device)
        out = torch.empty((B, S.n, S.T), dtype=torch.int64, device=r_dec.device)
        _check(lib().lolhip_evallin_batch(self._h, es._h, _stream(stream), _devptr(r_dec), _devptr(ys_crt),
                                          _devptr(out), _devptr(work), B))
        return Plan._unstage(host, out)

    def tunnel(self, es: "Ext", c0_dec, c1_pow, ys_crt, hints, base, stream=None):
        """SymmSHE `tunnel` after toMSD . absorbGFactors (SymmSHE.hs:549-570): self = E' in R', es = E' in S';
        [c0 (decoding basis), c1 (powerful basis)] over R' -> [2][B][n_S][T] CRT-basis linear ciphertext over S'.
        hints [n_R/n_E][L][2][n_S][T]."""
        import torch
        host, (c0_dec, c1_pow, ys_crt, hints) = Plan._stage(c0_dec, c1_pow, ys_crt, hints)
        B, S = self.hi._batch_t(c0_dec), es.hi
        wl = lib().lolhip_tunnel_work_len(self._h, es._h, int(base), B)
        _check(min(wl, 0))
        work = torch.empty((max(wl, 1),), dtype=torch.int64, device=c0_dec.device)
        out = torch.empty((2, B, S.n, S.T), dtype=torch.int64, device=c0_dec.device)
        _check(lib().lolhip_tunnel_batch(self._h, es._h, _stream(stream), _devptr(c0_dec), _devptr(c1_pow), _devptr(ys_crt),
                                         _devptr(hints), int(base), _devptr(out), _devptr(work), B))
        return Plan._unstage(host, out)

    def twacePowDec(self, x, out=None, stream=None): return self._run(EXT_TWACE_POWDEC, "twace_powdec", x, False, out, stream)
    def twaceCRT(self, x, out=None, stream=None): return self._run(EXT_TWACE_CRT, "twace_crt", x, False, out, stream)
    def embedPow(self, x, out=None, stream=None): return self._run(EXT_EMBED_POW, "embed_pow", x, True, out, stream)
    def embedDec(self, x, out=None, stream=None): return self._run(EXT_EMBED_DEC, "embed_dec", x, True, out, stream)
    def embedCRT(self, x, out=None, stream=None): return self._run(EXT_EMBED_CRT, "embed_crt", x, True, out, stream)
