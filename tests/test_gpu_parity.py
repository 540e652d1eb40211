"""GPU parity tests (-m gpu): every entry point of include/lolhip.h, called through the C ABI,
against (a) the committed golden vectors produced by the reference's own C++, (b) the CPU
oracle on fresh seeded inputs, and (c) at BASELINE.json's full sizes, size-independent
properties plus strided samples.  Bit-exact everywhere (integer arithmetic).

All comparisons are `np.array_equal`; there is no tolerance in this file.
"""
import ctypes as C

import numpy as np
import pytest

from oracle import lolmath as lm
from oracle.oracle import Params
from params import BIG, PLAN_NAME, PRIME_OPS, PRIMEOPS_ONLY, TENSOR1, TENSOR2, BENCH1, BENCH2

pytestmark = pytest.mark.gpu


def _i64(a):
    return np.ascontiguousarray(np.asarray(a).astype(np.int64))


def _params_nocrt(m, qs):
    P = Params.__new__(Params)
    P.pps = lm.factor_pps(m)
    P.qs, P.T, P.m, P.n = list(qs), len(qs), m, lm.totient_pps(P.pps)
    return P


# ---------------------------------------------------------------------------------------
# (a) golden vectors from the reference's own lol-cpp (tests/golden/make_golden.py)
# ---------------------------------------------------------------------------------------

@pytest.mark.parametrize("i", range(len(TENSOR1)))
def test_golden_tensor1(gpu, golden, i):
    m, qs = TENSOR1[i]
    P = gpu.Plan(lm.factor_pps(m), qs)
    y, z = _i64(golden[f"t1_{i}/y"]), _i64(golden[f"t1_{i}/z"])
    assert np.array_equal(P.crt(y), _i64(golden[f"t1_{i}/crt"]))
    assert np.array_equal(P.crtInv(y), _i64(golden[f"t1_{i}/crtinv"]))
    assert np.array_equal(P.mul(y, z), _i64(golden[f"t1_{i}/mul"]))
    assert np.array_equal(P.polymul(y, z), _i64(golden[f"t1_{i}/polymul"]))
    for op in PRIME_OPS:
        got = getattr(P, PLAN_NAME[op])(y)
        key = f"t1_{i}/{op}"
        if key in golden:
            assert np.array_equal(got, _i64(golden[key])), op
        else:
            assert got is None, op


@pytest.mark.parametrize("i", range(len(PRIMEOPS_ONLY)))
def test_golden_noncrt_moduli(gpu, golden, i):
    m, qs = PRIMEOPS_ONLY[i]
    P = gpu.Plan(lm.factor_pps(m), qs)
    assert not P.has_crt
    y = _i64(golden[f"po_{i}/y"])
    for op in PRIME_OPS:
        got = getattr(P, PLAN_NAME[op])(y)
        key = f"po_{i}/{op}"
        if key in golden:
            assert np.array_equal(got, _i64(golden[key])), op
        else:
            assert got is None
    with pytest.raises(gpu.LolHipError):
        P.crt(y)


@pytest.mark.parametrize("i", range(len(BIG)))
def test_golden_baseline_configs(gpu, golden, i):
    """config 1 exactly (m=1024, q=12289, seed 1) and configs 2/4 at CT-valid moduli."""
    m, qs, _seed = BIG[i]
    P = gpu.Plan(lm.factor_pps(m), qs)
    y, z = _i64(golden[f"big_{i}/y"]), _i64(golden[f"big_{i}/z"])
    assert np.array_equal(P.crt(y), _i64(golden[f"big_{i}/crt"]))
    assert np.array_equal(P.crtInv(y), _i64(golden[f"big_{i}/crtinv"]))
    assert np.array_equal(P.polymul(y, z), _i64(golden[f"big_{i}/polymul"]))


@pytest.mark.parametrize("i", range(len(TENSOR2)))
def test_golden_twace_embed(gpu, golden, i):
    m, m2, qs = TENSOR2[i]
    X = gpu.Ext(gpu.Plan(lm.factor_pps(m), qs), gpu.Plan(lm.factor_pps(m2), qs))
    lo, hi = _i64(golden[f"t2_{i}/lo"]), _i64(golden[f"t2_{i}/hi"])
    assert np.array_equal(X.embedPow(lo), _i64(golden[f"t2_{i}/embed_pow"]))
    assert np.array_equal(X.twacePowDec(hi), _i64(golden[f"t2_{i}/twace_powdec"]))
    assert np.array_equal(X.embedCRT(lo), _i64(golden[f"t2_{i}/embed_crt"]))
    assert np.array_equal(X.twaceCRT(hi), _i64(golden[f"t2_{i}/twace_crt"]))
    assert np.array_equal(X.embedDec(lo), _i64(golden[f"t2_{i}/embed_dec"]))


# ---------------------------------------------------------------------------------------
# the ten drop-in symbols, called exactly as lol-cpp's Haskell shim calls them
# (Backend.hs:193-241): host pointers, in place, caller-supplied twiddles
# ---------------------------------------------------------------------------------------

class _PP(C.Structure):
    _fields_ = [("prime", C.c_int16), ("exponent", C.c_int16)]


def _dropin(gpu):
    L = C.CDLL(gpu.lib_path())
    i16, i64, vp = C.c_int16, C.c_int64, C.c_void_p
    L.tensorCRTRq.argtypes = [i16, vp, i64, vp, i16, vp, vp]
    L.tensorCRTInvRq.argtypes = [i16, vp, i64, vp, i16, vp, vp, vp]
    L.mulRq.argtypes = [i16, vp, vp, i64, vp]
    for nm in ("tensorLRq", "tensorLInvRq", "tensorGPowRq", "tensorGDecRq", "tensorGInvPowRq", "tensorGInvDecRq"):
        getattr(L, nm).argtypes = [i16, vp, i64, vp, i16, vp]
    for nm in ("tensorCRTRq", "tensorCRTInvRq", "mulRq", "tensorLRq", "tensorLInvRq", "tensorGPowRq", "tensorGDecRq"):
        getattr(L, nm).restype = None
    L.tensorGInvPowRq.restype = i16
    L.tensorGInvDecRq.restype = i16
    return L


@pytest.mark.parametrize("m,qs", TENSOR1 + [(1024, [12289]), (45, [1171, 1531]), (2 ** 14, [lm.first_good_q(2 ** 14, 2 ** 60)])])
def test_dropin_symbols(gpu, cpuref, m, qs):
    L = _dropin(gpu)
    P = Params(lm.factor_pps(m), qs)
    rng = np.random.default_rng(m + 5)
    pe = (_PP * max(1, len(P.pps)))()
    for k, (p, e) in enumerate(P.pps):
        pe[k].prime, pe[k].exponent = p, e
    q = np.array(qs, dtype=np.int64)
    ru = [np.array(t, dtype=np.int64) for t in P.ru]
    rui = [np.array(t, dtype=np.int64) for t in P.ruinv]
    rup = (C.c_void_p * max(1, len(ru)))(*[a.ctypes.data for a in ru])
    ruip = (C.c_void_p * max(1, len(rui)))(*[a.ctypes.data for a in rui])
    mh = np.array(P.mhatinv, dtype=np.int64)
    y0 = P.random(rng, 1)[0]
    y = y0.copy()
    L.tensorCRTRq(P.T, y.ctypes.data, P.n, pe, len(P.pps), rup, q.ctypes.data)
    assert L.lolhip_last_status() == 0
    assert np.array_equal(y, cpuref.crt(P, y0).reshape(y0.shape))
    L.tensorCRTInvRq(P.T, y.ctypes.data, P.n, pe, len(P.pps), ruip, mh.ctypes.data, q.ctypes.data)
    assert np.array_equal(y, y0)
    z = P.random(rng, 1)[0]
    a = y0.copy()
    L.mulRq(P.T, a.ctypes.data, z.ctypes.data, P.n, q.ctypes.data)
    assert np.array_equal(a, cpuref.mul(P, y0, z).reshape(y0.shape))
    for sym, op in (("tensorLRq", "l"), ("tensorLInvRq", "linv"), ("tensorGPowRq", "gpow"), ("tensorGDecRq", "gdec")):
        a = y0.copy()
        getattr(L, sym)(P.T, a.ctypes.data, P.n, pe, len(P.pps), q.ctypes.data)
        assert np.array_equal(a, getattr(cpuref, op)(P, y0).reshape(y0.shape)), sym
    for sym, op in (("tensorGInvPowRq", "ginvpow"), ("tensorGInvDecRq", "ginvdec")):
        a = y0.copy()
        ret = getattr(L, sym)(P.T, a.ctypes.data, P.n, pe, len(P.pps), q.ctypes.data)
        want = getattr(cpuref, op)(P, y0)
        assert (ret == 1) == (want is not None)
        if want is not None:
            assert np.array_equal(a, want.reshape(y0.shape)), sym


def test_dropin_honours_caller_roots(gpu, cpuref):
    """The reference transforms with whatever roots the caller marshals (CPP.hs:422-432);
    a different primitive root must give the matching different CRT."""
    m, q = 32, 97
    pps = lm.factor_pps(m)
    P = Params(pps, [q])
    w = lm.omega(m, q)
    w2 = pow(w, 3, q)                       # another primitive 32nd root
    P.ru = [[pow(w2, i, q) for i in range(m)]]
    rng = np.random.default_rng(0)
    y0 = P.random(rng, 1)[0]
    L = _dropin(gpu)
    pe = (_PP * 1)()
    pe[0].prime, pe[0].exponent = 2, 5
    ru = np.array(P.ru[0], dtype=np.int64)
    rup = (C.c_void_p * 1)(ru.ctypes.data)
    qa = np.array([q], dtype=np.int64)
    y = y0.copy()
    L.tensorCRTRq(1, y.ctypes.data, P.n, pe, 1, rup, qa.ctypes.data)
    assert np.array_equal(y, cpuref.crt(P, y0).reshape(y0.shape))
    assert not np.array_equal(y, cpuref.crt(Params(pps, [q]), y0).reshape(y0.shape))


# ---------------------------------------------------------------------------------------
# (b) fresh random inputs against the oracle: every n = 2^L the fast path instantiates,
#     31/61-bit moduli, RNS tuples, mixed-radix indices, batch shapes
# ---------------------------------------------------------------------------------------

@pytest.mark.parametrize("L", range(1, 15))
@pytest.mark.parametrize("T,lower", [(1, 2 ** 60), (3, 2 ** 29), (2, 2 ** 61), (2, 2 ** 16)])
def test_pow2_all_sizes(gpu, cpuref, L, T, lower):
    m = 2 ** (L + 1)
    g = lm.good_qs(m, lower)
    qs = [next(g) for _ in range(T)]
    P, R = gpu.Plan([(2, L + 1)], qs), Params([(2, L + 1)], qs)
    rng = np.random.default_rng(L * 10 + T)
    for B in (1, 5):
        y, z = R.random(rng, B), R.random(rng, B)
        assert np.array_equal(P.crt(y), cpuref.crt(R, y))
        assert np.array_equal(P.crtInv(y), cpuref.crtinv(R, y))
        assert np.array_equal(P.polymul(y, z), cpuref.polymul(R, y, z))
        assert np.array_equal(P.polymul(y, y), cpuref.polymul(R, y, y))       # squaring path (a is b)


@pytest.mark.parametrize("L", range(4, 15))
def test_pow2_16_byte_and_8_byte_global_access(gpu, cpuref, L):
    """Single-modulus launches move 16 bytes per lane when the slabs are 16-byte aligned (k_pow2<..., T1>);
    8-byte-aligned slabs and the forced switch take the 8-byte kernels.  All agree with the oracle in every
    arithmetic class, incl. reference-style negative representatives and a ragged last workgroup."""
    torch = pytest.importorskip("torch")
    m = 2 ** (L + 1)
    for lower in (2 ** 20, 2 ** 29, 2 ** 30, 2 ** 58, 2 ** 61):
        q = lm.first_good_q(m, lower)
        P, R = gpu.Plan([(2, L + 1)], [q]), Params([(2, L + 1)], [q])
        rng = np.random.default_rng(L + lower % 97)
        B = 7
        y, z = R.random(rng, B), R.random(rng, B)
        y[0] = np.where(y[0] > 0, y[0] - q, 0)                        # (-q, 0] representatives
        want = cpuref.crt(R, y), cpuref.crtinv(R, y), cpuref.polymul(R, y, z), cpuref.mul(R, y, z)
        for forced in (False, True):
            gpu.debug_set("NO_T1", forced)
            assert np.array_equal(P.crt(y), want[0]), (L, q, forced)
            assert np.array_equal(P.crtInv(y), want[1]), (L, q, forced)
            assert np.array_equal(P.polymul(y, z), want[2]), (L, q, forced)
        gpu.debug_set("NO_T1", False)
        if L in (12, 13) and q < 2 ** 31:
            # the persistent DMA-pipelined poly-mul (pow2_pipe.hip), forced at a small batch: fewer polynomials
            # than workgroups, a ragged last round, squaring, c aliasing a
            gpu.debug_set("FORCE_PIPE", True)
            assert np.array_equal(P.polymul(y, z), want[2]), (L, q, "pipe")
            assert np.array_equal(P.polymul(y, y), cpuref.polymul(R, y, y)), (L, q, "pipe square")
            Bb = 1100 if L == 12 else 600                                # > one round of resident workgroups
            yb, zb = R.random(rng, Bb), R.random(rng, Bb)
            yb[3] = np.where(yb[3] > 0, yb[3] - q, 0)
            db_, dz_ = torch.from_numpy(yb).cuda(), torch.from_numpy(zb).cuda()
            dout = torch.empty_like(db_)
            P.polymul(db_, dz_, out=dout)
            gpu.debug_set("FORCE_PIPE", False)
            gpu.debug_set("NO_PIPE", True)
            dref = torch.empty_like(db_)
            P.polymul(db_, dz_, out=dref)
            gpu.debug_set("NO_PIPE", False)
            assert torch.equal(dout, dref), (L, q, "pipe vs one-polynomial-per-workgroup kernel")
            rows = [0, 3, 255, 256, 511, 512, 513, Bb - 1]
            assert np.array_equal(dout[rows].cpu().numpy(), cpuref.polymul(R, yb[rows], zb[rows])), (L, q, "pipe rows")
            gpu.debug_set("FORCE_PIPE", True)
            P.polymul(db_, dz_, out=db_)                                  # in place
            gpu.debug_set("FORCE_PIPE", False)
            assert torch.equal(db_, dref), (L, q, "pipe in place")
        # slabs that are only 8-byte aligned: one int64 into a 16-byte-aligned allocation
        n = R.n
        pad_a = torch.zeros(B * n + 1, dtype=torch.int64, device="cuda")
        pad_b = torch.zeros(B * n + 1, dtype=torch.int64, device="cuda")
        pad_c = torch.zeros(B * n + 1, dtype=torch.int64, device="cuda")
        da, db, dc = (t[1:].view(B, n, 1) for t in (pad_a, pad_b, pad_c))
        assert da.data_ptr() % 16 == 8
        da.copy_(torch.from_numpy(y)); db.copy_(torch.from_numpy(z))
        P.polymul(da, db, out=dc)
        assert np.array_equal(dc.cpu().numpy(), want[2]), (L, q, "misaligned polymul")
        x = da.clone()                                                 # clone: 16-byte aligned again
        P.mul(x, db)                                                   # one aligned, one not
        assert np.array_equal(x.cpu().numpy(), want[3]), (L, q, "mul mixed alignment")
        P.crt(da); assert np.array_equal(da.cpu().numpy(), want[0]), (L, q, "misaligned crt")
        da.copy_(torch.from_numpy(y)); P.crtInv(da)
        assert np.array_equal(da.cpu().numpy(), want[1]), (L, q, "misaligned crtInv")
        assert int(pad_a[0]) == 0 and int(pad_c[0]) == 0              # nothing written in front of the slab


def test_copy_slab_yardstick(gpu):
    torch = pytest.importorskip("torch")
    src = torch.randint(0, 2 ** 62, (1 << 20,), dtype=torch.int64, device="cuda")
    for variant in (0, 1):
        for cnt in (1 << 20, 1026, 2):
            dst = torch.zeros_like(src)
            assert gpu.lib().lolhip_copy_slab(None, dst.data_ptr(), src.data_ptr(), cnt * 8, variant) == 0
            torch.cuda.synchronize()
            assert torch.equal(dst[:cnt], src[:cnt]) and int(dst[cnt:].abs().sum()) == 0
    assert gpu.lib().lolhip_copy_slab(None, src.data_ptr() + 8, src.data_ptr(), 16, 0) != 0    # misaligned: refused


@pytest.mark.parametrize("m", [3, 5, 9, 15, 21, 25, 27, 33, 36, 45, 49, 63, 89, 105, 121, 280, 432, 1155, 1728, 5184, 14400, 15015])
def test_generic_indices(gpu, cpuref, m):
    pps = lm.factor_pps(m)
    for T, lower in ((1, 2 ** 60), (2, 2 ** 30)):
        g = lm.good_qs(m, lower)
        qs = [next(g) for _ in range(T)]
        P, R = gpu.Plan(pps, qs), Params(pps, qs)
        rng = np.random.default_rng(m)
        B = 3 if R.n <= 2000 else 1
        y, z = R.random(rng, B), R.random(rng, B)
        for op in ("crt", "crtinv") + PRIME_OPS:
            got, want = getattr(P, PLAN_NAME[op])(y), getattr(cpuref, op)(R, y)
            assert (got is None) == (want is None), op
            if want is not None:
                assert np.array_equal(got, want), (op, m, qs)
        assert np.array_equal(P.mul(y, z), cpuref.mul(R, y, z))
        assert np.array_equal(P.polymul(y, z), cpuref.polymul(R, y, z))
        # mulGCRT = crt . mulGPow . crtInv ; divGCRT its inverse (TensorTests.hs:107-112)
        assert np.array_equal(P.mulGCRT(y), cpuref.crt(R, cpuref.gpow(R, cpuref.crtinv(R, y))))
        assert np.array_equal(P.divGCRT(P.mulGCRT(y)), y)


@pytest.mark.parametrize("m", [9, 25, 27, 45, 225, 675, 1575, 1728, 3200, 14400, 192, 384, 3072, 6144, 11648, 15, 105, 1155, 15015, 21, 273, 81, 243, 5184, 405])
def test_merged_prime_powers(gpu, cpuref, m):
    """Class 2 of the vector interpreter (32-bit residues, one 64-bit accumulator per dot product): the plan
    replaces CRT_{p^e} for 3^2, 5^2, 3^3 by ONE dense phi x phi stage and the adjacent factors 3 (x) 5, 3 (x) 7 by one Kronecker
    stage (plan.cpp merge_stages).  Same linear maps, so the same residues: against the
    oracle and against a plan built with it turned off (NO_MERGE), at a modulus that takes the 20-term
    accumulator (q < 2^29.68), one that only takes 13 terms (3^2 merges, 5^2 and 3^3 stay staged) and two moduli
    per plan; crt, crtInv, poly-mul, squaring, ragged batch."""
    pps = lm.factor_pps(m)
    for qs in ([lm.first_good_q(m, 2 ** 20)], [lm.first_good_q(m, 2 ** 29)], [lm.first_good_q(m, 2 ** 30)],
               [lm.first_good_q(m, 2 ** 26), lm.first_good_q(m, 2 ** 29 + 2 ** 28)]):
        R = Params(pps, qs)
        rng = np.random.default_rng(m + len(qs))
        B = 5 if R.n <= 2000 else 2
        y, z = R.random(rng, B), R.random(rng, B)
        want = cpuref.crt(R, y), cpuref.crtinv(R, y), cpuref.polymul(R, y, z), cpuref.polymul(R, y, y)
        for merged in (True, False):
            gpu.debug_set("NO_MERGE", not merged)
            P = gpu.Plan(pps, qs)
            gpu.debug_set("NO_MERGE", False)
            assert np.array_equal(P.crt(y), want[0]), (m, qs, merged)
            assert np.array_equal(P.crtInv(y), want[1]), (m, qs, merged)
            assert np.array_equal(P.polymul(y, z), want[2]), (m, qs, merged)
            assert np.array_equal(P.polymul(y, y), want[3]), (m, qs, merged)
            assert np.array_equal(P.crtInv(P.crt(y)), y)


@pytest.mark.parametrize("m", [13, 25, 27, 225, 1575, 15015, 14400, 11648])
def test_lazy_dense_stages_at_the_top_of_their_range(gpu, cpuref, m):
    """Class 4 of the vector interpreter (every modulus below 2^27): dense stages hand values in [0,2q) to each other
    and reduce with the bare 3-instruction Montgomery step, whose bound (T / 2^32 < q for 16 terms of values below 2q)
    is tightest for the largest moduli and the largest residues.  Moduli just below 2^27, inputs of q - 1 everywhere,
    alternating 0 / q - 1, negative representatives and random ones; against the oracle and against the same plan built
    as class 2 (NO_LAZY); crt, crtInv, poly-mul, the G and L maps (which must still see canonical residues)."""
    pps = lm.factor_pps(m)
    g = lm.good_qs(m, 2 ** 27 - 2 ** 22)
    qs = [q for q in (next(g) for _ in range(40)) if q < 2 ** 27][-2:]
    assert qs and all(q < 2 ** 27 for q in qs)
    for qsel in ([qs[-1]], qs):
        R = Params(pps, qsel)
        rng = np.random.default_rng(m)
        B = 4 if R.n <= 2000 else 2
        y = R.random(rng, B)
        qv = np.array(qsel, dtype=np.int64)
        y[0] = qv - 1                                    # every coefficient q - 1
        y[1, ::2] = 0; y[1, 1::2] = qv - 1
        if B > 2:
            y[2] = -(qv - 1)                             # negative representatives (-q, 0]
        z = R.random(rng, B); z[0] = qv - 1
        want = {op: getattr(cpuref, op)(R, y) for op in ("crt", "crtinv") + PRIME_OPS}
        wmul, wsq = cpuref.polymul(R, y, z), cpuref.polymul(R, y, y)
        for lazy in (True, False):
            gpu.debug_set("NO_LAZY", not lazy)
            P = gpu.Plan(pps, qsel)
            gpu.debug_set("NO_LAZY", False)
            for op in ("crt", "crtinv") + PRIME_OPS:
                got = getattr(P, PLAN_NAME[op])(y)
                assert (got is None) == (want[op] is None), op
                if got is not None:
                    assert np.array_equal(got, want[op]), (op, m, qsel, lazy)
            assert np.array_equal(P.polymul(y, z), wmul), (m, qsel, lazy)
            assert np.array_equal(P.polymul(y, y), wsq), (m, qsel, lazy)


@pytest.mark.parametrize("m", [12, 40, 48, 96, 160, 768, 1728, 2912, 11648, 14336, 2 ** 12 * 3])
def test_two_power_factor_routes(gpu, cpuref, monkeypatch, m):
    """m = 2^e * odd: the innermost tensor factor CRT_{2^e} acts on contiguous blocks of 2^(e-1)
    coefficients.  Three routes against the oracle, every arithmetic class, ragged batch:
    "fused" (default, e >= 2): register tiles of up to four levels inside the vector interpreter, one
    launch, fused poly-mul; "split" (e >= 5): the m = 2^k kernels for the 2-power factor and the stage
    program for the odd primes; "stages": the reference's own factorisation, stage by stage.
    The m cover one to three tiles per transform with 1, 2, 3 and 4 levels in the last one."""
    pps = lm.factor_pps(m)
    for qs in ([lm.first_good_q(m, 2 ** 20)], [lm.first_good_q(m, 2 ** 29), lm.first_good_q(m, 2 ** 31)],
               [lm.first_good_q(m, 2 ** 45)], [lm.first_good_q(m, 2 ** 60), lm.first_good_q(m, 2 ** 61)]):
        R = Params(pps, qs)
        rng = np.random.default_rng(m + len(qs))
        y, z = R.random(rng, 5), R.random(rng, 5)
        want = cpuref.crt(R, y), cpuref.crtinv(R, y), cpuref.polymul(R, y, z)
        for route in ("fused", "split", "stages"):
            gpu.debug_set("NO_FUSED2", route != "fused")
            gpu.debug_set("NO_POW2_PART", route == "stages")
            P = gpu.Plan(pps, qs)
            assert np.array_equal(P.crt(y), want[0]), (m, qs, route)
            assert np.array_equal(P.crtInv(y), want[1]), (m, qs, route)
            assert np.array_equal(P.polymul(y, z), want[2]), (m, qs, route)
            assert np.array_equal(P.polymul(y, y), cpuref.polymul(R, y, y)), (m, qs, route)
            gpu.debug_set("NO_POW2_PART", False)
            gpu.debug_set("NO_FUSED2", False)


@pytest.mark.parametrize("m,q", BENCH1)
def test_reference_benchmark_parameters(gpu, cpuref, m, q):
    """lol/Crypto/Lol/Benchmarks/Default.hs:42-46"""
    pps = lm.factor_pps(m)
    P, R = gpu.Plan(pps, [q]), Params(pps, [q])
    y = R.random(np.random.default_rng(q), 4)
    assert np.array_equal(P.crt(y), cpuref.crt(R, y))
    assert np.array_equal(P.crtInv(P.crt(y)), y)
    assert np.array_equal(P.lInv(P.l(y)), y)
    assert np.array_equal(P.divGPow(P.mulGPow(y)), y)
    assert np.array_equal(P.divGDec(P.mulGDec(y)), y)


@pytest.mark.parametrize("m,m2,q", BENCH2 + [(12, 60, 61), (9, 45, 181), (1, 8, 17), (8, 8, 17), (56, 2912, 8737)])
def test_twace_embed_vs_oracle(gpu, cpuref, m, m2, q):
    a, b = lm.factor_pps(m), lm.factor_pps(m2)
    qs = [q, lm.first_good_q(m2, q)]
    Pl, Ph = gpu.Plan(a, qs), gpu.Plan(b, qs)
    X = gpu.Ext(Pl, Ph)
    Rl, Rh = Params(a, qs), Params(b, qs)
    rng = np.random.default_rng(m2)
    lo, hi = Rl.random(rng, 2), Rh.random(rng, 2)
    assert np.array_equal(X.embedPow(lo), cpuref.embed_pow(Rl, Rh, lo))
    assert np.array_equal(X.embedDec(lo), cpuref.embed_dec(Rl, Rh, lo))
    assert np.array_equal(X.embedCRT(lo), cpuref.embed_crt(Rl, Rh, lo))
    assert np.array_equal(X.twacePowDec(hi), cpuref.twace_powdec(Rl, Rh, hi))
    assert np.array_equal(X.twaceCRT(hi), cpuref.twace_crt(Rl, Rh, hi))
    # the reference's identities (TensorTests.hs:133-234), with the GPU on both sides
    assert np.array_equal(X.twacePowDec(X.embedPow(lo)), lo)
    assert np.array_equal(X.twaceCRT(X.embedCRT(lo)), lo)
    assert np.array_equal(X.embedCRT(lo), Ph.crt(X.embedPow(Pl.crtInv(lo))))
    assert np.array_equal(X.twaceCRT(hi), Pl.crt(X.twacePowDec(Ph.crtInv(hi))))
    assert np.array_equal(X.embedDec(lo), Ph.lInv(X.embedPow(Pl.l(lo))))
    # coeffs (Tensor.hs:174) and prop_coeffsBasis (CycTests.hs:71-76) with the GPU on both sides
    cs = X.coeffs(hi)
    assert np.array_equal(cs, cpuref.coeffs(Rl, Rh, hi))
    assert np.array_equal(cs[0], X.twacePowDec(hi))                       # vector 0 is the twace gather
    if Rh.n <= 4096:
        acc = np.zeros_like(hi)
        tab = X.table(5).reshape(-1, Rl.n)
        for i1 in range(tab.shape[0]):
            basis = np.zeros_like(hi); basis[:, tab[i1, 0], :] = 1
            acc = (acc + Ph.polymul(X.embedPow(cs[i1]), basis)) % np.array(qs)
        assert np.array_equal(acc, hi)


@pytest.mark.parametrize("T", [1, 2, 3, 4])
def test_streaming_kernels_layouts(gpu, cpuref, T):
    """mulRq / mulG*CRT / twace* / embed* on device slabs: every tupSize path (16-byte chunks for even T,
    the magic-division path for odd T), slabs that are only 8-byte aligned, a batch that does not fill the
    last tile, a source polynomial too large for the LDS-staged gather, and a g vector shorter than a tile."""
    import torch
    for m, m2, B in ((8, 56, 5), (64, 64 * 7 * 3, 3), (2 ** 13, 2 ** 13 * 3, 2), (3, 3, 4)):
        a, b = lm.factor_pps(m), lm.factor_pps(m2)
        qs, lo_q = [], 2 ** 20
        for _ in range(T):
            lo_q = lm.first_good_q(m2, lo_q); qs.append(lo_q)
        Pl, Ph = gpu.Plan(a, qs), gpu.Plan(b, qs)
        X = gpu.Ext(Pl, Ph)
        Rl, Rh = Params(a, qs), Params(b, qs)
        rng = np.random.default_rng(m2 + T)
        lo, hi = Rl.random(rng, B), Rh.random(rng, B)

        def dev(x, shift):                              # a device copy whose base address is 8 mod 16 when shift = 1
            buf = torch.empty(x.size + 2, dtype=torch.int64, device="cuda")
            off = (1 if (buf.data_ptr() % 16 == 0) else 0) if shift else (0 if (buf.data_ptr() % 16 == 0) else 1)
            v = buf[off:off + x.size].view(x.shape)
            v.copy_(torch.from_numpy(x))
            return v
        for shift in (0, 1):
            dl, dh = dev(lo, shift), dev(hi, shift)
            assert np.array_equal(X.embedPow(dl).cpu().numpy(), cpuref.embed_pow(Rl, Rh, lo)), (m, m2, shift)
            assert np.array_equal(X.embedDec(dl).cpu().numpy(), cpuref.embed_dec(Rl, Rh, lo)), (m, m2, shift)
            assert np.array_equal(X.embedCRT(dl).cpu().numpy(), cpuref.embed_crt(Rl, Rh, lo)), (m, m2, shift)
            assert np.array_equal(X.twacePowDec(dh).cpu().numpy(), cpuref.twace_powdec(Rl, Rh, hi)), (m, m2, shift)
            assert np.array_equal(X.twaceCRT(dh).cpu().numpy(), cpuref.twace_crt(Rl, Rh, hi)), (m, m2, shift)
            out = dev(np.zeros_like(hi), shift)
            X.embedCRT(dl, out=out)
            assert np.array_equal(out.cpu().numpy(), cpuref.embed_crt(Rl, Rh, lo)), (m, m2, shift)
            h2 = dev(Rh.random(rng, B), shift)
            assert np.array_equal(Ph.mul(dh.clone(), h2).cpu().numpy(), cpuref.mul(Rh, hi, h2.cpu().numpy())), (m, m2, shift)
            assert np.array_equal(Ph.mulGCRT(dh.clone()).cpu().numpy(), cpuref.crt(Rh, cpuref.gpow(Rh, cpuref.crtinv(Rh, hi)))), (m, m2, shift)
            assert np.array_equal(Pl.mulGCRT(dl.clone()).cpu().numpy(), cpuref.crt(Rl, cpuref.gpow(Rl, cpuref.crtinv(Rl, lo)))), (m, m2, shift)   # TensorTests.hs:107-112


# ---------------------------------------------------------------------------------------
# edge cases
# ---------------------------------------------------------------------------------------

def test_empty_and_ragged_batches(gpu, cpuref):
    for m in (64, 512, 45):           # packed pow2 launch (several polynomials per workgroup), generic
        pps = lm.factor_pps(m)
        q = lm.first_good_q(m, 2 ** 40)
        P, R = gpu.Plan(pps, [q]), Params(pps, [q])
        empty = np.zeros((0, R.n, 1), dtype=np.int64)
        assert P.crt(empty).shape == (0, R.n, 1)
        assert P.polymul(empty, empty).shape == (0, R.n, 1)
        rng = np.random.default_rng(m)
        for B in (1, 2, 3, 7, 13, 33):
            y, z = R.random(rng, B), R.random(rng, B)
            assert np.array_equal(P.crt(y), cpuref.crt(R, y)), (m, B)
            assert np.array_equal(P.crtInv(y), cpuref.crtinv(R, y)), (m, B)
            assert np.array_equal(P.polymul(y, z), cpuref.polymul(R, y, z)), (m, B)


def test_negative_representatives(gpu, cpuref):
    """inputs in (-q, q) as the reference's Zq allows (types.h:52-57); outputs canonical"""
    for m in (21, 256, 2 ** 12):
        pps = lm.factor_pps(m)
        q = lm.first_good_q(m, 2 ** 45)
        P, R = gpu.Plan(pps, [q]), Params(pps, [q])
        y = R.random(np.random.default_rng(1), 2) - q // 2
        z = R.random(np.random.default_rng(2), 2) - q // 3
        for op in ("crt", "crtinv") + PRIME_OPS:
            got = getattr(P, PLAN_NAME[op])(y)
            assert np.array_equal(got, getattr(cpuref, op)(R, y)), op
            assert got.min() >= 0 and got.max() < q
        assert np.array_equal(P.mul(y, z), cpuref.mul(R, y, z))
        assert np.array_equal(P.polymul(y, z), cpuref.polymul(R, y, z))


def test_extreme_values_and_moduli(gpu, cpuref):
    """all-zero, all-(q-1), and moduli at both ends of the supported range"""
    for m, lower in ((2 ** 10, 2 ** 62 - 2 ** 30), (2 ** 10, 2 ** 61 - 2 ** 20), (2 ** 10, 2 ** 61 + 5), (16, 16), (45, 2 ** 61)):
        pps = lm.factor_pps(m)
        q = lm.first_good_q(m, lower)
        if q >= 2 ** 62:
            continue
        P, R = gpu.Plan(pps, [q]), Params(pps, [q])
        for fill in (0, q - 1, 1):
            y = np.full((2, R.n, 1), fill, dtype=np.int64)
            assert np.array_equal(P.crt(y), cpuref.crt(R, y))
            assert np.array_equal(P.crtInv(y), cpuref.crtinv(R, y))
            assert np.array_equal(P.polymul(y, y), cpuref.polymul(R, y, y))
    with pytest.raises(gpu.LolHipError):
        gpu.Plan([(2, 4)], [2 ** 62 + 1])


@pytest.mark.parametrize("m", [2 ** 5, 2 ** 10, 2 ** 12, 2 ** 14])
def test_arithmetic_class_boundaries(gpu, cpuref, m):
    """The m = 2^k path picks its arithmetic per plan: 32-bit residues when every modulus is
    below 2^27 (no conditional subtraction in the forward transform: values grow to 29 q), below
    2^30 (lazy ranges) or below 2^31 (tighter ranges), 64-bit lazy Shoup below 2^61, exact above.
    Moduli hugging each boundary, the extreme residues, and tuples that mix classes (the widest
    class must win)."""
    pps = lm.factor_pps(m)

    def last_good_below(bound):
        q = (bound - 2) // m * m + 1
        while not lm.is_prime(q):
            q -= m
        return q

    q_27 = last_good_below(2 ** 27)            # largest modulus of the subtraction-free class
    q_27h = lm.first_good_q(m, 2 ** 27)        # smallest modulus of the lazy 32-bit class
    assert q_27 < 2 ** 27 < q_27h
    q_lo = last_good_below(2 ** 30)            # largest modulus of the lazy 32-bit class
    q_mid = lm.first_good_q(m, 2 ** 30)        # smallest modulus of the second 32-bit class
    q_31 = last_good_below(2 ** 31)            # largest 32-bit modulus
    q_31h = lm.first_good_q(m, 2 ** 31)        # smallest modulus that needs 64-bit residues
    q_hi = last_good_below(2 ** 61)            # largest lazy-class modulus
    q_top = lm.first_good_q(m, 2 ** 61)        # smallest exact-class modulus
    assert q_lo < 2 ** 30 < q_mid < q_31 < 2 ** 31 < q_31h and q_hi < 2 ** 61 < q_top
    rng = np.random.default_rng(m)
    for qs in ([q_27], [q_27h], [q_27, 12289 if m <= 4096 else 65537], [q_27, q_27h],
               [q_lo], [q_mid], [q_31], [q_31h], [q_hi], [q_top], [q_lo, 12289 if m <= 4096 else 65537],
               [q_lo, q_mid], [q_31, q_lo, q_mid], [q_31, q_31h], [q_mid, q_hi, q_lo], [q_top, q_lo]):
        P, R = gpu.Plan(pps, qs), Params(pps, qs)
        y, z = R.random(rng, 3), R.random(rng, 3)
        for t, q in enumerate(qs):                 # extreme residues in two of the polynomials
            y[0, :, t] = q - 1
            z[1, :, t] = q - 1
        y[2, ::2] = 0
        assert np.array_equal(P.crt(y), cpuref.crt(R, y)), qs
        assert np.array_equal(P.crtInv(y), cpuref.crtinv(R, y)), qs
        assert np.array_equal(P.polymul(y, z), cpuref.polymul(R, y, z)), qs
        assert np.array_equal(P.polymul(y, y), cpuref.polymul(R, y, y)), qs
        neg = np.where(y > 0, y - np.asarray(qs, dtype=np.int64), 0)   # representatives in (-q, 0]
        assert np.array_equal(P.crt(neg), cpuref.crt(R, y)), qs


def test_divg_failure_is_reported(gpu):
    """oddRad(m) not invertible mod q -> Nothing in the reference (g.cpp:194-199, CPP.hs:321-323)"""
    P = gpu.Plan([(3, 1), (7, 1)], [21])
    y = np.arange(12, dtype=np.int64).reshape(1, 12, 1)
    assert P.divGPow(y) is None and P.divGDec(y) is None
    assert P.mulGPow(y) is not None


def test_large_generic_polynomial_uses_scratch_path(gpu, cpuref):
    """n = 16384 with a non power-of-two index: stage program runs out of HBM scratch"""
    m = 3 * 2 ** 14
    pps = lm.factor_pps(m)
    q = lm.first_good_q(m, 2 ** 50)
    P, R = gpu.Plan(pps, [q]), Params(pps, [q])
    assert R.n == 16384
    y = R.random(np.random.default_rng(4), 2)
    assert np.array_equal(P.crt(y), cpuref.crt(R, y))
    assert np.array_equal(P.crtInv(y), cpuref.crtinv(R, y))
    assert np.array_equal(P.l(y), cpuref.l(R, y))
    assert np.array_equal(P.divGDec(y), cpuref.ginvdec(R, y))


# ---------------------------------------------------------------------------------------
# device-pointer API (what bench.py times) + aliasing rules of polymul
# ---------------------------------------------------------------------------------------

def test_device_tensors_and_aliasing(gpu, cpuref):
    torch = pytest.importorskip("torch")
    for m in (2 ** 12, 2 ** 14, 45, 96, 89):      # fused kernel x2, vector interpreter, 2-power split, scalar interpreter
        pps = lm.factor_pps(m)
        g = lm.good_qs(m, 2 ** 58)
        qs = [next(g), next(g)]
        P, R = gpu.Plan(pps, qs), Params(pps, qs)
        rng = np.random.default_rng(m)
        a, b = R.random(rng, 6), R.random(rng, 6)
        want = cpuref.polymul(R, a, b)
        da, db = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
        dc = torch.empty_like(da)
        P.polymul(da, db, out=dc)
        assert np.array_equal(dc.cpu().numpy(), want)
        assert np.array_equal(da.cpu().numpy(), a) and np.array_equal(db.cpu().numpy(), b)   # inputs untouched
        x = da.clone(); P.polymul(x, db, out=x)                                            # c aliases a
        assert np.array_equal(x.cpu().numpy(), want)
        x = db.clone(); P.polymul(da, x, out=x)                                            # c aliases b
        assert np.array_equal(x.cpu().numpy(), want)
        x = da.clone(); P.polymul(x, x, out=x)                                             # in-place square
        assert np.array_equal(x.cpu().numpy(), cpuref.polymul(R, a, a))
        x = da.clone(); P.crt(x); P.crtInv(x)
        assert np.array_equal(x.cpu().numpy(), a)
        x = da.clone(); P.mulGPow(x); P.divGPow(x)
        assert np.array_equal(x.cpu().numpy(), a)
        x = da.clone(); P.mul(x, db)
        assert np.array_equal(x.cpu().numpy(), cpuref.mul(R, a, b))
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            x = da.clone(); P.crt(x, stream=s.cuda_stream); s.synchronize()
        assert np.array_equal(x.cpu().numpy(), cpuref.crt(R, a))


# ---------------------------------------------------------------------------------------
# (c) BASELINE.json full sizes: properties on the whole batch + strided samples vs the oracle
# ---------------------------------------------------------------------------------------

def _sample(cpuref_fn, got, B, k=3):
    idx = sorted(set([0, B // 2, B - 1]))[:k]
    return idx


@pytest.mark.parametrize("lower,route", [(2 ** 60, "default"), (2 ** 26, "default"), (2 ** 29, "default"), (2 ** 30, "default"),
                                         (2 ** 29, "NO_PIPE"), (2 ** 30, "NO_T1")])
def test_config2_full_batch(gpu, cpuref, lower, route):
    """m = 2^14, batch 4096, at the 61-bit modulus of the metric and at one modulus of every 32-bit class
    (q = 1073872897 > 2^30 is the one where lol-cpp's CT itself is valid): crtInv.crt = id over the batch;
    poly-mul rows at both sides of every dispatch-round boundary (1024 workgroups resident at a time; the
    persistent kernel's 512) against the oracle; ring identities (commutativity, distributivity, unit)."""
    torch = pytest.importorskip("torch")
    m = 2 ** 14
    q = lm.first_good_q(m, lower)
    P, R = gpu.Plan([(2, 14)], [q]), Params([(2, 14)], [q])
    B, n = 4096, R.n
    if route != "default":
        gpu.debug_set(route, True)
    g = torch.Generator(device="cuda"); g.manual_seed(2)
    a = torch.randint(0, q, (B, n, 1), dtype=torch.int64, device="cuda", generator=g)
    b = torch.randint(0, q, (B, n, 1), dtype=torch.int64, device="cuda", generator=g)
    x = a.clone(); P.crt(x); P.crtInv(x)
    assert torch.equal(x, a)
    c = torch.empty_like(a); P.polymul(a, b, out=c)
    c2 = torch.empty_like(a); P.polymul(b, a, out=c2)
    assert torch.equal(c, c2)                                              # commutative
    idx = [0, 1, 511, 512, 1023, 1024, B // 3, 2047, 2048, 3071, 3072, 3583, 3584, B - 1]
    want = cpuref.polymul(R, a[idx].cpu().numpy(), b[idx].cpu().numpy())
    assert np.array_equal(c[idx].cpu().numpy(), want)
    # (a + b) * b == a*b + b*b  (distributive), all mod q
    s = (a + b) % q
    l1 = torch.empty_like(a); P.polymul(s, b, out=l1)
    bb = torch.empty_like(a); P.polymul(b, b, out=bb)
    assert torch.equal(l1, (c + bb) % q)
    one = torch.zeros_like(a); one[:, 0, :] = 1                              # scalarPow 1 (CPP.hs:413-417)
    u = torch.empty_like(a); P.polymul(a, one, out=u)
    assert torch.equal(u, a)
    # in place (c aliases a), and the 61-bit and 32-bit kernels agree on the batch wherever both apply
    x = a.clone(); P.polymul(x, b, out=x)
    assert torch.equal(x, c)


def test_short_lived_host_threads_leave_nothing_behind(gpu, cpuref):
    """Host-pointer calls from many short-lived OS threads (a GHC safe-FFI worker pool, a thread pool): staging sets
    are leased from a process-wide pool per call, so HBM in use after 48 such threads equals HBM in use after the
    first, every result is the oracle's, and lolhip_thread_release gives the pooled memory back."""
    import threading
    torch = pytest.importorskip("torch")
    pps, qs = [(2, 12)], [lm.first_good_q(2 ** 12, 2 ** 40)]
    P, R = gpu.Plan(pps, qs), Params(pps, qs)
    rng = np.random.default_rng(5)
    y = R.random(rng, 1024)                       # 16 MiB per operand: a leaked staging set per thread would show in mem_get_info
    want = cpuref.crt(R, y)
    gpu.lib().lolhip_thread_release()
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    bad = []

    def work():
        try:
            if not np.array_equal(P.crt(y), want):
                bad.append("mismatch")
        except Exception as e:                     # noqa: BLE001
            bad.append(repr(e))

    t = threading.Thread(target=work); t.start(); t.join()
    free1 = torch.cuda.mem_get_info()[0]
    assert free1 <= free0                           # the pool keeps the first call's buffers
    for _ in range(48):
        t = threading.Thread(target=work); t.start(); t.join()
    assert not bad, bad[:3]
    assert torch.cuda.mem_get_info()[0] == free1    # ... and 48 more threads added nothing
    # concurrent calls each get their own set; afterwards at most that many are pooled
    ts = [threading.Thread(target=work) for _ in range(6)]
    [t.start() for t in ts]; [t.join() for t in ts]
    assert not bad, bad[:3]
    # releasing the idle sets and starting over lands on the same footprint: nothing accumulates across cycles
    # (absolute comparisons with free0 are not meaningful: the HIP runtime keeps a few MiB per stream it has created)
    gpu.lib().lolhip_thread_release()
    t = threading.Thread(target=work); t.start(); t.join()
    free3 = torch.cuda.mem_get_info()[0]
    for _ in range(16):
        t = threading.Thread(target=work); t.start(); t.join()
    assert not bad, bad[:3]
    assert torch.cuda.mem_get_info()[0] == free3


def test_concurrent_host_threads(gpu, cpuref):
    """The reference is non-reentrant (process-global modulus, types.h:59).  Here four host
    threads drive their own plans (different rings and moduli, one of them the mixed-radix path
    with its plan-owned work buffer) on their own streams at the same time, and two more share
    ONE plan through the drop-in-style host calls; every result must be the oracle's."""
    import threading
    torch = pytest.importorskip("torch")
    jobs = [(2 ** 12, 2 ** 60), (2 ** 10, 2 ** 29), (45, 2 ** 58), (2 ** 8 * 3, 2 ** 30)]
    errors = []

    def own_plan(m, lower, seed):
        try:
            pps = lm.factor_pps(m)
            qs = [lm.first_good_q(m, lower), lm.first_good_q(m, lower * 2)]
            P, R = gpu.Plan(pps, qs), Params(pps, qs)
            rng = np.random.default_rng(seed)
            st = torch.cuda.Stream()
            for _ in range(6):
                a, b = R.random(rng, 7), R.random(rng, 7)
                da, db = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
                dc = torch.empty_like(da)
                with torch.cuda.stream(st):
                    P.polymul(da, db, out=dc, stream=st.cuda_stream)
                    P.crt(da, stream=st.cuda_stream)
                st.synchronize()
                if not np.array_equal(dc.cpu().numpy(), cpuref.polymul(R, a, b).reshape(a.shape)):
                    errors.append(("polymul", m))
                if not np.array_equal(da.cpu().numpy(), cpuref.crt(R, a).reshape(a.shape)):
                    errors.append(("crt", m))
        except Exception as ex:      # noqa: BLE001
            errors.append((type(ex).__name__, str(ex), m))

    shared_pps, shared_qs = [(2, 9)], [lm.first_good_q(512, 2 ** 40)]
    PS, RS = gpu.Plan(shared_pps, shared_qs), Params(shared_pps, shared_qs)

    def shared_plan(seed):
        try:
            rng = np.random.default_rng(seed)
            for _ in range(6):
                a = RS.random(rng, 3)
                if not np.array_equal(PS.crtInv(PS.crt(a)), a):      # numpy in/out: host round trips
                    errors.append(("shared", seed))
        except Exception as ex:      # noqa: BLE001
            errors.append((type(ex).__name__, str(ex)))

    # two threads share ONE mixed-radix plan (m = 45) on two streams: its unfused poly-mul needs a
    # per-call work buffer, which used to live in the plan (a race: use after free, or each
    # other's operand).  Batch sizes differ so a shared buffer would also have to grow.
    mix_pps, mix_qs = lm.factor_pps(45), [lm.first_good_q(45, 2 ** 40), lm.first_good_q(45, 2 ** 29)]
    PM, RM = gpu.Plan(mix_pps, mix_qs), Params(mix_pps, mix_qs)

    def shared_mixed(seed):
        try:
            rng = np.random.default_rng(seed)
            st = torch.cuda.Stream()
            for it in range(8):
                B = 5 + 37 * ((seed + it) % 3)
                a, b = RM.random(rng, B), RM.random(rng, B)
                da, db = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
                dc = torch.empty_like(da)
                with torch.cuda.stream(st):
                    PM.polymul(da, db, out=dc, stream=st.cuda_stream)
                st.synchronize()
                if not np.array_equal(dc.cpu().numpy(), cpuref.polymul(RM, a, b).reshape(a.shape)):
                    errors.append(("shared mixed-radix polymul", seed, it))
        except Exception as ex:      # noqa: BLE001
            errors.append((type(ex).__name__, str(ex), "shared mixed"))

    threads = [threading.Thread(target=own_plan, args=(m, lo, i)) for i, (m, lo) in enumerate(jobs)]
    threads += [threading.Thread(target=shared_plan, args=(100 + i,)) for i in range(2)]
    threads += [threading.Thread(target=shared_mixed, args=(200 + i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_slabs_larger_than_4GiB(gpu, cpuref):
    """70,000 polynomials of n = 8192: 4.6 GB per operand, past every 32-bit byte offset.  The
    m = 2^k kernels address through per-workgroup buffer windows, the streaming kernels with
    64-bit indices; rows from both ends and across the 4 GiB line against the oracle."""
    torch = pytest.importorskip("torch")
    free, _ = torch.cuda.mem_get_info()
    if free < 24e9:
        pytest.skip("needs 24 GB of free HBM")
    m = 2 ** 14
    q = lm.first_good_q(m, 2 ** 60)
    P, R = gpu.Plan([(2, 14)], [q]), Params([(2, 14)], [q])
    B, n = 70000, R.n
    g = torch.Generator(device="cuda"); g.manual_seed(9)
    a = torch.randint(0, q, (B, n, 1), dtype=torch.int64, device="cuda", generator=g)
    b = torch.randint(0, q, (B, n, 1), dtype=torch.int64, device="cuda", generator=g)
    c = torch.empty_like(a)
    P.polymul(a, b, out=c)
    line = (1 << 32) // (n * 8)                                   # first polynomial past 4 GiB
    idx = [0, 1, line - 1, line, line + 1, B - 2, B - 1]
    want = cpuref.polymul(R, a[idx].cpu().numpy(), b[idx].cpu().numpy())
    assert np.array_equal(c[idx].cpu().numpy(), want.reshape(len(idx), n, 1))
    a_rows, b_rows = a[idx].cpu().numpy(), b[idx].cpu().numpy()
    P.mul(a, b)                                                   # streaming kernel, in place, 64-bit indices
    assert np.array_equal(a[idx].cpu().numpy(), cpuref.mul(R, a_rows, b_rows).reshape(len(idx), n, 1))
    del a, b, c
    torch.cuda.empty_cache()


def test_config3_she_ciphertext_product(gpu, cpuref):
    """m = 2^15, four ~59-bit moduli: (c0,c1)*(d0,d1) -> mulG of the three products
    (SymmSHE.hs:444-452), all on the GPU, sample rows against the oracle composition."""
    torch = pytest.importorskip("torch")
    m = 2 ** 15
    g = lm.good_qs(m, 2 ** 59)
    qs = [next(g) for _ in range(4)]
    P, R = gpu.Plan([(2, 15)], qs), Params([(2, 15)], qs)
    B = 64
    rng = np.random.default_rng(3)
    c0, c1, d0, d1 = (R.random(rng, B) for _ in range(4))
    t = {k: torch.from_numpy(v).cuda() for k, v in dict(c0=c0, c1=c1, d0=d0, d1=d1).items()}
    qv = torch.tensor(qs, dtype=torch.int64, device="cuda")
    e0 = torch.empty_like(t["c0"]); P.polymul(t["c0"], t["d0"], out=e0)
    x01 = torch.empty_like(e0); P.polymul(t["c0"], t["d1"], out=x01)
    x10 = torch.empty_like(e0); P.polymul(t["c1"], t["d0"], out=x10)
    e1 = (x01 + x10) % qv
    e2 = torch.empty_like(e0); P.polymul(t["c1"], t["d1"], out=e2)
    for e in (e0, e1, e2):
        P.mulGPow(e)                       # identity for m = 2^k apart from canonicalisation
    idx = [0, B - 1]
    w0 = cpuref.gpow(R, cpuref.polymul(R, c0[idx], d0[idx]))
    w2 = cpuref.gpow(R, cpuref.polymul(R, c1[idx], d1[idx]))
    s = (cpuref.polymul(R, c0[idx], d1[idx]).astype(object) + cpuref.polymul(R, c1[idx], d0[idx]).astype(object)) % np.array(qs, dtype=object)
    assert np.array_equal(e0[idx].cpu().numpy(), w0)
    assert np.array_equal(e2[idx].cpu().numpy(), w2)
    assert np.array_equal(e1[idx].cpu().numpy(), cpuref.gpow(R, s.astype(np.int64)))


@pytest.mark.parametrize("m", [64 * 9 * 25, 128 * 7 * 13, 2 ** 11 * 7])
def test_reference_index_shapes_full_batch(gpu, cpuref, m):
    """The reference's own benchmark / tunnelling indices (Benchmarks/Default.hs:42-50; lol-apps Default.hs:49-56),
    m = 2^e * odd through the one-launch route at a launch-filling batch: samples against the oracle, then the
    size-independent properties over the whole batch (round trips, ring laws, aliasing), two moduli of different
    classes in one tuple and a reference-sized single modulus."""
    torch = pytest.importorskip("torch")
    pps = lm.factor_pps(m)
    for qs in ([lm.first_good_q(m, 2 ** 26)], [lm.first_good_q(m, 2 ** 24), lm.first_good_q(m, 2 ** 58)]):
        P, R = gpu.Plan(pps, qs), Params(pps, qs)
        B = 2048 // len(qs)
        g = torch.Generator(device="cuda"); g.manual_seed(m)
        rnd = lambda: torch.stack([torch.randint(0, q, (B, R.n), dtype=torch.int64, device="cuda", generator=g) for q in qs], dim=-1)
        a, b = rnd(), rnd()
        x = a.clone(); P.crt(x)
        idx = [0, B // 2 - 1, B // 2, B - 1]
        assert np.array_equal(x[idx].cpu().numpy(), cpuref.crt(R, a[idx].cpu().numpy()))
        P.crtInv(x)
        assert torch.equal(x, a)
        for f, finv in ((P.l, P.lInv), (P.mulGPow, P.divGPow), (P.mulGDec, P.divGDec)):
            x = a.clone(); f(x); finv(x)
            assert torch.equal(x, a)
        c = torch.empty_like(a)
        P.polymul(a, b, out=c)
        assert np.array_equal(c[idx].cpu().numpy(), cpuref.polymul(R, a[idx].cpu().numpy(), b[idx].cpu().numpy()).reshape(len(idx), R.n, len(qs)))
        c2 = torch.empty_like(a)
        P.polymul(b, a, out=c2)
        assert torch.equal(c, c2)
        one = torch.zeros_like(a); one[:, 0, :] = 1
        P.polymul(a, one, out=c2)
        assert torch.equal(c2, a)
        x = a.clone(); P.polymul(x, b, out=x)
        assert torch.equal(x, c)
        # crtInv(crt a * crt b) op by op equals the fused launch
        ah, bh = a.clone(), b.clone()
        P.crt(ah); P.crt(bh); P.mul(ah, bh); P.crtInv(ah)
        assert torch.equal(ah, c)


def test_config4_mixed_radix_full_batch(gpu, cpuref):
    """m = 15015 = 3*5*7*11*13 (n = 5760), batch 1024, q just above 2^60 and 2^30"""
    torch = pytest.importorskip("torch")
    m = 15015
    pps = lm.factor_pps(m)
    for lower in (2 ** 60, 2 ** 30):
        q = lm.first_good_q(m, lower)
        P, R = gpu.Plan(pps, [q]), Params(pps, [q])
        B = 1024
        g = torch.Generator(device="cuda"); g.manual_seed(4)
        a = torch.randint(0, q, (B, R.n, 1), dtype=torch.int64, device="cuda", generator=g)
        x = a.clone(); P.crt(x)
        idx = [0, 511, 1023]
        assert np.array_equal(x[idx].cpu().numpy(), cpuref.crt(R, a[idx].cpu().numpy()))
        P.crtInv(x)
        assert torch.equal(x, a)
        x = a.clone(); P.mulGDec(x); P.divGDec(x)
        assert torch.equal(x, a)
        x = a.clone(); P.l(x); P.lInv(x)
        assert torch.equal(x, a)
        # the fused one-launch poly-mul at full batch: samples against the oracle, then the ring
        # laws over the whole batch (commutativity, the unit, c = a aliasing, squaring)
        b = torch.randint(0, q, (B, R.n, 1), dtype=torch.int64, device="cuda", generator=g)
        c = torch.empty_like(a)
        P.polymul(a, b, out=c)
        idx = [0, 1, 340, 341, 342, 1022, 1023]           # both ends and a workgroup-round boundary
        assert np.array_equal(c[idx].cpu().numpy(), cpuref.polymul(R, a[idx].cpu().numpy(), b[idx].cpu().numpy()).reshape(len(idx), R.n, 1))
        c2 = torch.empty_like(a)
        P.polymul(b, a, out=c2)
        assert torch.equal(c, c2)
        one = torch.zeros_like(a); one[:, 0, 0] = 1
        P.polymul(a, one, out=c2)
        assert torch.equal(c2, a)
        x = a.clone(); P.polymul(x, b, out=x)
        assert torch.equal(x, c)
        P.polymul(a, a, out=c2)                           # the a == b path skips the second transform
        aa = a.clone(); P.polymul(a, aa, out=c)
        assert torch.equal(c, c2)
        assert np.array_equal(c2[[5]].cpu().numpy(), cpuref.polymul(R, a[[5]].cpu().numpy(), a[[5]].cpu().numpy()).reshape(1, R.n, 1))


def test_config5_keyswitch_shapes(gpu, cpuref):
    """n = 1024, two 20-bit moduli (lol-apps Benchmarks/Default.hs:49), a large batch:
    the op mix of a TrivGad key switch — crtInv, per-component lift, crt, multiply-accumulate
    with the hint — plus ring embed 2048 -> 2048*7 (SymmSHE.hs:302-314, 361-371, 477-487)."""
    torch = pytest.importorskip("torch")
    m, m2 = 2048, 2048 * 7
    qs = [1017857, 1032193]
    qs_ok = all(lm.is_prime(q) and (q - 1) % m == 0 for q in qs)
    assert qs_ok
    P, R = gpu.Plan([(2, 11)], qs), Params([(2, 11)], qs)
    B = 8192
    rng = np.random.default_rng(5)
    c2 = R.random(rng, B)                       # the quadratic ciphertext component, CRT basis
    hint0, hint1 = R.random(rng, 1), R.random(rng, 1)
    d = torch.from_numpy(c2).cuda()
    P.crtInv(d)                                 # decompose works in the powerful basis
    digits = []
    for t, qt in enumerate(qs):                 # TrivGad: digit t = lift of component t, reduced mod every q
        lifted = d[:, :, t:t + 1].clone()
        half = qt // 2
        lifted = torch.where(lifted > half, lifted - qt, lifted)      # centred lift (ZqBasic.hs:227-232)
        dig = torch.remainder(lifted.expand(-1, -1, len(qs)), torch.tensor(qs, device="cuda")).contiguous()
        P.crt(dig)
        digits.append(dig)
    h0 = torch.from_numpy(np.broadcast_to(hint0, c2.shape).copy()).cuda()
    acc = torch.zeros_like(d)
    for dig in digits:
        prod = dig.clone(); P.mul(prod, h0)
        acc = (acc + prod) % torch.tensor(qs, device="cuda")
    # oracle for two sample rows
    idx = [0, B - 1]
    pw = cpuref.crtinv(R, c2[idx])
    want = np.zeros_like(pw)
    for t, qt in enumerate(qs):
        lifted = pw[:, :, t:t + 1].astype(np.int64)
        lifted = np.where(lifted > qt // 2, lifted - qt, lifted)
        dig = np.mod(np.broadcast_to(lifted, pw.shape), np.array(qs)).astype(np.int64)
        want = (want + cpuref.mul(R, cpuref.crt(R, dig), np.broadcast_to(hint0, dig.shape))) % np.array(qs)
    assert np.array_equal(acc[idx].cpu().numpy(), want)
    # ring embed of the result into the larger ring, CRT basis both sides
    g2 = [q for q in qs if (q - 1) % m2 == 0]
    if len(g2) == len(qs):
        Ph = gpu.Plan(lm.factor_pps(m2), qs)
        X = gpu.Ext(P, Ph)
        e = X.embedCRT(acc[:16].contiguous())
        assert np.array_equal(X.twaceCRT(e).cpu().numpy(), acc[:16].cpu().numpy())
