"""CPU tests of the ORACLE itself (no GPU): the C restatement oracle/cpu_ref.c is pinned
 (a) against the committed golden vectors, which are outputs of the reference's own C++,
 (b) against the reference's C++ live, when oracle/_ref is present,
 (c) against the closed-form definition incl. 61-bit moduli,
 (d) through the reference's property list (lol/Crypto/Lol/Tests/TensorTests.hs:38-78,133-234).
"""
import numpy as np
import pytest

from oracle import lolmath as lm
from oracle.oracle import Params
from params import BIG, PRIME_OPS, PRIMEOPS_ONLY, TENSOR1, TENSOR2


def _params_nocrt(m, qs):
    P = Params.__new__(Params)
    P.pps = lm.factor_pps(m)
    P.qs, P.T, P.m, P.n = list(qs), len(qs), m, lm.totient_pps(P.pps)
    return P


def _i64(a):
    return np.asarray(a).astype(np.int64)


@pytest.mark.parametrize("i", range(len(TENSOR1)))
def test_restatement_matches_golden_tensor1(cpuref, golden, i):
    m, qs = TENSOR1[i]
    P = Params(lm.factor_pps(m), qs)
    y, z = _i64(golden[f"t1_{i}/y"]), _i64(golden[f"t1_{i}/z"])
    assert np.array_equal(cpuref.crt(P, y), _i64(golden[f"t1_{i}/crt"]))
    assert np.array_equal(cpuref.crtinv(P, y), _i64(golden[f"t1_{i}/crtinv"]))
    assert np.array_equal(cpuref.mul(P, y, z), _i64(golden[f"t1_{i}/mul"]))
    assert np.array_equal(cpuref.polymul(P, y, z), _i64(golden[f"t1_{i}/polymul"]))
    for op in PRIME_OPS:
        got = getattr(cpuref, op)(P, y)
        key = f"t1_{i}/{op}"
        if key in golden:
            assert np.array_equal(got, _i64(golden[key])), op
        else:
            assert got is None, op


@pytest.mark.parametrize("i", range(len(PRIMEOPS_ONLY)))
def test_restatement_matches_golden_noncrt_moduli(cpuref, golden, i):
    m, qs = PRIMEOPS_ONLY[i]
    P = _params_nocrt(m, qs)
    y = _i64(golden[f"po_{i}/y"])
    for op in PRIME_OPS:
        got = getattr(cpuref, op)(P, y)
        key = f"po_{i}/{op}"
        if key in golden:
            assert np.array_equal(got, _i64(golden[key])), op
        else:
            assert got is None


@pytest.mark.parametrize("i", range(len(BIG)))
def test_restatement_matches_golden_baseline_configs(cpuref, golden, i):
    m, qs, _seed = BIG[i]
    P = Params(lm.factor_pps(m), qs)
    y, z = _i64(golden[f"big_{i}/y"]), _i64(golden[f"big_{i}/z"])
    assert np.array_equal(cpuref.crt(P, y), _i64(golden[f"big_{i}/crt"]))
    assert np.array_equal(cpuref.crtinv(P, y), _i64(golden[f"big_{i}/crtinv"]))
    assert np.array_equal(cpuref.polymul(P, y, z), _i64(golden[f"big_{i}/polymul"]))


@pytest.mark.parametrize("i", range(len(TENSOR2)))
def test_twace_embed_restatement_matches_golden(cpuref, golden, i):
    m, m2, qs = TENSOR2[i]
    Pl, Ph = Params(lm.factor_pps(m), qs), Params(lm.factor_pps(m2), qs)
    lo, hi = _i64(golden[f"t2_{i}/lo"]), _i64(golden[f"t2_{i}/hi"])
    assert np.array_equal(cpuref.embed_pow(Pl, Ph, lo), _i64(golden[f"t2_{i}/embed_pow"]))
    assert np.array_equal(cpuref.twace_powdec(Pl, Ph, hi), _i64(golden[f"t2_{i}/twace_powdec"]))
    # these three golden arrays were produced THROUGH the reference's CRT / L (make_golden.py)
    assert np.array_equal(cpuref.embed_crt(Pl, Ph, lo), _i64(golden[f"t2_{i}/embed_crt"]))
    assert np.array_equal(cpuref.twace_crt(Pl, Ph, hi), _i64(golden[f"t2_{i}/twace_crt"]))
    assert np.array_equal(cpuref.embed_dec(Pl, Ph, lo), _i64(golden[f"t2_{i}/embed_dec"]))


def test_restatement_matches_reference_live(cpuref, ctref):
    """Fresh random inputs through the reference's own C++ (only where oracle/_ref exists)."""
    rng = np.random.default_rng(11)
    for m, qs in TENSOR1 + [(45, [1171, 1531]), (1155, [1051051]), (432, [1297]), (2048, [12289])]:
        P = Params(lm.factor_pps(m), qs)
        y, z = P.random(rng, 2), P.random(rng, 2)
        for op in ("crt", "crtinv") + PRIME_OPS:
            a, b = getattr(cpuref, op)(P, y), getattr(ctref, op)(P, y)
            assert (a is None) == (b is None), (op, m)
            if a is not None:
                assert np.array_equal(a.reshape(b.shape), b), (op, m, qs)
        assert np.array_equal(cpuref.mul(P, y, z).reshape(2, P.n, P.T), ctref.mul(P, y, z))


def test_negative_inputs_like_reference(cpuref, ctref):
    """The C side accepts representatives in (-q, q) (types.h:52-57) and returns [0, q)."""
    rng = np.random.default_rng(12)
    P = Params(lm.factor_pps(21), [8191])
    y = P.random(rng, 1) - 8191 // 2
    for op in ("crt", "crtinv", "l", "linv", "gpow", "gdec", "ginvpow", "ginvdec"):
        a, b = getattr(cpuref, op)(P, y), getattr(ctref, op)(P, y)
        assert np.array_equal(a.reshape(b.shape), b), op
        assert a.min() >= 0 and a.max() < 8191


@pytest.mark.parametrize("m,lower", [(7, 2 ** 60), (12, 2 ** 61 - 2 ** 40), (16, 2 ** 60), (45, 2 ** 59),
                                     (64, 2 ** 60), (27, 2 ** 45), (22, 2 ** 33), (105, 2 ** 60)])
def test_closed_form_large_moduli(cpuref, m, lower):
    """Beyond q ~ 2^31.5 the reference overflows (types.h:79-84); there the pin is the exact
    mathematical definition CT implements."""
    pps = lm.factor_pps(m)
    q = lm.first_good_q(m, lower)
    P = Params(pps, [q])
    rng = np.random.default_rng(m)
    y = P.random(rng, 1)
    want = cpuref.crt_naive(P, y[0, :, 0])
    assert np.array_equal(cpuref.crt(P, y)[0, :, 0], want)
    # and independently in Python big-ints for the first few outputs
    for i in range(min(P.n, 3)):
        acc = 0
        for j in range(P.n):
            ii, jj, tw = i, j, 1
            for k, (p, e) in enumerate(pps):
                phi = lm.totient_pp((p, e))
                ik, jk = ii % phi, jj % phi
                ii //= phi
                jj //= phi
                w = P.ru[k][1]
                tw = tw * pow(w, lm.index_to_pow((p, e), jk) * lm.index_to_zms((p, e), ik), q) % q
            acc = (acc + int(y[0, j, 0]) * tw) % q
        assert acc == int(want[i])


def test_pow2_index_convention(cpuref):
    """m = 2^e: Y[i] = sum_j a[j] * omega^(bitrev(j) * (2i+1))  (SURVEY.md fact 5)."""
    m, q = 32, lm.first_good_q(32, 2 ** 60)
    P = Params([(2, 5)], [q])
    w = lm.omega(m, q)
    rng = np.random.default_rng(3)
    y = P.random(rng, 1)
    out = cpuref.crt(P, y)[0, :, 0]
    n = 16
    for i in range(n):
        s = sum(int(y[0, j, 0]) * pow(w, lm.digit_rev(2, 4, j) * (2 * i + 1), q) for j in range(n)) % q
        assert s == int(out[i])


# ---- the reference's property list, on the oracle (TensorTests.hs:38-78) ---------------

@pytest.mark.parametrize("m,qs", TENSOR1)
def test_tensor_properties(cpuref, m, qs):
    P = Params(lm.factor_pps(m), qs)
    rng = np.random.default_rng(m * 7 + len(qs))
    y = P.random(rng, 3)
    assert np.array_equal(cpuref.crtinv(P, cpuref.crt(P, y)), y)                         # prop_crt_inv
    assert np.array_equal(cpuref.linv(P, cpuref.l(P, y)), y)                             # prop_l_inv
    assert np.array_equal(cpuref.ginvpow(P, cpuref.gpow(P, y)), y)                       # prop_ginv_pow
    assert np.array_equal(cpuref.ginvdec(P, cpuref.gdec(P, y)), y)                       # prop_ginv_dec
    assert np.array_equal(cpuref.gdec(P, y), cpuref.linv(P, cpuref.gpow(P, cpuref.l(P, y))))   # prop_g_dec
    # prop_scalar_crt: crt of the constant polynomial is the all-c vector
    c = np.zeros_like(y)
    c[:, 0, :] = y[:, 0, :]
    assert np.array_equal(cpuref.crt(P, c), np.broadcast_to(y[:, :1, :], y.shape))
    # prop_g_crt with the g vector built by Tensor.hs:315-337
    g = np.stack([lm.g_crt(P.pps, q) for q in qs], axis=-1).astype(object)
    want = cpuref.crt(P, cpuref.gpow(P, cpuref.crtinv(P, y)))
    got = (y.astype(object) * g[None]) % np.array(qs, dtype=object)
    assert np.array_equal(got.astype(np.int64), want)
    gi = np.stack([lm.g_crt(P.pps, q, inverse=True) for q in qs], axis=-1).astype(object)
    assert np.array_equal(((got * gi[None]) % np.array(qs, dtype=object)).astype(np.int64), y)   # prop_ginv_crt


@pytest.mark.parametrize("m,m2,qs", TENSOR2)
def test_tensor_two_index_properties(cpuref, m, m2, qs):
    Pl, Ph = Params(lm.factor_pps(m), qs), Params(lm.factor_pps(m2), qs)
    rng = np.random.default_rng(m * 100 + m2)
    lo, hi = Pl.random(rng, 2), Ph.random(rng, 2)
    assert np.array_equal(cpuref.twace_powdec(Pl, Ph, cpuref.embed_pow(Pl, Ph, lo)), lo)      # prop_twEmID pow
    assert np.array_equal(cpuref.twace_powdec(Pl, Ph, cpuref.embed_dec(Pl, Ph, lo)), lo)      # prop_twEmID dec
    assert np.array_equal(cpuref.twace_crt(Pl, Ph, cpuref.embed_crt(Pl, Ph, lo)), lo)          # prop_twEmID crt
    assert np.array_equal(cpuref.embed_dec(Pl, Ph, lo), cpuref.linv(Ph, cpuref.embed_pow(Pl, Ph, cpuref.l(Pl, lo))))
    assert np.array_equal(cpuref.embed_crt(Pl, Ph, lo), cpuref.crt(Ph, cpuref.embed_pow(Pl, Ph, cpuref.crtinv(Pl, lo))))
    assert np.array_equal(cpuref.twace_crt(Pl, Ph, hi), cpuref.crt(Pl, cpuref.twace_powdec(Pl, Ph, cpuref.crtinv(Ph, hi))))
    # prop_twace_invar1 (TensorTests.hs:199-215): Tw(g'^-1 ... ) — here its Pow-basis corollary:
    # twace of a divisible-by-g element commutes with divG when m = m'
    if m == m2:
        assert np.array_equal(cpuref.twace_powdec(Pl, Ph, hi), hi)


@pytest.mark.parametrize("m,m2,q", [(4, 12, 13), (3, 21, 43), (8, 8, 17), (1, 8, 17), (7, 63, 127), (12, 60, 61)])
def test_coeffs_against_the_relative_powerful_basis(cpuref, m, m2, q):
    """prop_coeffsBasis (CycTests.hs:71-76): sum_i embed(coeffs_i x) * b_i = x, where b_i is the
    i-th relative powerful-basis element — in the powerful basis of O_m' the unit vector at
    fromIndexPair (i, 0) (powBasisPow, Tensor.hs:177 over :472-477).  Ring products by the
    C++-pinned poly-mul."""
    a, b = lm.factor_pps(m), lm.factor_pps(m2)
    Rl, Rh = Params(a, [q]), Params(b, [q])
    rng = np.random.default_rng(m2)
    x = Rh.random(rng, 2)
    cs = cpuref.coeffs(Rl, Rh, x)
    idx = lm.ext_indices_coeffs(a, b)
    assert cs.shape == (Rh.n // Rl.n, 2, Rl.n, 1)
    assert sorted(e for row in idx for e in row) == list(range(Rh.n))          # a permutation of O_m'
    acc = np.zeros_like(x)
    for i1, row in enumerate(idx):
        basis = np.zeros_like(x)
        basis[:, row[0], :] = 1
        acc = (acc + cpuref.polymul(Rh, cpuref.embed_pow(Rl, Rh, cs[i1]), basis).reshape(x.shape)) % q
    assert np.array_equal(acc, x)
