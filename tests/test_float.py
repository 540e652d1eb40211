"""Floating-point members of the Tensor class: CRT over C and the Gaussian
decoding-basis map, float64.  Tolerance contract, stated once: the largest absolute error over a
polynomial is at most 1e-12 times the largest absolute value of the expected result.  (The integer
paths are bit-exact; these are not — the reference itself depends on libm and on summation order.)

Oracle: lol-cpp's own tensorCRTC / tensorCRTInvC / tensorGaussianDec (crt.cpp:583-598,
random.cpp:61-64), run from the reference's sources in this container by
tests/golden/make_golden_float.py -> tests/golden/golden_float.npz; plus the numpy restatement
oracle/floatref.py (closed form of SURVEY.md Appendix A over C), pinned on those fixtures."""
import os

import numpy as np
import pytest

from oracle import floatref as fr
from oracle import lolmath as lm

HERE = os.path.dirname(os.path.abspath(__file__))
RTOL = 1e-12


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(HERE, "golden", "golden_float.npz"))


def close(got, want):
    got, want = np.asarray(got), np.asarray(want)
    return got.shape == want.shape and np.max(np.abs(got - want)) <= RTOL * max(1.0, np.max(np.abs(want)))


def _indices(gold):
    return [int(m) for m in gold["indices"]]


def test_restatement_matches_reference_fixtures(gold):
    for m in _indices(gold):
        pps = lm.factor_pps(m)
        n = lm.totient_pps(pps)
        assert close(fr.gaussian_dec(pps, gold[f"m{m}_gin"]), gold[f"m{m}_gauss"]), m
        if n <= 600:                                     # dense n x n closed form
            assert close(fr.crt_c(pps, gold[f"m{m}_cin"]), gold[f"m{m}_crtc"]), m
            # the inverse through a numerical matrix inverse: looser (conditioning of inv), still tight
            got = fr.crtinv_c(pps, gold[f"m{m}_cin"])
            assert np.max(np.abs(got - gold[f"m{m}_crtinvc"])) <= 1e-10 * np.max(np.abs(gold[f"m{m}_crtinvc"])), m


def test_restatement_matches_reference_live():
    from oracle.oracle import CTREF_SO, CTRef
    if not os.path.exists(CTREF_SO):
        pytest.skip("reference library not built here (no /root/reference)")
    ct = CTRef()
    rng = np.random.default_rng(5)
    for m in (12, 25, 35, 2 ** 6 * 3):
        pps = lm.factor_pps(m)
        n = lm.totient_pps(pps)
        z = rng.standard_normal((2, n)) + 1j * rng.standard_normal((2, n))
        assert close(fr.crt_c(pps, z), ct.crtc(pps, z)), m
        assert close(ct.crtinvc(pps, ct.crtc(pps, z)), z), m
        g = rng.standard_normal((2, n))
        assert close(fr.gaussian_dec(pps, g), ct.gaussian_dec(pps, g)), m


@pytest.mark.gpu
def test_gpu_float_ops_match_lolcpp(gpu, gold):
    for m in _indices(gold):
        pps = lm.factor_pps(m)
        q = lm.first_good_q(m, 1000)
        P = gpu.Plan(pps, [q])
        assert close(P.crtC(gold[f"m{m}_cin"]), gold[f"m{m}_crtc"]), ("crtC", m)
        assert close(P.crtInvC(gold[f"m{m}_cin"]), gold[f"m{m}_crtinvc"]), ("crtInvC", m)
        assert close(P.gaussianDec(gold[f"m{m}_gin"]), gold[f"m{m}_gauss"]), ("gaussianDec", m)
        z = gold[f"m{m}_cin"]
        assert close(P.crtInvC(P.crtC(z)), z), ("round trip", m)


@pytest.mark.gpu
def test_gpu_float_ops_without_a_crt_basis_and_limits(gpu):
    """CRT over C is what UCyc uses exactly when the modulus has NO CRT basis (UCyc.hs:422-444):
    the plan's moduli play no role.  Primes above 13 and n > 8192 are refused, not mangled."""
    import torch
    pps = lm.factor_pps(45)
    P = gpu.Plan(pps, [17])                              # 45 does not divide 16: no CRT basis mod 17
    assert not P.has_crt
    rng = np.random.default_rng(3)
    z = rng.standard_normal((4, P.n)) + 1j * rng.standard_normal((4, P.n))
    assert close(P.crtC(z), fr.crt_c(pps, z))
    d = torch.from_numpy(z).cuda()
    P.crtC(d); P.crtInvC(d)                               # device tensors: in place
    assert close(d.cpu().numpy(), z)
    g = rng.standard_normal((4, P.n))
    assert close(P.gaussianDec(g), fr.gaussian_dec(pps, g))
    with pytest.raises(gpu.LolHipError):
        gpu.Plan(lm.factor_pps(17), [103]).crtC(np.zeros((1, 16), dtype=np.complex128))
    with pytest.raises(gpu.LolHipError):
        gpu.Plan([(2, 15)], [lm.first_good_q(2 ** 15, 1000)]).gaussianDec(np.zeros((1, 2 ** 14)))
    assert P.crtC(np.zeros((0, P.n), dtype=np.complex128)).shape == (0, P.n)
