"""The reference's own end-to-end SHE properties (lol-apps/Crypto/Lol/Applications/Tests/
SHETests.hs:40-248) over the model of SymmSHE in oracle/she_model.py, with every ring operation
executed by the engine under test: the CPU oracle (this proves the model), then the GPU.

    prop_encDec   Dec . Enc = id                                  SHETests.hs (decTest)
    prop_ctmul    Dec (c * d) = Dec c * Dec d                     SHETests.hs (CTMul)
    prop_ksQuad   Dec (keySwitchQuadCirc hint (c * d)) = ...      SHETests.hs (KSQuad), TrivGad and BaseBGad
    modSwitch     Dec under the modulus with its first component dropped (SymmSHE.hs:243-246)

These runs pin what fixtures cannot (the reference ships none for its Haskell-only SHE layer, so
the pipelines stay "parity unpinned" by the fixture rule): digit order and centring of decompose,
the hint slab layout [L][K][n][T], gadget conventions, MSD/LSD scaling, the basis each step
expects, mulG/divG bookkeeping — an error in any of them breaks decryption."""
import numpy as np
import pytest

from oracle import lolmath as lm
from oracle import she_model as sm
from oracle.oracle import Params

# (m, plaintext modulus p = 1 mod m, lower bound of the two ciphertext moduli)
CASES = [(64, 257, 2 ** 29), (48, 97, 2 ** 29), (45, 181, 2 ** 30)]
BASES = [0, 256, 5]


def _setup(m, p, lower):
    pps = lm.factor_pps(m)
    g = lm.good_qs(m, lower)
    qs = [next(g), next(g)]
    return pps, qs


def _expected_product(cpuref, pps, p, a, b):
    P = Params(pps, [p])
    return cpuref.polymul(P, a[..., None], b[..., None]).reshape(a.shape)


def _run_properties(make_engine, cpuref, m, p, lower, base, seed):
    pps, qs = _setup(m, p, lower)
    rng = np.random.default_rng(seed)
    she = sm.SHE(make_engine(pps, qs), make_engine(pps, [p]), qs, p, rng)
    she.keygen()
    B = 3
    n = she.n
    pt1 = rng.integers(0, p, size=(B, n), dtype=np.int64)
    pt2 = rng.integers(0, p, size=(B, n), dtype=np.int64)
    pt1[0], pt2[1] = 0, 1                                         # zero and a non-trivial constant-ish edge
    ct1, ct2 = she.encrypt(pt1), she.encrypt(pt2)
    assert np.array_equal(she.decrypt(ct1), pt1), "Dec . Enc"
    assert np.array_equal(she.decrypt(she.toMSD(ct1)), pt1), "Dec . toMSD . Enc"
    prod = she.mul(ct1, ct2)
    want = _expected_product(cpuref, pps, p, pt1, pt2)
    assert np.array_equal(she.decrypt(prod), want), "CTMul"
    hint = she.ks_quad_hint(base)
    assert hint.shape == (she.e.decomposeLen(base), 2, n, 2)
    lin = she.key_switch_quad(hint, base, prod)
    assert len(lin["c"]) == 2
    assert np.array_equal(she.decrypt(lin), want), "KSQuad"
    # modulus switching of a fresh ciphertext down to the second modulus alone
    ct_small, she2 = she.mod_switch_drop_first(ct1, make_engine(pps, qs[1:]))
    assert np.array_equal(she2.decrypt(ct_small), pt1), "modSwitch"


@pytest.mark.parametrize("m,p,lower", CASES[:2])
@pytest.mark.parametrize("base", BASES[:2])
def test_she_properties_hold_for_the_model_on_the_cpu_oracle(cpuref, m, p, lower, base):
    _run_properties(lambda pps, qs: sm.CpuEngine(cpuref, Params(pps, qs)), cpuref, m, p, lower, base, seed=m + base)


@pytest.mark.gpu
@pytest.mark.parametrize("m,p,lower", CASES)
@pytest.mark.parametrize("base", BASES)
def test_gpu_she_properties(gpu, cpuref, m, p, lower, base):
    _run_properties(lambda pps, qs: gpu.Plan(pps, qs), cpuref, m, p, lower, base, seed=1000 + m + base)


@pytest.mark.gpu
def test_gpu_she_properties_at_config5_shape(gpu):
    """m' = 2048, q = (1017857, 1032193) — BASELINE config 5's ring and moduli (lol-apps
    Benchmarks/Default.hs:49), plaintext modulus 2: the fused single-pass key switch is the
    kernel that runs here (TrivGad and base 256)."""
    pps, qs, p = lm.factor_pps(2048), [1017857, 1032193], 2
    rng = np.random.default_rng(77)
    she = sm.SHE(gpu.Plan(pps, qs), gpu.Plan(pps, [p]), qs, p, rng)
    she.keygen()
    pt1 = rng.integers(0, p, size=(2, she.n), dtype=np.int64)
    pt2 = rng.integers(0, p, size=(2, she.n), dtype=np.int64)
    ct1, ct2 = she.encrypt(pt1), she.encrypt(pt2)
    assert np.array_equal(she.decrypt(ct1), pt1)
    # the plaintext product through the ring product mod q of the 0/1 polynomials (no wrap-around: |coeff| <= n)
    want = (she.lift(she.rmul(she.reduce(pt1), she.reduce(pt2))) % p).astype(np.int64)
    prod = she.mul(ct1, ct2)
    assert np.array_equal(she.decrypt(prod), want)
    for base in (0, 256):
        lin = she.key_switch_quad(she.ks_quad_hint(base), base, prod)
        assert np.array_equal(she.decrypt(lin), want), base
