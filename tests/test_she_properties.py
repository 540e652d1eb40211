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
    # ... and of the key-switched product: multiply -> key switch -> rescale -> decrypt, one chain
    lin_small, she3 = she.mod_switch_drop_first(lin, make_engine(pps, qs[1:]))
    assert np.array_equal(she3.decrypt(lin_small), want), "KSQuad . modSwitch"


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


# ---------------------------------------------------------------------------------------------
# prop_ringTunnel (SHETests.hs:228-249): Dec_skout (tunnel hints (Enc_skin x)) = evalLin f x for a random E-linear
# f : R -> S given on the relative decoding basis, here with r' = r and s' = s.  Until round 3 lolhip_tunnel_batch was
# only compared with the same composition of the restatement — itself; this runs the reference's own contract.
# ---------------------------------------------------------------------------------------------
TUNNEL_CASES = [(4, 12, 20), (8, 16, 40), (1, 8, 8)]          # (e, r, s) with e = gcd(r, s)
TUNNEL_CHAIN = (128, 128 * 7, 128 * 13)                        # the shape of lol-apps' tunnelling chain hops


class TunnelEngine:
    """lol_amd.Ext pairs as the engine of oracle/she_model.py's tunnel."""

    def __init__(self, gpu, GE, GR, GS):
        self.XR, self.XS = gpu.Ext(GE, GR), gpu.Ext(GE, GS)

    def evalLin(self, r_dec, ys_crt): return self.XR.evalLin(self.XS, r_dec, ys_crt)
    def tunnel(self, c0_dec, c1_pow, ys_crt, hints, base): return self.XR.tunnel(self.XS, c0_dec, c1_pow, ys_crt, hints, base)


def _run_tunnel(make_engine, make_tunnel_engine, cpuref, e, r, s, base, seed, B=2):
    import math
    from oracle import she_ref as sr
    lcm = r * s // math.gcd(r, s)
    assert math.gcd(r, s) % e == 0                               # the property fixes e = gcd(r, s); any common subring works
    p = lm.first_good_q(lcm, 40)                                  # plaintext modulus with a CRT basis in R and S
    g = lm.good_qs(lcm, 2 ** 29)
    qs = [next(g), next(g)]
    pe, pr, ps = (lm.factor_pps(m) for m in (e, r, s))
    rng = np.random.default_rng(seed)
    she_in = sm.SHE(make_engine(pr, qs), make_engine(pr, [p]), qs, p, rng)
    she_out = sm.SHE(make_engine(ps, qs), make_engine(ps, [p]), qs, p, rng)
    she_in.keygen(); she_out.keygen()
    xeng = make_tunnel_engine(pe, pr, ps, qs)
    rel_index = [row[0] for row in lm.ext_indices_coeffs(pe, pr)]
    rel = len(rel_index)
    f_vals = rng.integers(0, p, size=(rel, she_out.n), dtype=np.int64)          # bs <- replicateM basisSize getRandom
    x = rng.integers(0, p, size=(B, she_in.n), dtype=np.int64)
    x[0] = 0
    ys_crt, hints = sm.tunnel_hint(she_in, she_out, xeng, rel_index, f_vals, base)
    assert hints.shape[:3] == (rel, she_out.e.decomposeLen(base), 2)
    ct = she_in.encrypt(x)
    assert np.array_equal(she_in.decrypt(ct), x)
    out = sm.tunnel(she_in, xeng, ys_crt, hints, base, ct)
    out["c"] = [she_out.e.crtInv(np.ascontiguousarray(c)) for c in out["c"]]
    got = she_out.decrypt(out)
    # expected = evalLin f x over Z_p, by the CPU restatement of Linear.hs:75-79 (CRT basis of S_p -> powerful basis)
    PEp, PRp, PSp = (Params(q_, [p]) for q_ in (pe, pr, ps))
    x_dec = cpuref.linv(PRp, x[..., None]).reshape(B, she_in.n, 1)
    f_crt = cpuref.crt(PSp, f_vals[..., None]).reshape(rel, she_out.n, 1)
    want = cpuref.crtinv(PSp, sr.evallin(cpuref, PEp, PRp, PSp, x_dec, f_crt)).reshape(B, she_out.n)
    assert np.array_equal(got, want), (e, r, s, base)
    assert want.any()                                                           # not the zero map by accident


@pytest.mark.parametrize("e,r,s", TUNNEL_CASES[:2])
def test_ring_tunnel_property_holds_for_the_model_on_the_cpu_oracle(cpuref, e, r, s):
    _run_tunnel(lambda pps, qs: sm.CpuEngine(cpuref, Params(pps, qs)),
                lambda pe, pr, ps, qs: sm.CpuTunnelEngine(cpuref, Params(pe, qs), Params(pr, qs), Params(ps, qs)),
                cpuref, e, r, s, base=0, seed=e + r + s)


@pytest.mark.gpu
@pytest.mark.parametrize("e,r,s", TUNNEL_CASES + [TUNNEL_CHAIN])
@pytest.mark.parametrize("base", [0, 16])
def test_gpu_ring_tunnel_property(gpu, cpuref, e, r, s, base):
    _run_tunnel(lambda pps, qs: gpu.Plan(pps, qs),
                lambda pe, pr, ps, qs: TunnelEngine(gpu, gpu.Plan(pe, qs), gpu.Plan(pr, qs), gpu.Plan(ps, qs)),
                cpuref, e, r, s, base, seed=2000 + e + r + s + base)
