"""Ring-level pipelines of SymmSHE: ciphertext product, gadget
decomposition, knapsack / key switch, modulus rescaling.

The reference has these only in Haskell (no GHC here), and holds no vectors for them, so:
  * CPU tests pin the restatement `oracle/she_ref.py` by the reference's own algebraic
    contracts — the gadget law <g, decompose x> = x (class Decompose, Gadget.hs:60-72),
    centred digit ranges (Numeric.hs:200-205,225-234), rescale's defining congruence
    (Cyc.hs:529-542), and the ciphertext product against the ring product of the two
    ciphertext polynomials computed with the C++-pinned transforms (SymmSHE.hs:444-449);
  * GPU tests compare liblolhip, through the C ABI, bit for bit with that restatement, and
    re-check the same contracts at BASELINE configs 3 and 5 shapes.
"""
import numpy as np
import pytest

from oracle import lolmath as lm
from oracle import she_ref as sr
from oracle.oracle import Params

Q5 = [1017857, 1032193]                       # config 5 moduli (lol-apps Benchmarks/Default.hs:49)


def _qs_for(m, lower, T):
    g = lm.good_qs(m, lower)
    return [next(g) for _ in range(T)]


CASES = [                                      # (pps, qs)
    ([(2, 5)], [7681]),                        # one small modulus
    ([(2, 6)], _qs_for(64, 2 ** 20, 3)),
    ([(2, 11)], Q5),                           # m' = 2048 of config 5
    ([(3, 1), (5, 1)], _qs_for(15, 2 ** 29, 2)),
    ([(2, 4)], _qs_for(16, 2 ** 59, 4)),       # config-3-like 59-bit tuple
    ([(7, 1)], [lm.first_good_q(7, 2 ** 61)]),
]
BASES = [0, 2, 3, 7, 256, 2 ** 20]


# ------------------------------------------------------------------------------------
# CPU: the restatement obeys the reference's contracts
# ------------------------------------------------------------------------------------

def test_div_mod_cent_is_centred_floor_division():
    for b in (2, 3, 7, 10, 256):
        for a in range(-40, 41):
            q, r = sr.div_mod_cent(np.array([a]), b)
            assert int(q[0]) * b + int(r[0]) == a
            assert -(b // 2) <= int(r[0]) < b - b // 2        # [-b/2, b/2) (Numeric.hs:225-226)


@pytest.mark.parametrize("pps,qs", CASES)
@pytest.mark.parametrize("base", BASES)
def test_gadget_law(pps, qs, base):
    """<gadget, decompose c> = c in every component, digit counts per ZqBasic.hs:238-240."""
    R = Params(pps, qs)
    rng = np.random.default_rng(len(qs) * 100 + base % 97)
    c = R.random(rng, 2)
    c[0, 0] = 0
    c[0, 1] = np.array(qs) - 1
    c[0, 2 % R.n] = np.array(qs) // 2                         # the lift's break point
    d = sr.decompose(R, c, base)
    g = sr.gadget(R, base)
    assert d.shape == (sum(sr.digit_counts(R, base)), 2, R.n, R.T) and g.shape == (d.shape[0], R.T)
    assert d.min() >= 0 and (d < np.array(qs)).all()
    acc = sum(d[j].astype(object) * g[j].astype(object) for j in range(d.shape[0])) % np.array(qs, dtype=object)
    assert np.array_equal(acc.astype(np.int64), c)
    if base >= 2:                                             # centred digits, except the last of each component
        j = 0
        for t, k in enumerate(sr.digit_counts(R, base)):
            for kk in range(k - 1):
                lifted = sr.lift_centered(d[j + kk][..., t], qs[t]) if qs[t] > base else None
                if lifted is not None:
                    assert (lifted >= -(base // 2)).all() and (lifted < base - base // 2).all()
            j += k


@pytest.mark.parametrize("pps,qs", [c for c in CASES if len(c[1]) >= 2])
def test_rescale_contract(pps, qs):
    """q_a * rescale(c) + lift(a) = b in every remaining component; exact on multiples of q_a."""
    R = Params(pps, qs)
    rng = np.random.default_rng(11)
    c = R.random(rng, 3)
    out = sr.rescale_drop_first(R, c)
    z = sr.lift_centered(c[..., 0], qs[0])
    for s in range(1, R.T):
        lhs = (out[..., s - 1].astype(object) * qs[0] + z) % qs[s]
        assert np.array_equal(lhs.astype(np.int64), c[..., s])
    x = Params(pps, qs[1:]).random(rng, 3)                    # c = q_a * x  ->  rescale c = x
    c2 = np.concatenate([np.zeros_like(x[..., :1]), (x.astype(object) * qs[0] % np.array(qs[1:], dtype=object)).astype(np.int64)], axis=-1)
    assert np.array_equal(sr.rescale_drop_first(R, np.ascontiguousarray(c2)), x)


@pytest.mark.parametrize("pps,qs", CASES[:5])
def test_ctmul_is_the_ring_product_of_the_ciphertext_polynomials(cpuref, pps, qs):
    """crtInv of the three outputs = mulGPow of the coefficients of (c0 + c1 s)(d0 + d1 s),
    each ring product computed by the C++-pinned poly-mul (SymmSHE.hs:444-449)."""
    R = Params(pps, qs)
    rng = np.random.default_rng(5)
    c0, c1, d0, d1 = (R.random(rng, 2) for _ in range(4))
    crt = lambda x: cpuref.crt(R, x).reshape(x.shape)
    e = sr.ctmul_crt(cpuref, R, crt(c0), crt(c1), crt(d0), crt(d1))
    qv = np.array(qs, dtype=object)
    pm = lambda a, b: cpuref.polymul(R, a, b).reshape(a.shape)
    want = [pm(c0, d0), ((pm(c0, d1).astype(object) + pm(c1, d0).astype(object)) % qv).astype(np.int64), pm(c1, d1)]
    for got, w in zip(e, want):
        assert np.array_equal(cpuref.crtinv(R, got).reshape(w.shape), cpuref.gpow(R, np.ascontiguousarray(w)).reshape(w.shape))


@pytest.mark.parametrize("base", [0, 2, 256])
def test_keyswitch_with_a_noise_free_hint_multiplies_by_the_encoded_value(cpuref, base):
    """hint_j = (gadget_j * v, 0)  =>  switch hint c = (c * v, 0): the algebra behind
    prop_ksQuad (SHETests.hs) without the error terms."""
    pps, qs = [(2, 6)], _qs_for(64, 2 ** 20, 2)
    R = Params(pps, qs)
    rng = np.random.default_rng(base + 1)
    c, v = R.random(rng, 3), R.random(rng, 1)[0]
    g = sr.gadget(R, base)
    v_crt = cpuref.crt(R, v[None]).reshape(R.n, R.T)
    hint = np.zeros((g.shape[0], 2, R.n, R.T), dtype=np.int64)
    for j in range(g.shape[0]):
        hint[j, 0] = (v_crt.astype(object) * g[j].astype(object) % np.array(qs, dtype=object)).astype(np.int64)
    out = sr.keyswitch(cpuref, R, c, base, hint)
    want = cpuref.crt(R, cpuref.polymul(R, c, np.ascontiguousarray(np.broadcast_to(v, c.shape)))).reshape(c.shape)
    assert np.array_equal(out[0], want) and not out[1].any()


EXT_CASES = [(4, 12, 12, 13), (3, 21, 21, 43), (1, 8, 8, 17), (4, 12, 20, 61), (3, 21, 15, 211), (8, 16, 40, 241)]   # (e, r, s, q)


def _identity_ys(cpuref, PE, PR):
    """values of the identity function on the relative decoding basis of R/E, CRT basis of R.
    d_R = d_{R/E} (x) d_E under the index pairing of Tensor.hs:472-477, so the relative basis
    element d_{R/E,i} = d_{R/E,i} * 1_E has decoding coordinates u_k at index (i, k), where u is
    1 in E written in E's decoding basis (lInv of the powerful-basis unit vector)."""
    idx = lm.ext_indices_coeffs(PE.pps, PR.pps)
    one = np.zeros((1, PE.n, PE.T), dtype=np.int64)
    one[0, 0, :] = 1
    u = cpuref.linv(PE, one).reshape(PE.n, PE.T)
    ys = np.zeros((len(idx), PR.n, PR.T), dtype=np.int64)
    for i, row in enumerate(idx):
        ys[i, row, :] = u
    return cpuref.crt(PR, cpuref.l(PR, ys)).reshape(ys.shape)


@pytest.mark.parametrize("e,r,s,q", EXT_CASES)
def test_evallin_of_the_identity_and_linearity(cpuref, e, r, s, q):
    """evalLin (linearDec ds) = id when ds is the relative decoding basis itself (the defining
    property of linearDec, Linear.hs:62-72), and evalLin is additive in its function argument."""
    PE, PR, PS = (Params(lm.factor_pps(m), [q]) for m in (e, r, s))
    rng = np.random.default_rng(r * 7 + s)
    x = PR.random(rng, 2)
    got = sr.evallin(cpuref, PE, PR, PR, x, _identity_ys(cpuref, PE, PR))
    assert np.array_equal(got, cpuref.crt(PR, cpuref.l(PR, x)).reshape(x.shape))
    rel = PR.n // PE.n
    y1 = np.stack([PS.random(rng, 1)[0] for _ in range(rel)])
    y2 = np.stack([PS.random(rng, 1)[0] for _ in range(rel)])
    f1, f2 = sr.evallin(cpuref, PE, PR, PS, x, y1), sr.evallin(cpuref, PE, PR, PS, x, y2)
    f12 = sr.evallin(cpuref, PE, PR, PS, x, (y1 + y2) % q)
    assert np.array_equal(f12, (f1 + f2) % q)


@pytest.mark.gpu
@pytest.mark.parametrize("e,r,s,q", EXT_CASES + [(128, 128 * 7, 128 * 13, 23297)])
def test_gpu_evallin(gpu, cpuref, e, r, s, q):
    import math
    qs = [q, lm.first_good_q(r * s // math.gcd(r, s), q)]
    pe, pr, ps = (lm.factor_pps(m) for m in (e, r, s))
    PE, PR, PS = (Params(p_, qs) for p_ in (pe, pr, ps))
    GE, GR, GS = (gpu.Plan(p_, qs) for p_ in (pe, pr, ps))
    XR, XS = gpu.Ext(GE, GR), gpu.Ext(GE, GS)
    rng = np.random.default_rng(r + s)
    x = PR.random(rng, 3)
    ys = np.stack([PS.random(rng, 1)[0] for _ in range(PR.n // PE.n)])
    assert np.array_equal(XR.evalLin(XS, x, ys), sr.evallin(cpuref, PE, PR, PS, x, ys))
    if PR.n <= 2048:                                                   # identity function, GPU on both sides
        ident = _identity_ys(cpuref, PE, PR)
        assert np.array_equal(XR.evalLin(XR, x, ident), GR.crt(GR.l(x)))


@pytest.mark.gpu
@pytest.mark.parametrize("e,r,s,q", EXT_CASES[:4] + [(128, 128 * 7, 128 * 13, 23297)])
@pytest.mark.parametrize("base", [0, 16])
def test_gpu_tunnel(gpu, cpuref, e, r, s, q, base):
    """lolhip_tunnel_batch = evalLin on c0 + one key switch per relative powerful-basis coefficient of
    c1, against the same composition of the restatement (SymmSHE.hs:549-570)."""
    import math
    qs = [q, lm.first_good_q(r * s // math.gcd(r, s), q)]
    pe, pr, ps = (lm.factor_pps(m) for m in (e, r, s))
    PE, PR, PS = (Params(p_, qs) for p_ in (pe, pr, ps))
    GE, GR, GS = (gpu.Plan(p_, qs) for p_ in (pe, pr, ps))
    XR, XS = gpu.Ext(GE, GR), gpu.Ext(GE, GS)
    rng = np.random.default_rng(r + s + base)
    B, rel, L = 2, PR.n // PE.n, sum(sr.digit_counts(PS, base))
    c0, c1 = PR.random(rng, B), PR.random(rng, B)
    ys = np.stack([PS.random(rng, 1)[0] for _ in range(rel)])
    hints = np.stack([np.stack([np.stack([PS.random(rng, 1)[0] for _ in range(2)]) for _ in range(L)]) for _ in range(rel)])
    got = XR.tunnel(XS, c0, c1, ys, hints, base)
    assert np.array_equal(got, sr.tunnel(cpuref, PE, PR, PS, c0, c1, ys, hints, base))


# ------------------------------------------------------------------------------------
# GPU: liblolhip against the restatement, bit for bit
# ------------------------------------------------------------------------------------

@pytest.mark.gpu
@pytest.mark.parametrize("pps,qs", CASES)
def test_gpu_ctmul(gpu, cpuref, pps, qs):
    P, R = gpu.Plan(pps, qs), Params(pps, qs)
    rng = np.random.default_rng(21)
    for B in (1, 5):
        ops = [R.random(rng, B) for _ in range(4)]
        ops[0][0, 0] = np.array(qs) - 1
        got = P.ctMulCRT(*ops)
        want = sr.ctmul_crt(cpuref, R, *ops)
        for g_, w in zip(got, want):
            assert np.array_equal(g_, w)


@pytest.mark.gpu
@pytest.mark.parametrize("pps,qs", CASES)
@pytest.mark.parametrize("base", BASES)
def test_gpu_decompose_and_gadget(gpu, pps, qs, base):
    P, R = gpu.Plan(pps, qs), Params(pps, qs)
    rng = np.random.default_rng(base % 89 + 3)
    c = R.random(rng, 3)
    c[0, 0] = 0
    c[0, 1 % R.n] = np.array(qs) - 1
    c[1, 0] = np.array(qs) // 2
    c[1, 1 % R.n] = np.array(qs) // 2 - 1
    assert P.decomposeLen(base) == sum(sr.digit_counts(R, base))
    assert np.array_equal(P.gadget(base), sr.gadget(R, base))
    assert np.array_equal(P.decompose(c, base), sr.decompose(R, c, base))
    neg = np.where(c > 0, c - np.array(qs), 0)                   # reference-style (-q, 0] inputs
    assert np.array_equal(P.decompose(neg, base), sr.decompose(R, c, base))


@pytest.mark.gpu
@pytest.mark.parametrize("pps,qs", CASES)
@pytest.mark.parametrize("L,K", [(1, 1), (2, 2), (9, 2), (17, 3)])
def test_gpu_knapsack(gpu, cpuref, pps, qs, L, K):
    P, R = gpu.Plan(pps, qs), Params(pps, qs)
    rng = np.random.default_rng(L * 10 + K)
    B = 3
    xs = np.stack([R.random(rng, B) for _ in range(L)])
    hint = np.stack([np.stack([R.random(rng, 1)[0] for _ in range(K)]) for _ in range(L)])
    xs[0, 0], hint[0, 0] = np.array(qs) - 1, np.array(qs) - 1     # largest products
    add = np.stack([R.random(rng, B) for _ in range(K)])
    want = sr.knapsack(cpuref, R, xs, hint)
    assert np.array_equal(P.knapsack(xs, hint), want)
    qv = np.array(qs, dtype=object)
    assert np.array_equal(P.knapsack(xs, hint, addend=add), ((want.astype(object) + add) % qv).astype(np.int64))


@pytest.mark.gpu
@pytest.mark.parametrize("pps,qs", [c for c in CASES if c[0] != [(7, 1)]])
@pytest.mark.parametrize("base", [0, 2, 256])
def test_gpu_keyswitch(gpu, cpuref, pps, qs, base):
    P, R = gpu.Plan(pps, qs), Params(pps, qs)
    rng = np.random.default_rng(base + 40)
    B, L = 4, sum(sr.digit_counts(R, base))
    c2 = R.random(rng, B)
    hint = np.stack([np.stack([R.random(rng, 1)[0] for _ in range(2)]) for _ in range(L)])
    add = np.stack([R.random(rng, B) for _ in range(2)])
    want = sr.keyswitch(cpuref, R, c2, base, hint)
    assert np.array_equal(P.keySwitch(c2, base, hint), want)
    qv = np.array(qs, dtype=object)
    assert np.array_equal(P.keySwitch(c2, base, hint, addend=add), ((want.astype(object) + add) % qv).astype(np.int64))


@pytest.mark.gpu
@pytest.mark.parametrize("m,lower,T", [(15, 2 ** 20, 2), (45, 2 ** 29, 3), (48, 2 ** 26, 2), (1728, 2 ** 20, 2), (14400, 2 ** 19, 2),
                                       (11648, 2 ** 20, 2), (15015, 2 ** 29, 1), (105, 2 ** 30, 2)])
def test_gpu_keyswitch_fused_mixed_radix(gpu, cpuref, m, lower, T):
    """The one-pass key switch on the vector interpreter (k_mixed_keyswitch, arithmetic class 2: every modulus below
    2^30.15) at indices with odd prime powers — incl. the reference's own key-switch benchmark F64*F9*F25
    (Benchmarks/Default.hs:49-50, m = 14400) and every coefficients-per-thread variant (n = 8 ... 5760): TrivGad and two
    bases, with and without addends, (-q, 0] representatives of every operand; against the restatement and against
    the three-launch path.  (105 with a modulus above 2^30.15 stays on the three-launch path: same answers.)"""
    pps = lm.factor_pps(m)
    g = lm.good_qs(m, lower)
    qs = [next(g) for _ in range(T)]
    P, R = gpu.Plan(pps, qs), Params(pps, qs)
    rng = np.random.default_rng(m)
    qv = np.array(qs)
    for base, B in ((0, 3), (256, 2), (5, 1)):
        if R.n > 2000 and base == 5:
            continue                                          # the CPU restatement of 13+ digits at n = 3840 takes minutes
        Ld = sum(sr.digit_counts(R, base))
        c2 = R.random(rng, B)
        c2[0, 0] = qv - 1
        c2[0, 1] = qv // 2
        hint = np.stack([np.stack([R.random(rng, 1)[0] for _ in range(2)]) for _ in range(Ld)])
        add = np.stack([R.random(rng, B) for _ in range(2)])
        want = sr.keyswitch(cpuref, R, c2, base, hint)
        wadd = ((want.astype(object) + add) % np.array(qs, dtype=object)).astype(np.int64)
        assert np.array_equal(P.keySwitch(c2, base, hint), want), (m, base)
        assert np.array_equal(P.keySwitch(c2, base, hint, addend=add), wadd), (m, base, "addend")
        neg = lambda x: np.where(x > 0, x - qv, 0)
        assert np.array_equal(P.keySwitch(neg(c2), base, neg(hint), addend=neg(add)), wadd), (m, base, "negative")
        gpu.debug_set("KEYSWITCH_UNFUSED", True)
        assert np.array_equal(P.keySwitch(c2, base, hint, addend=add), wadd), (m, base, "three launches")
        gpu.debug_set("KEYSWITCH_UNFUSED", False)


@pytest.mark.gpu
@pytest.mark.parametrize("L", range(4, 15))
def test_gpu_keyswitch_fused_all_sizes(gpu, cpuref, monkeypatch, L):
    """Every n = 2^L the fused single-pass kernel instantiates (moduli < 2^30), TrivGad and two
    bases, odd batch sizes (ragged last workgroup), with and without addends; and the fused
    and the three-launch path agree."""
    m = 2 ** (L + 1)
    qs = _qs_for(m, 2 ** 20, 2) + _qs_for(m, 2 ** 29, 1)
    pps = [(2, L + 1)]
    P, R = gpu.Plan(pps, qs), Params(pps, qs)
    rng = np.random.default_rng(L)
    for base, B in ((0, 3), (4, 1), (1000, 9 if L < 12 else 2)):
        Ld = sum(sr.digit_counts(R, base))
        c2 = R.random(rng, B)
        c2[0, 0] = np.array(qs) - 1
        c2[0, 1] = np.array(qs) // 2
        hint = np.stack([np.stack([R.random(rng, 1)[0] for _ in range(2)]) for _ in range(Ld)])
        add = np.stack([R.random(rng, B) for _ in range(2)])
        want = sr.keyswitch(cpuref, R, c2, base, hint)
        got = P.keySwitch(c2, base, hint)
        assert np.array_equal(got, want), (L, base)
        wadd = ((want.astype(object) + add) % np.array(qs, dtype=object)).astype(np.int64)
        assert np.array_equal(P.keySwitch(c2, base, hint, addend=add), wadd), (L, base)
        gpu.debug_set("KEYSWITCH_UNFUSED", True)
        assert np.array_equal(P.keySwitch(c2, base, hint, addend=add), wadd), (L, base, "unfused")
        gpu.debug_set("KEYSWITCH_UNFUSED", False)
        # every operand may be a reference-style representative in (-q, 0]: the hint as well
        qv = np.array(qs)
        neg = lambda x: np.where(x > 0, x - qv, 0)
        assert np.array_equal(P.keySwitch(neg(c2), base, neg(hint), addend=neg(add)), wadd), (L, base, "negative fused")
        gpu.debug_set("KEYSWITCH_UNFUSED", True)
        assert np.array_equal(P.keySwitch(neg(c2), base, neg(hint), addend=neg(add)), wadd), (L, base, "negative unfused")
        gpu.debug_set("KEYSWITCH_UNFUSED", False)


@pytest.mark.gpu
@pytest.mark.parametrize("pps,qs", [c for c in CASES if len(c[1]) >= 2])
def test_gpu_rescale(gpu, pps, qs):
    P, R = gpu.Plan(pps, qs), Params(pps, qs)
    rng = np.random.default_rng(8)
    c = R.random(rng, 5)
    c[0, 0] = np.array(qs) - 1
    c[0, 1 % R.n] = np.array(qs) // 2
    assert np.array_equal(P.rescaleDropFirst(c), sr.rescale_drop_first(R, c))
    with pytest.raises(gpu.LolHipError):
        gpu.Plan(pps, qs[:1]).rescaleDropFirst(c[..., :1])       # nothing to drop


@pytest.mark.gpu
def test_gpu_pipeline_argument_checks(gpu):
    P = gpu.Plan([(2, 4)], [97])
    with pytest.raises(gpu.LolHipError):
        P.decomposeLen(1)
    with pytest.raises(gpu.LolHipError):
        P.decomposeLen(-3)
    assert P.decomposeLen(0) == 1 and P.decomposeLen(2) == 7 and P.decomposeLen(10) == 2   # 97 = 1100001b = "97"


@pytest.mark.gpu
def test_config5_keyswitch_full_shape(gpu, cpuref):
    """BASELINE config 5 (lol-apps Benchmarks/Default.hs:49): m' = 2048, q = (1017857, 1032193),
    TrivGad (L = 2), quadratic ciphertexts; keySwitchQuadCirc on a 4096-ciphertext shard in HBM.
    Sampled rows against the restatement; the noise-free-hint identity over the whole batch."""
    torch = pytest.importorskip("torch")
    pps = [(2, 11)]
    P, R = gpu.Plan(pps, Q5), Params(pps, Q5)
    B, L = 4096, 2
    gen = torch.Generator(device="cuda"); gen.manual_seed(5)
    qv = torch.tensor(Q5, dtype=torch.int64, device="cuda")
    rnd = lambda *shape: torch.stack([torch.randint(0, q, shape, dtype=torch.int64, device="cuda", generator=gen) for q in Q5], dim=-1)
    c0, c1, c2 = rnd(B, R.n), rnd(B, R.n), rnd(B, R.n)
    hint = rnd(L, 2, R.n)
    add = torch.stack([P.crt(c0.clone()), P.crt(c1.clone())])
    out = P.keySwitch(c2, 0, hint, addend=add)
    idx = [0, 1, B // 2, B - 1]
    want = sr.keyswitch(cpuref, R, c2[idx].cpu().numpy(), 0, hint.cpu().numpy())
    want = (want.astype(object) + add[:, idx].cpu().numpy()) % np.array(Q5, dtype=object)
    assert np.array_equal(out[:, idx].cpu().numpy(), want.astype(np.int64))
    # hint_j = (gadget_j * v, 0): the switched part equals c2 * v on every ciphertext of the shard
    v = rnd(1, R.n)
    v_crt = P.crt(v.clone())[0]
    g = torch.from_numpy(P.gadget(0)).cuda()
    h2 = torch.zeros_like(hint)
    for j in range(L):
        h2[j, 0] = v_crt * g[j] % qv
    sw = P.keySwitch(c2, 0, h2)
    prod = torch.empty_like(c2); P.polymul(c2, v.expand(B, R.n, 2).contiguous(), out=prod)
    assert torch.equal(sw[0], P.crt(prod)) and not bool(sw[1].any())


@pytest.mark.gpu
def test_config3_ctmul_full_shape(gpu, cpuref):
    """BASELINE config 3: m = 2^15, four ~59-bit moduli; the fused ciphertext product equals
    crt of (mulGPow of) the poly-mul composition over the batch, and sampled rows equal the
    restatement."""
    torch = pytest.importorskip("torch")
    m = 2 ** 15
    qs = _qs_for(m, 2 ** 59, 4)
    P, R = gpu.Plan([(2, 15)], qs), Params([(2, 15)], qs)
    B = 32
    gen = torch.Generator(device="cuda"); gen.manual_seed(3)
    rnd = lambda: torch.stack([torch.randint(0, q, (B, R.n), dtype=torch.int64, device="cuda", generator=gen) for q in qs], dim=-1)
    c0, c1, d0, d1 = rnd(), rnd(), rnd(), rnd()
    crt = lambda x: P.crt(x.clone())
    e0, e1, e2 = P.ctMulCRT(crt(c0), crt(c1), crt(d0), crt(d1))
    qv = torch.tensor(qs, dtype=torch.int64, device="cuda")
    pm = lambda a, b: P.polymul(a, b, out=torch.empty_like(a))
    w0, w2 = pm(c0, d0), pm(c1, d1)
    w1 = (pm(c0, d1) + pm(c1, d0)) % qv
    for e, w in ((e0, w0), (e1, w1), (e2, w2)):
        assert torch.equal(P.crtInv(e.clone()), P.mulGPow(w.clone()))
    idx = [0, B - 1]
    h = lambda x: cpuref.crt(R, x[idx].cpu().numpy()).reshape(len(idx), R.n, R.T)
    want = sr.ctmul_crt(cpuref, R, h(c0), h(c1), h(d0), h(d1))
    for e, w in zip((e0, e1, e2), want):
        assert np.array_equal(e[idx].cpu().numpy(), w)
