"""Ad-hoc GPU parity sweep (development aid; the formal suite is tests/ -m gpu)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lol_amd
from lol_amd import Plan, Ext
from oracle.oracle import Params, CpuRef
from oracle import lolmath as lm

cr = CpuRef()
rng = np.random.default_rng(7)
fails = 0
def check(name, got, exp, ctx):
    global fails
    if exp is None or got is None:
        ok = (exp is None and got is None)
    else:
        ok = np.array_equal(np.asarray(got).reshape(np.asarray(exp).shape), exp)
    if not ok:
        fails += 1
        print("FAIL", name, ctx); sys.stdout.flush()

ms = [1,2,3,4,6,7,8,12,21,42,89,9,27,25,45,32,64,128,256,512,1024,2048,4096,280,1155,15015,2**14,2**15, 64*27, 64*9*25]
for m in ms:
    pps = lm.factor_pps(m)
    for T, lower in ((1, 2**20), (2, 2**59), (3, 1000 if m < 5000 else 2**25)):
        g = lm.good_qs(m, lower); qs = [next(g) for _ in range(T)]
        t0 = time.time()
        P = Plan(pps, qs); R = Params(pps, qs)
        B = 3 if P.n <= 4096 else 2
        y = R.random(rng, B); z = R.random(rng, B)
        ctx = (m, qs)
        check("crt", P.crt(y), cr.crt(R, y), ctx)
        check("crtInv", P.crtInv(y), cr.crtinv(R, y), ctx)
        check("mul", P.mul(y, z), cr.mul(R, y, z), ctx)
        check("polymul", P.polymul(y, z), cr.polymul(R, y, z), ctx)
        for nm, rn in (("l","l"),("lInv","linv"),("mulGPow","gpow"),("mulGDec","gdec"),("divGPow","ginvpow"),("divGDec","ginvdec")):
            check(nm, getattr(P, nm)(y), getattr(cr, rn)(R, y), ctx)
        # mulGCRT = crt . mulGPow . crtInv  (TensorTests.hs:107-112)
        check("mulGCRT", P.mulGCRT(y), cr.crt(R, cr.gpow(R, cr.crtinv(R, y))), ctx)
        gi = cr.ginvpow(R, cr.crtinv(R, y))
        check("divGCRT", P.divGCRT(y), None if gi is None else cr.crt(R, gi), ctx)
        print("m=%d T=%d n=%d  %.2fs fails=%d" % (m, T, P.n, time.time() - t0, fails)); sys.stdout.flush()

for (m, m2) in [(4,12),(3,21),(7,21),(1,8),(8,8),(12,60),(9,45),(56,2912),(128,11648//7)]:
    a, b = lm.factor_pps(m), lm.factor_pps(m2)
    g = lm.good_qs(m2, 2**30); qs = [next(g), next(g)]
    Pl, Ph = Plan(a, qs), Plan(b, qs); X = Ext(Pl, Ph)
    Rl, Rh = Params(a, qs), Params(b, qs)
    lo = Rl.random(rng, 2); hi = Rh.random(rng, 2)
    ctx = (m, m2)
    check("embedPow", X.embedPow(lo), cr.embed_pow(Rl, Rh, lo), ctx)
    check("embedDec", X.embedDec(lo), cr.embed_dec(Rl, Rh, lo), ctx)
    check("embedCRT", X.embedCRT(lo), cr.embed_crt(Rl, Rh, lo), ctx)
    check("twacePowDec", X.twacePowDec(hi), cr.twace_powdec(Rl, Rh, hi), ctx)
    check("twaceCRT", X.twaceCRT(hi), cr.twace_crt(Rl, Rh, hi), ctx)
    # reference identities (TensorTests.hs:133-234)
    check("embedCRT=crt.embedPow.crtInv", X.embedCRT(lo), cr.crt(Rh, cr.embed_pow(Rl, Rh, cr.crtinv(Rl, lo))), ctx)
    check("twaceCRT=crt.twacePowDec.crtInv", X.twaceCRT(hi), cr.crt(Rl, cr.twace_powdec(Rl, Rh, cr.crtinv(Rh, hi))), ctx)
    check("embedDec=lInv.embedPow.l", X.embedDec(lo), cr.linv(Rh, cr.embed_pow(Rl, Rh, cr.l(Rl, lo))), ctx)
    print("ext", m, m2, "fails=%d" % fails); sys.stdout.flush()
print("TOTAL FAILS", fails)
sys.exit(1 if fails else 0)
