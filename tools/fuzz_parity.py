#!/usr/bin/env python3
"""tools/fuzz_parity.py — randomized GPU-vs-oracle parity campaign through the C ABI.
usage: fuzz_parity.py [seconds] [seed].  Prints one line per failing case and a summary;
exit status 1 if anything differed.  Development aid (the committed tests are the contract)."""
import os, sys, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lol_amd
from oracle import lolmath as lm, she_ref as sr
from oracle.oracle import CpuRef, Params

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rnd = random.Random(seed)
rng = np.random.default_rng(seed)
cpu = CpuRef()
PRIMES = [2, 3, 5, 7, 11, 13]
BIG = bool(os.environ.get("FUZZ_BIG"))          # only polynomials of 2048..16384 coefficients
fails = cases = 0
t_end = time.time() + budget
last = time.time()


def random_m():
    while True:
        pps = []
        for p in PRIMES:
            r = rnd.random()
            if p == 2:
                e = rnd.choice([0, 0, 1, 2, 3, 5, 6, 7, 9, 11, 12, 13, 14, 15]) if r < 0.8 else 0
            else:
                e = rnd.choice([0, 0, 0, 1, 1, 2]) if p <= 5 else rnd.choice([0, 0, 0, 1])
            if e:
                pps.append((p, e))
        if not pps:
            continue
        n = lm.totient_pps(pps)
        if (2048 if BIG else 2) <= n <= (16384 if BIG else 8192):
            return pps


def check(name, got, want, ctx):
    global fails
    ok = (got is None and want is None) or (got is not None and want is not None and np.array_equal(got, np.asarray(want).reshape(np.asarray(got).shape)))
    if not ok:
        fails += 1
        print("FAIL", name, ctx, flush=True)


while time.time() < t_end:
    pps = random_m()
    m = lm.value_pps(pps)
    T = rnd.choice([1, 1, 2, 3, 4])
    bits = [rnd.choice([14, 20, 29, 30, 31, 40, 59, 60, 61]) for _ in range(T)]
    if rnd.random() < 0.3:
        bits = [bits[0]] * T
    qs, used = [], set()
    for b in bits:
        g = lm.good_qs(m, 2 ** b + rnd.randrange(0, 1 << min(b - 1, 20)))
        q = next(g)
        while q in used:
            q = next(g)
        used.add(q); qs.append(q)
    B = rnd.choice([1, 2, 3, 5, 8, 9])
    R = Params(pps, qs)
    cap = 400000 if BIG else 60000
    if R.n * B * T > cap:
        B = max(1, cap // (R.n * T))
    ctx = (m, qs, B)
    try:
        P = lol_amd.Plan(pps, qs)
        y, z = R.random(rng, B), R.random(rng, B)
        check("crt", P.crt(y), cpu.crt(R, y), ctx)
        check("crtInv", P.crtInv(y), cpu.crtinv(R, y), ctx)
        check("polymul", P.polymul(y, z), cpu.polymul(R, y, z), ctx)
        check("mul", P.mul(y, z), cpu.mul(R, y, z), ctx)
        for op, oop in (("l", "l"), ("lInv", "linv"), ("mulGPow", "gpow"), ("mulGDec", "gdec"), ("divGPow", "ginvpow"), ("divGDec", "ginvdec")):
            check(op, getattr(P, op)(y), getattr(cpu, oop)(R, y), ctx)
        c = [R.random(rng, B) for _ in range(4)]
        got, want = P.ctMulCRT(*c), sr.ctmul_crt(cpu, R, *c)
        for k in range(3):
            check(f"ctmul{k}", got[k], want[k], ctx)
        base = rnd.choice([0, 0, 2, 3, 16, 256, 1000, 2 ** 20])
        Ld = sum(sr.digit_counts(R, base))
        if Ld <= 40:
            check("decompose", P.decompose(y, base), sr.decompose(R, y, base), ctx + (base,))
            hint = np.stack([np.stack([R.random(rng, 1)[0] for _ in range(2)]) for _ in range(Ld)])
            add = np.stack([R.random(rng, B) for _ in range(2)]) if rnd.random() < 0.5 else None
            want = sr.keyswitch(cpu, R, y, base, hint)
            if add is not None:
                want = ((want.astype(object) + add) % np.array(qs, dtype=object)).astype(np.int64)
            check("keyswitch", P.keySwitch(y, base, hint, addend=add), want, ctx + (base,))
        if T >= 2:
            check("rescale", P.rescaleDropFirst(y), sr.rescale_drop_first(R, y), ctx)
        # a random subring E of R (drop or lower some prime powers): twace / embed / coeffs
        sub = [(p, rnd.randint(0, e)) for p, e in pps]
        sub = [(p, e) for p, e in sub if e > 0]
        if sub != pps and R.n * B <= 20000 and all(lm.value_pps(pps) % q_ == 1 % q_ or True for q_ in qs):
            RE = Params(sub, qs)
            PE = lol_amd.Plan(sub, qs)
            X = lol_amd.Ext(PE, P)
            lo = RE.random(rng, B)
            check("embedPow", X.embedPow(lo), cpu.embed_pow(RE, R, lo), ctx + (sub,))
            check("embedDec", X.embedDec(lo), cpu.embed_dec(RE, R, lo), ctx + (sub,))
            check("embedCRT", X.embedCRT(lo), cpu.embed_crt(RE, R, lo), ctx + (sub,))
            check("twacePowDec", X.twacePowDec(y), cpu.twace_powdec(RE, R, y), ctx + (sub,))
            check("twaceCRT", X.twaceCRT(y), cpu.twace_crt(RE, R, y), ctx + (sub,))
            check("coeffs", X.coeffs(y), cpu.coeffs(RE, R, y), ctx + (sub,))
    except Exception as ex:      # noqa: BLE001
        fails += 1
        print("EXC", type(ex).__name__, ex, ctx, flush=True)
    cases += 1
    if time.time() - last > 30:
        print(f"... {cases} parameter sets, {fails} failures", flush=True)
        last = time.time()
print(f"fuzz: {cases} parameter sets, {fails} failures (seed {seed})", flush=True)
sys.exit(1 if fails else 0)
