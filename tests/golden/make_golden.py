#!/usr/bin/env python3
"""tests/golden/make_golden.py — generate known-answer vectors from the REFERENCE ITSELF.

Runs the reference's own lol-cpp C++ (oracle/_ref/libctensor.so, built from
/root/reference by oracle/Makefile) on seeded inputs and stores inputs + outputs as
data in tests/golden/golden_ct.npz.  Only runs where /root/reference exists; the
fixtures are what travels.  The reference has no golden vectors of its own
(every test is a QuickCheck property, lol/Crypto/Lol/Tests/TensorTests.hs:38-78),
so the parameter grid below is the one its tests use
(lol/Crypto/Lol/Tests/Default.hs:45-77), plus BASELINE.json's config 1 exactly and
config 2 / config 4 at moduli < 2^31 where CT is itself correct.

twace*/embed* exist only in Haskell (Extension.hs:54-129).  Their vectors are produced
through the reference's own identities (TensorTests.hs:133-234), with every CRT inside
them computed by the reference's C++:
    embedCRT  = crt_m' . embedPow . crtInv_m         twaceCRT = crt_m . twacePowDec . crtInv_m'
    embedDec  = lInv_m' . embedPow . l_m
so the index tables are pinned against real CT transforms, not only against our own
restatement of Tensor.hs:390-509.

Usage:  python tests/golden/make_golden.py          (rewrites golden_ct.npz)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import lolmath as lm  # noqa: E402
from oracle.oracle import CTRef, Params, build  # noqa: E402

ZQ1 = [18869761]
ZQ2 = [19393921, 18869761]
ZQ3 = [19918081, 19393921, 18869761]
SMOOTH1 = [2148249601]
SMOOTH3 = [2148854401, 2148249601, 2150668801]

# lol/Crypto/Lol/Tests/Default.hs:48-63
TENSOR1 = [(7, [29]), (12, SMOOTH1), (1, [17]), (2, [17]), (4, [17]), (8, [17]), (21, [8191]),
           (42, [8191]), (42, ZQ1), (2, ZQ2), (3, ZQ2), (7, ZQ2), (6, ZQ2), (42, SMOOTH3),
           (42, ZQ2), (89, [179])]
# lol/Crypto/Lol/Tests/Default.hs:65-77
TENSOR2 = [(1, 7, [29]), (4, 12, [536871001]), (4, 12, SMOOTH1), (2, 8, [17]), (8, 8, [17]),
           (2, 8, SMOOTH1), (4, 8, [17]), (3, 21, [8191]), (7, 21, [8191]), (3, 42, [8191]),
           (3, 21, ZQ1), (7, 21, ZQ2), (3, 42, ZQ3)]
# non-CRT moduli used by the Cyc tests for the Pow/Dec-basis operators (Default.hs:80-125)
PRIMEOPS_ONLY = [(7, [32]), (42, [1024]), (28, [8]), (91, [4]), (448, [16])]
# BASELINE.json configs at moduli where CT is a valid oracle
BIG = [(1024, [12289], 1), (2 ** 14, [1073872897], 2), (15015, [1073842771], 4)]

PRIME_OPS = ("l", "linv", "gpow", "gdec", "ginvpow", "ginvdec")


def pack(a, qs):
    a = np.asarray(a)
    return a.astype(np.uint32) if max(qs) < 2 ** 32 else a.astype(np.int64)


def main():
    build(ref=True)
    ct = CTRef()
    out = {}
    manifest = []
    rng = np.random.default_rng(20261004)

    def put(key, arr, qs):
        out[key] = pack(arr, qs)

    for i, (m, qs) in enumerate(TENSOR1):
        P = Params(lm.factor_pps(m), qs)
        y, z = P.random(rng, 2), P.random(rng, 2)
        tag = f"t1_{i}"
        manifest.append(f"{tag} m={m} qs={qs}")
        put(f"{tag}/y", y, qs)
        put(f"{tag}/z", z, qs)
        put(f"{tag}/crt", ct.crt(P, y), qs)
        put(f"{tag}/crtinv", ct.crtinv(P, y), qs)
        put(f"{tag}/mul", ct.mul(P, y, z), qs)
        put(f"{tag}/polymul", ct.polymul(P, y, z), qs)
        for op in PRIME_OPS:
            r = getattr(ct, op)(P, y)
            if r is not None:
                put(f"{tag}/{op}", r, qs)

    for i, (m, qs) in enumerate(PRIMEOPS_ONLY):
        pps = lm.factor_pps(m)
        n = lm.totient_pps(pps)
        y = np.stack([rng.integers(0, q, size=(2, n), dtype=np.int64) for q in qs], axis=-1)
        P = Params.__new__(Params)   # no CRT basis for these moduli: skip twiddle generation
        P.pps, P.qs, P.T, P.m, P.n = pps, qs, len(qs), m, n
        tag = f"po_{i}"
        manifest.append(f"{tag} m={m} qs={qs}")
        put(f"{tag}/y", y, qs)
        for op in PRIME_OPS:
            r = getattr(ct, op)(P, y)
            if r is not None:
                put(f"{tag}/{op}", r, qs)

    for i, (m, m2, qs) in enumerate(TENSOR2):
        a, b = lm.factor_pps(m), lm.factor_pps(m2)
        Pl, Ph = Params(a, qs), Params(b, qs)
        lo, hi = Pl.random(rng, 2), Ph.random(rng, 2)
        tag = f"t2_{i}"
        manifest.append(f"{tag} m={m} m'={m2} qs={qs}")

        def per_comp(arr, nout, f):
            res = np.zeros((arr.shape[0], nout, len(qs)), dtype=np.int64)
            for bb in range(arr.shape[0]):
                for t in range(len(qs)):
                    res[bb, :, t] = f([int(v) for v in arr[bb, :, t]])
            return res
        put(f"{tag}/lo", lo, qs)
        put(f"{tag}/hi", hi, qs)
        emb_pow = lambda x: per_comp(x, Ph.n, lambda v: lm.embed_pow(a, b, v))   # noqa: E731
        tw_pd = lambda x: per_comp(x, Pl.n, lambda v: lm.twace_powdec(a, b, v))  # noqa: E731
        put(f"{tag}/embed_pow", emb_pow(lo), qs)
        put(f"{tag}/twace_powdec", tw_pd(hi), qs)
        put(f"{tag}/embed_crt", ct.crt(Ph, emb_pow(ct.crtinv(Pl, lo))), qs)
        put(f"{tag}/twace_crt", ct.crt(Pl, tw_pd(ct.crtinv(Ph, hi))), qs)
        put(f"{tag}/embed_dec", ct.linv(Ph, emb_pow(ct.l(Pl, lo))), qs)

    for i, (m, qs, seed) in enumerate(BIG):
        P = Params(lm.factor_pps(m), qs)
        r = np.random.default_rng(seed)          # SURVEY.md 8d seeds
        y, z = P.random(r, 1), P.random(r, 1)
        tag = f"big_{i}"
        manifest.append(f"{tag} m={m} qs={qs} seed={seed}")
        put(f"{tag}/y", y, qs)
        put(f"{tag}/z", z, qs)
        put(f"{tag}/crt", ct.crt(P, y), qs)
        put(f"{tag}/crtinv", ct.crtinv(P, y), qs)
        put(f"{tag}/polymul", ct.polymul(P, y, z), qs)

    out["manifest"] = np.array("\n".join(manifest))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_ct.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {len(out)} arrays, {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
