"""oracle/she_ref.py — CPU restatement of the ring-level pipelines SymmSHE builds on top of
the Tensor ops.  TEST INFRASTRUCTURE: only tests/ and smoke() import this.

These functions exist only in Haskell in the reference (no GHC in the image),
so each is restated from the source lines cited and pinned by the reference's own algebraic
contracts in tests/ (gadget law <g, decompose x> = x; rescale exactness on multiples of q_a;
ct x ct against the ring product of the two ciphertext polynomials).  Every transform and
pointwise product inside goes through oracle.CpuRef, which IS pinned against the reference's
C++ (tests/golden).  Slab layout everywhere: int64 [B][n][T], component t innermost.

Arithmetic is exact Python/NumPy integer arithmetic: object arrays wherever a product can
exceed 63 bits.
"""
from __future__ import annotations

import numpy as np

from . import lolmath as lm
from .oracle import CpuRef, Params


def _obj(a):
    return np.asarray(a).astype(object)


def _qs(P: Params):
    return np.array(P.qs, dtype=object)


# ---- lift / divModCent / decomp ---------------------------------------------------------

def lift_centered(x, q: int):
    """ZqBasic.hs:92-94 (decode'): representative in [-q/2, q/2): x if 2x < q else x - q."""
    x = _obj(x) % q
    return np.where(2 * x < q, x, x - q)


def div_mod_cent(a, b: int):
    """Numeric.hs:227-234: (quotient, remainder) with the remainder in [-b/2, b/2);
    floor division of a + b div 2."""
    shift = b // 2
    a = _obj(a) + shift
    return a // b, a % b - shift


def gadlen(b: int, q: int) -> int:
    """ZqBasic.hs:238-240: number of base-b digits of q."""
    return 0 if q == 0 else 1 + gadlen(b, q // b)


def digit_counts(P: Params, base: int) -> list[int]:
    """digits per RNS component: 1 for TrivGad (base 0, ZqBasic.hs:227-232), gadlen otherwise
    (ZqBasic.hs:258-264)."""
    return [1 if base == 0 else gadlen(base, q) for q in P.qs]


def gadget(P: Params, base: int):
    """The gadget vector of the product ring as [L][T] residues: b^k in its own component,
    zero in the others (Gadget.hs:92-94 over ZqBasic.hs:227-229,248-253)."""
    rows = []
    for t, (q, k) in enumerate(zip(P.qs, digit_counts(P, base))):
        for j in range(k):
            row = [0] * P.T
            row[t] = 1 if base == 0 else pow(base, j, q)
            rows.append(row)
    return np.array(rows, dtype=np.int64).reshape(-1, P.T)


def decompose(P: Params, c, base: int):
    """Cyc.hs:592-604 (powerful-basis coefficients) over Gadget.hs:96-101 (components
    concatenated) over ZqBasic.hs:231-232 / 258-264 (lift, then centred base-b digits,
    Numeric.hs:202-205), then `fmap reduce` of SymmSHE.hs:314: every integer digit polynomial
    reduced into all T components.  c: [B][n][T] -> [L][B][n][T] canonical residues."""
    c = np.asarray(c)
    out = []
    for t, (q, k) in enumerate(zip(P.qs, digit_counts(P, base))):
        v = lift_centered(c[..., t], q)
        digits = []
        if base == 0:
            digits = [v]
        else:
            for _ in range(k - 1):
                v, r = div_mod_cent(v, base)
                digits.append(r)
            digits.append(v)
        for d in digits:
            out.append(np.stack([(d % qs).astype(np.int64) for qs in P.qs], axis=-1))
    return np.ascontiguousarray(np.stack(out, axis=0))


# ---- knapsack / key switch --------------------------------------------------------------

def _addmod(a, b, P: Params):
    return ((_obj(a) + _obj(b)) % _qs(P)).astype(np.int64)


def knapsack(cpu: CpuRef, P: Params, xs_crt, hint):
    """SymmSHE.hs:302-304: sum_j x_j *>> hint_j with x_j, hint_jk in the CRT basis.
    xs_crt [L][B][n][T], hint [L][K][n][T] (shared by the batch) -> [K][B][n][T]."""
    xs_crt, hint = np.asarray(xs_crt), np.asarray(hint)
    L, B = xs_crt.shape[0], xs_crt.shape[1]
    K = hint.shape[1]
    out = np.zeros((K, B, P.n, P.T), dtype=np.int64)
    for j in range(L):
        for k in range(K):
            h = np.ascontiguousarray(np.broadcast_to(hint[j, k], (B, P.n, P.T)))
            out[k] = _addmod(out[k], cpu.mul(P, np.ascontiguousarray(xs_crt[j]), h).reshape(B, P.n, P.T), P)
    return out


def keyswitch(cpu: CpuRef, P: Params, c2_pow, base: int, hint):
    """SymmSHE.hs:312-314 `switch`: knapsack hint (crt . reduce <$> decompose c)."""
    d = decompose(P, c2_pow, base)
    L, B = d.shape[0], d.shape[1]
    d_crt = cpu.crt(P, np.ascontiguousarray(d.reshape(L * B, P.n, P.T))).reshape(L, B, P.n, P.T)
    return knapsack(cpu, P, d_crt, hint)


# ---- ciphertext product -----------------------------------------------------------------

def ctmul_crt(cpu: CpuRef, P: Params, c0, c1, d0, d1):
    """SymmSHE.hs:444-449: coefficients of mulG <$> (c * d) for two linear ciphertexts, every
    operand in the CRT basis: (g c0 d0, g (c0 d1 + c1 d0), g c1 d1); mulG in the CRT basis is
    the pointwise product with gCRT (CPP.hs:230)."""
    g = np.stack([np.array(lm.g_crt(P.pps, q), dtype=np.int64) for q in P.qs], axis=-1)   # [n][T]
    B = np.asarray(c0).shape[0]
    gb = np.ascontiguousarray(np.broadcast_to(g, (B, P.n, P.T)))
    mul = lambda a, b: cpu.mul(P, np.ascontiguousarray(a), np.ascontiguousarray(b)).reshape(B, P.n, P.T)
    e0 = mul(gb, mul(c0, d0))
    e1 = mul(gb, _addmod(mul(c0, d1), mul(c1, d0), P))
    e2 = mul(gb, mul(c1, d1))
    return e0, e1, e2


# ---- modulus rescaling ------------------------------------------------------------------

def rescale_drop_first(P: Params, c):
    """Cyc.hs:529-542, RescaleCyc (a,b) -> b: with (a, b) = unzip c and z = lift a
    (coefficient-wise, basis Pow or Dec), the result is q_a^-1 * (b - reduce z) in every
    remaining component.  c: [B][n][T] -> [B][n][T-1]."""
    c = np.asarray(c)
    qa = P.qs[0]
    z = lift_centered(c[..., 0], qa)
    cols = []
    for s in range(1, P.T):
        q = P.qs[s]
        inv = pow(qa % q, -1, q)
        cols.append((((_obj(c[..., s]) - z) % q) * inv % q).astype(np.int64))
    return np.ascontiguousarray(np.stack(cols, axis=-1))


# ---- E-linear functions (tunnelling) ----------------------------------------------------

def evallin(cpu: CpuRef, PE: Params, PR: Params, PS: Params, r_dec, ys_crt):
    """lol Linear.hs:75-79: evalLin (RD ys) r = sum (zipWith (*) ys (embed <$> coeffsDec r)).
    r_dec [B][n_R][T] (decoding basis of R); ys_crt [n_R/n_E][n_S][T] (CRT basis of S, as
    linearDec stores them, Linear.hs:68-72) -> [B][n_S][T] in the CRT basis of S.  embed of a
    decoding-basis element is embedDec; the product needs the CRT basis: l (Dec -> Pow), crt."""
    r_dec = np.asarray(r_dec)
    B = r_dec.shape[0]
    cs = cpu.coeffs(PE, PR, r_dec)                                       # [rel][B][n_E][T]
    acc = np.zeros((B, PS.n, PS.T), dtype=np.int64)
    for i in range(cs.shape[0]):
        e = cpu.embed_dec(PE, PS, cs[i])
        e = cpu.crt(PS, cpu.l(PS, np.ascontiguousarray(e))).reshape(B, PS.n, PS.T)
        y = np.ascontiguousarray(np.broadcast_to(np.asarray(ys_crt)[i], (B, PS.n, PS.T)))
        acc = _addmod(acc, cpu.mul(PS, np.ascontiguousarray(e), y).reshape(B, PS.n, PS.T), PS)
    return acc


def tunnel(cpu: CpuRef, PE: Params, PR: Params, PS: Params, c0_dec, c1_pow, ys_crt, hints, base: int):
    """SymmSHE.hs:549-570 after toMSD . absorbGFactors:
         c0' = evalLin f c0;  c1s = coeffsPow c1;  c1' = sum (zipWith switch hints (embed <$> c1s))
         result = const c0' + c1'
    c0_dec, c1_pow [B][n_R][T]; hints [n_R/n_E][L][2][n_S][T] (CRT basis) -> [2][B][n_S][T], CRT basis of S'."""
    B = np.asarray(c0_dec).shape[0]
    out = np.zeros((2, B, PS.n, PS.T), dtype=np.int64)
    out[0] = evallin(cpu, PE, PR, PS, c0_dec, ys_crt)
    cs = cpu.coeffs(PE, PR, c1_pow)                                      # [rel][B][n_E][T]
    for i in range(cs.shape[0]):
        emb = cpu.embed_pow(PE, PS, cs[i]).reshape(B, PS.n, PS.T)
        sw = keyswitch(cpu, PS, np.ascontiguousarray(emb), base, np.asarray(hints)[i])
        out = ((out.astype(object) + sw) % _qs(PS)).astype(np.int64)
    return out
