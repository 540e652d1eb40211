"""oracle/she_model.py — a small executable model of lol-apps' SymmSHE on top of a Tensor ENGINE.
TEST INFRASTRUCTURE (only tests/ import this).

The reference's SHE layer exists only in Haskell and ships no vectors, so the ring-level
pipelines of include/lolhip.h (ct x ct, decompose, knapsack / key switch, rescale) cannot be pinned
by fixtures.  What the reference DOES fix is their contract: its own end-to-end properties
(lol-apps/Crypto/Lol/Applications/Tests/SHETests.hs:40-248): Dec . Enc = id, Dec (c * d) =
Dec c * Dec d, Dec (keySwitchQuadCirc hint c) = Dec c, and the same after modSwitch.  This module
restates encrypt / decrypt / (*) / toMSD / toLSD / ksQuadCircHint / keySwitchQuadCirc / modSwitch
from the cited lines of lol-apps/Crypto/Lol/Applications/SymmSHE.hs, with EVERY ring operation
delegated to an engine object — a lol_amd.Plan (the GPU executes them) or CpuEngine below (the
CPU oracle).  A convention error in digit order, hint layout [L][K][n][T], gadget, basis or
MSD/LSD handling breaks decryption instead of matching itself.

Elements of R'_q are int64 slabs [B][n][T] in the POWERFUL basis unless a name says otherwise.
Noise is small and bounded (uniform in {-1, 0, 1} per decoding-basis coefficient): the
properties hold for any small enough error, which is all they are used for here.
"""
from __future__ import annotations

from math import prod

import numpy as np

from . import she_ref as sr
from .oracle import CpuRef, Params


class CpuEngine:
    """The Tensor operations the model needs, from the CPU oracle (same method names as lol_amd.Plan)."""

    def __init__(self, cpu: CpuRef, P: Params):
        self.cpu, self.P = cpu, P
        self.n, self.T, self.qs = P.n, P.T, P.qs

    def _r(self, y): return np.asarray(y).reshape(-1, self.n, self.T)
    def crt(self, y): return self._r(self.cpu.crt(self.P, y))
    def crtInv(self, y): return self._r(self.cpu.crtinv(self.P, y))
    def l(self, y): return self._r(self.cpu.l(self.P, y))
    def lInv(self, y): return self._r(self.cpu.linv(self.P, y))
    def mul(self, a, b): return self._r(self.cpu.mul(self.P, a, b))
    def divGPow(self, y):
        r = self.cpu.ginvpow(self.P, y)
        return None if r is None else self._r(r)
    def ctMulCRT(self, c0, c1, d0, d1): return sr.ctmul_crt(self.cpu, self.P, c0, c1, d0, d1)
    def gadget(self, base): return sr.gadget(self.P, base)
    def decomposeLen(self, base): return sum(sr.digit_counts(self.P, base))

    def keySwitch(self, c2_pow, base, hint, addend=None):
        out = sr.keyswitch(self.cpu, self.P, c2_pow, base, hint)
        if addend is not None:
            out = ((out.astype(object) + np.asarray(addend)) % np.array(self.qs, dtype=object)).astype(np.int64)
        return out

    def rescaleDropFirst(self, c): return sr.rescale_drop_first(self.P, c)


class SHE:
    """SymmSHE over index m' = m (no ring switching here), plaintext modulus p, ciphertext moduli qs."""

    def __init__(self, eng, eng_p, qs, p, rng):
        self.e, self.ep, self.qs, self.p, self.rng = eng, eng_p, [int(q) for q in qs], int(p), rng
        self.n, self.T = eng.n, len(qs)
        self.Q = prod(self.qs)

    # ---- representation helpers -------------------------------------------------------------
    def reduce(self, x_int):
        """integers [B][n] -> residues [B][n][T]"""
        x = np.asarray(x_int).astype(object)
        return np.ascontiguousarray(np.stack([(x % q).astype(np.int64) for q in self.qs], axis=-1))

    def lift(self, x):
        """residues [B][n][T] -> centred integers [B][n] (CRT reconstruction over the T moduli)"""
        x = np.asarray(x).astype(object)
        acc = np.zeros(x.shape[:-1], dtype=object)
        for t, q in enumerate(self.qs):
            Qt = self.Q // q
            acc = acc + x[..., t] * (Qt * pow(Qt % q, -1, q))
        acc = acc % self.Q
        return np.where(2 * acc < self.Q, acc, acc - self.Q)

    def scal(self, x, k):
        """multiply by the integer scalar k in every component"""
        x = np.asarray(x).astype(object)
        return np.ascontiguousarray(np.stack([(x[..., t] * (k % q) % q).astype(np.int64) for t, q in enumerate(self.qs)], axis=-1))

    def add(self, a, b):
        return ((np.asarray(a).astype(object) + np.asarray(b).astype(object)) % np.array(self.qs, dtype=object)).astype(np.int64)

    def sub(self, a, b):
        return ((np.asarray(a).astype(object) - np.asarray(b).astype(object)) % np.array(self.qs, dtype=object)).astype(np.int64)

    def rmul(self, a, b):
        """ring product, powerful basis in and out: crtInv (crt a * crt b)"""
        return self.e.crtInv(self.e.mul(self.e.crt(a), self.e.crt(b)))

    def small_dec(self, B):
        """a small element given by decoding-basis coefficients in {-1,0,1}, returned in the powerful basis"""
        return self.e.l(self.reduce(self.rng.integers(-1, 2, size=(B, self.n))))

    def uniform(self, B):
        return np.ascontiguousarray(np.stack([self.rng.integers(0, q, size=(B, self.n), dtype=np.int64) for q in self.qs], axis=-1))

    # ---- SymmSHE.hs:120-146 ----------------------------------------------------------------
    def keygen(self):
        self.s = self.small_dec(1)                         # [1][n][T], powerful basis
        self.s_crt = self.e.crt(self.s)
        return self.s

    def _bs(self, x, B):
        return np.ascontiguousarray(np.broadcast_to(x, (B,) + x.shape[1:]))

    def encrypt(self, pt_pow):
        """SymmSHE.hs:138-146: e <- errorCoset (embed pt); c1 uniform; CT LSD 0 1 [reduce e - c1 s, c1].
        pt_pow: [B][n] residues mod p, powerful basis."""
        B = pt_pow.shape[0]
        # decoding-basis coefficients of the plaintext (lInv over the integers: exact on small values)
        pt_dec = self.lift(self.e.lInv(self.reduce(pt_pow))) % self.p
        pt_dec = np.where(2 * pt_dec < self.p, pt_dec, pt_dec - self.p)
        e_dec = pt_dec + self.p * self.rng.integers(-1, 2, size=(B, self.n)).astype(object)     # the coset pt + p R'
        e = self.e.l(self.reduce(e_dec))
        c1 = self.uniform(B)
        c0 = self.sub(e, self.rmul(c1, self._bs(self.s, B)))
        return {"enc": "LSD", "k": 0, "l": 1, "c": [c0, c1]}

    def evaluate(self, cs):
        """c(s) = sum_i c_i s^i, powerful basis"""
        B = cs[0].shape[0]
        s_crt = self._bs(self.s_crt, B)
        acc, pw = self.e.crt(cs[0]), None
        for c in cs[1:]:
            pw = s_crt if pw is None else self.e.mul(pw, s_crt)
            acc = self.add(acc, self.e.mul(self.e.crt(c), pw))
        return self.e.crtInv(acc)

    # ---- SymmSHE.hs:216-240, ZqBasic.hs:132-137 ---------------------------------------------
    def toMSD(self, ct):
        if ct["enc"] == "MSD":
            return ct
        zq = [pow(self.p % q, -1, q) for q in self.qs]               # recip (reduce p), per component
        c = [np.ascontiguousarray(np.stack([(x[..., t].astype(object) * zq[t] % q).astype(np.int64) for t, q in enumerate(self.qs)], axis=-1)) for x in ct["c"]]
        return {"enc": "MSD", "k": ct["k"], "l": ct["l"] * (-self.Q % self.p) % self.p, "c": c}

    def toLSD(self, ct):
        if ct["enc"] == "LSD":
            return ct
        c = [self.scal(x, self.p) for x in ct["c"]]
        return {"enc": "LSD", "k": ct["k"], "l": ct["l"] * pow(-self.Q % self.p, -1, self.p) % self.p, "c": c}

    def decrypt(self, ct):
        """SymmSHE.hs:165-178: lift the decoding-basis coefficients of c(s), reduce mod p, divide by
        g k times, scale by l.  Returns [B][n] residues mod p, powerful basis (m' = m: twace = id)."""
        ct = self.toLSD(ct)
        v_dec = self.lift(self.e.lInv(self.evaluate(ct["c"])))       # liftCyc Dec
        x = (v_dec % self.p).astype(np.int64)[..., None]             # R'_p, decoding basis
        x = self.ep.l(np.ascontiguousarray(x))                       # -> powerful basis of R'_p
        for _ in range(ct["k"]):
            x = self.ep.divGPow(x)
            assert x is not None, "divG failed"
        return (x[..., 0].astype(object) * ct["l"] % self.p).astype(np.int64)

    # ---- SymmSHE.hs:432-449 ------------------------------------------------------------------
    def mul(self, a, b):
        if a["enc"] == "MSD" and b["enc"] == "MSD":
            a = self.toLSD(a)
        if a["enc"] != "LSD":
            a, b = b, a
        assert len(a["c"]) == 2 and len(b["c"]) == 2
        e = self.e.ctMulCRT(*[self.e.crt(x) for x in (a["c"][0], a["c"][1], b["c"][0], b["c"][1])])
        return {"enc": b["enc"], "k": a["k"] + b["k"] + 1, "l": a["l"] * b["l"] % self.p, "c": [self.e.crtInv(x) for x in e]}

    # ---- SymmSHE.hs:262-300, 345-371 ---------------------------------------------------------
    def ks_quad_hint(self, base):
        """ksHint sk (s*s): hint_j = const (g_j s^2) + [c1_j (-s) + e_j, c1_j], CRT basis, [L][2][n][T]"""
        g = self.e.gadget(base)                                      # [L][T]
        s2_crt = self.e.mul(self.s_crt, self.s_crt)
        rows = []
        for j in range(g.shape[0]):
            c1 = self.uniform(1)                                     # uniform in the CRT basis is uniform
            err = self.e.crt(self.small_dec(1))
            gs2 = np.ascontiguousarray(np.stack([(s2_crt[..., t].astype(object) * int(g[j, t]) % q).astype(np.int64) for t, q in enumerate(self.qs)], axis=-1))
            h0 = self.add(self.add(gs2, err), self.sub(np.zeros_like(c1), self.e.mul(c1, self.s_crt)))
            rows.append(np.stack([h0[0], c1[0]]))
        return np.ascontiguousarray(np.stack(rows))

    def key_switch_quad(self, hint, base, ct):
        ct = self.toMSD(ct)
        c0, c1, c2 = ct["c"]
        add = np.ascontiguousarray(np.stack([self.e.crt(c0), self.e.crt(c1)]))
        out = self.e.keySwitch(c2, base, hint, addend=add)           # [2][B][n][T], CRT basis
        return {"enc": "MSD", "k": ct["k"], "l": ct["l"], "c": [self.e.crtInv(np.ascontiguousarray(out[0])), self.e.crtInv(np.ascontiguousarray(out[1]))]}

    # ---- SymmSHE.hs:232-246, Cyc.hs:529-542 --------------------------------------------------
    def mod_switch_drop_first(self, ct, eng2, eng2_p=None):
        """modSwitch to the modulus without its first component: c0 rescaled in the decoding basis,
        the others in the powerful basis.  Returns (ciphertext, SHE over the remaining moduli)."""
        ct = self.toMSD(ct)
        she2 = SHE(eng2, eng2_p or self.ep, self.qs[1:], self.p, self.rng)
        she2.s = np.ascontiguousarray(self.s[..., 1:])
        she2.s_crt = eng2.crt(she2.s)
        c = [eng2.l(self.e.rescaleDropFirst(self.e.lInv(ct["c"][0])))] + [self.e.rescaleDropFirst(x) for x in ct["c"][1:]]
        return {"enc": "MSD", "k": ct["k"], "l": ct["l"], "c": c}, she2


# =============================================================================================
# ring tunnelling (SymmSHE.hs:500-570, Linear.hs:55-119) with r' = r, s' = s (so e' = e)
# =============================================================================================
class CpuTunnelEngine:
    """The operations between E = O_e, R = O_r and S = O_s the tunnel needs, from the CPU oracle
    (same method names as TunnelEngine over lol_amd.Ext in tests/test_she_properties.py)."""

    def __init__(self, cpu: CpuRef, PE: Params, PR: Params, PS: Params):
        self.cpu, self.PE, self.PR, self.PS = cpu, PE, PR, PS

    def evalLin(self, r_dec, ys_crt): return sr.evallin(self.cpu, self.PE, self.PR, self.PS, r_dec, ys_crt)
    def tunnel(self, c0_dec, c1_pow, ys_crt, hints, base): return sr.tunnel(self.cpu, self.PE, self.PR, self.PS, c0_dec, c1_pow, ys_crt, hints, base)


def ks_hint(she: SHE, value_crt, base):
    """ksHint skout r (SymmSHE.hs:262-300): hint_j = const (g_j r) + [c1_j (-s) + e_j, c1_j], CRT basis, [L][2][n][T].
    value_crt [1][n][T]: the element the hint lets one multiply by, CRT basis, under she's secret key."""
    g = she.e.gadget(base)
    rows = []
    for j in range(g.shape[0]):
        c1 = she.uniform(1)
        err = she.e.crt(she.small_dec(1))
        gv = np.ascontiguousarray(np.stack([(value_crt[..., t].astype(object) * int(g[j, t]) % q).astype(np.int64) for t, q in enumerate(she.qs)], axis=-1))
        h0 = she.add(she.add(gv, err), she.sub(np.zeros_like(c1), she.e.mul(c1, she.s_crt)))
        rows.append(np.stack([h0[0], c1[0]]))
    return np.ascontiguousarray(np.stack(rows))


def tunnel_hint(she_in: SHE, she_out: SHE, xeng, rel_pow_index, f_vals_p, base):
    """tunnelHint f skout skin (SymmSHE.hs:531-545).  f is given by its values f_vals_p [rel][n_S] (residues mod p,
    powerful basis of S) on the relative decoding basis of R/E (linearDec, Linear.hs:65-72).
      f'  = lift f      : liftPow of every value (Linear.hs:104-107), then reduced mod q -> ys (CRT basis of S_q)
      ps  = powBasis    : the relative powerful basis of R/E, p_i = the powerful-basis unit vector of R at
                          rel_pow_index[i] (Tensor.hs:472-477 pairs index (i, 0) with it)
      comps_i = evalLin f' (s_in * p_i);  hints_i = ksHint skout comps_i
    Returns (ys_crt [rel][n_S][T], hints [rel][L][2][n_S][T])."""
    p = she_in.p
    v = np.asarray(f_vals_p).astype(object) % p
    v = np.where(2 * v < p, v, v - p)                                    # liftPow: centred representatives
    ys_crt = she_out.e.crt(she_out.reduce(v))                            # [rel][n_S][T]
    hints = []
    for i, idx in enumerate(rel_pow_index):
        pi = np.zeros((1, she_in.n, she_in.T), dtype=np.int64)
        pi[0, idx, :] = 1
        sp_dec = she_in.e.lInv(she_in.rmul(she_in.s, pi))                # s_in * p_i, decoding basis of R
        comp_crt = xeng.evalLin(sp_dec, ys_crt)                          # [1][n_S][T], CRT basis of S
        hints.append(ks_hint(she_out, comp_crt, base))
    return ys_crt, np.ascontiguousarray(np.stack(hints))


def tunnel(she_in: SHE, xeng, ys_crt, hints, base, ct):
    """tunnel (SymmSHE.hs:549-570): toMSD . absorbGFactors (a ciphertext with k = 0 here), c0' = evalLin f'q c0,
    c1' = sum_i switch hints_i (embed (coeffsPow_i c1)) — the fused lolhip_tunnel_batch / its restatement."""
    ct = she_in.toMSD(ct)
    assert ct["k"] == 0 and len(ct["c"]) == 2
    c0_dec = she_in.e.lInv(ct["c"][0])
    out = xeng.tunnel(c0_dec, ct["c"][1], ys_crt, hints, base)           # [2][B][n_S][T], CRT basis of S
    return {"enc": "MSD", "k": 0, "l": ct["l"], "c": out}                # still in the CRT basis: the caller converts
