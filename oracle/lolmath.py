"""oracle/lolmath.py — host-side number theory and index tables of Lol, restated in
plain Python big-ints.  TEST INFRASTRUCTURE ONLY: the product computes the same
tables in C++ (lol_amd/csrc/hostmath.cpp); tests compare the two.

Reference (all under /root/reference/):
  goodQs                 lol/Crypto/Lol/Types/Unsafe/ZqBasic.hs:71-73
  principalRootUnity     lol/Crypto/Lol/Types/Unsafe/ZqBasic.hs:144-165
  mhatInv                lol/Crypto/Lol/Types/Unsafe/ZqBasic.hs:167-171
  valueHat               lol/Crypto/Lol/FactoredDefs.hs:444-445
  ru / ruInv             lol-cpp/Crypto/Lol/Cyclotomic/Tensor/CPP.hs:422-442
  gCRTK / gInvCRTK       lol/Crypto/Lol/Cyclotomic/Tensor.hs:290-337
  digitRev, indexToPow/Zms lol/Crypto/Lol/Cyclotomic/Tensor.hs:342-379
  toIndexPair ... totients lol/Crypto/Lol/Cyclotomic/Tensor.hs:390-509
  embed*/twace* gathers  lol-cpp/Crypto/Lol/Cyclotomic/Tensor/CPP/Extension.hs:54-129

Third-party note: Lol finds "the smallest generator of Z_q^*" with arithmoi
(`factorise`, `isPrime`; lol/lol.cabal pins arithmoi >=0.4.1.3 && <0.5), which is
not under /root/reference.  The result is mathematically determined, so any
correct factoriser reproduces it; no reference test pins the numeric value of
omega — "parity unpinned" at that one boundary.
"""
from __future__ import annotations

import math
from functools import lru_cache

# ----------------------------------------------------------------------------
# primes / factoring (any correct algorithm; stands in for arithmoi)
# ----------------------------------------------------------------------------

_MR_BASES = (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37)  # deterministic for n < 3.3e24


def is_prime(n: int) -> bool:
    if n < 2:
        return False
    for p in _MR_BASES:
        if n % p == 0:
            return n == p
    d, s = n - 1, 0
    while d % 2 == 0:
        d //= 2
        s += 1
    for a in _MR_BASES:
        x = pow(a, d, n)
        if x in (1, n - 1):
            continue
        for _ in range(s - 1):
            x = x * x % n
            if x == n - 1:
                break
        else:
            return False
    return True


def _pollard_rho(n: int) -> int:
    if n % 2 == 0:
        return 2
    c = 1
    while True:
        x = y = 2
        d = 1
        while d == 1:
            x = (x * x + c) % n
            y = (y * y + c) % n
            y = (y * y + c) % n
            d = math.gcd(abs(x - y), n)
        if d != n:
            return d
        c += 1


def prime_factors(n: int) -> list[int]:
    """Sorted distinct prime factors."""
    out: set[int] = set()
    stack = [n]
    while stack:
        k = stack.pop()
        if k == 1:
            continue
        if is_prime(k):
            out.add(k)
            continue
        for p in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31):
            if k % p == 0:
                out.add(p)
                while k % p == 0:
                    k //= p
                stack.append(k)
                break
        else:
            d = _pollard_rho(k)
            stack.extend((d, k // d))
    return sorted(out)


def factor_pps(m: int) -> list[tuple[int, int]]:
    """[(p,e)] ascending primes == ppsFact (FactoredDefs.hs:360-361)."""
    out = []
    for p in prime_factors(m):
        e = 0
        while m % p == 0:
            m //= p
            e += 1
        out.append((p, e))
    return out


def value_pps(pps) -> int:
    return math.prod(p ** e for p, e in pps)


def totient_pp(pe) -> int:
    p, e = pe
    return 1 if e == 0 else (p - 1) * p ** (e - 1)


def totient_pps(pps) -> int:
    return math.prod(totient_pp(pe) for pe in pps)


def value_hat(m: int) -> int:
    """FactoredDefs.hs:444-445: m/2 if m even else m."""
    return m // 2 if m % 2 == 0 else m


def odd_rad(pps) -> int:
    return math.prod(p for p, _ in pps if p != 2)


def good_qs(m: int, lower: int):
    """ZqBasic.hs:71-73: primes > lower congruent to 1 mod m, ascending (generator)."""
    q = lower + ((m - lower) % m) + 1
    while True:
        if is_prime(q):
            yield q
        q += m


def first_good_q(m: int, lower: int) -> int:
    return next(good_qs(m, lower))


# ----------------------------------------------------------------------------
# roots of unity (ZqBasic.hs:144-177)
# ----------------------------------------------------------------------------

@lru_cache(maxsize=None)
def smallest_generator(q: int) -> int:
    assert is_prime(q)
    if q == 2:
        return 1
    order = q - 1
    exps = [order // p for p in prime_factors(order)]
    x = 1
    while True:
        # `values` enumerates 0,1,2,...; 0 and (for q>2) 1 fail isGen
        if pow(x, order, q) == 1 and all(pow(x, e, q) != 1 for e in exps):
            return x
        x += 1


def omega(m: int, q: int) -> int:
    """principal m-th root of unity: g0^((q-1)/m); None-case raises."""
    if not is_prime(q) or (q - 1) % m != 0:
        raise ValueError(f"no CRT basis: q={q} m={m}")
    return pow(smallest_generator(q), (q - 1) // m, q)


def mhat_inv(m: int, q: int) -> int:
    return pow(value_hat(m) % q, -1, q)


def ru_tables(pps, qs, inverse: bool = False):
    """CPP.hs:422-442.  Returns one list per prime power, interleaved over the RNS
    tuple exactly as the C side indexes it: tab[k][i*T+t] = omega_{m,t}^(+-i*m/pp_k)."""
    m = value_pps(pps)
    T = len(qs)
    ws = [omega(m, q) for q in qs]
    out = []
    for p, e in pps:
        pp = p ** e
        step = m // pp
        tab = [0] * (pp * T)
        for t, q in enumerate(qs):
            base = pow(ws[t], step, q)
            if inverse:
                base = pow(base, -1, q)
            x = 1
            for i in range(pp):
                tab[i * T + t] = x
                x = x * base % q
        out.append(tab)
    return out


# ----------------------------------------------------------------------------
# g vectors in the CRT basis (Tensor.hs:290-337)
# ----------------------------------------------------------------------------

def _kron_vec(pps, per_prime):
    """fKron/ppKron/indexK with width-1 matrices: entry i = prod_k M_k[i_k mod (p_k-1)],
    i = i_1 + phi_1*(i_2 + ...), smallest prime innermost."""
    n = totient_pps(pps)
    out = [0] * n
    phis = [totient_pp(pe) for pe in pps]
    for i in range(n):
        ii = i
        acc = None
        for (p, _), phi, vec in zip(pps, phis, per_prime):
            ik = ii % phi
            ii //= phi
            acc = vec[ik % (p - 1)] if acc is None else acc * vec[ik % (p - 1)]
        out[i] = 1 if acc is None else acc
    return out


def g_crt(pps, q: int, inverse: bool = False) -> list[int]:
    """gCRTK / gInvCRTK flattened (CPP.hs:444-454) for one modulus."""
    per_prime = []
    for p, _ in pps:
        if p == 2:
            per_prime.append([1])
            continue
        wp = omega(p, q)  # crtInfo at index p: consistent with omega_m^(m/p)
        if not inverse:
            per_prime.append([(1 - pow(wp, i + 1, q)) % q for i in range(p - 1)])
        else:
            phatinv = pow(p % q, -1, q)
            per_prime.append([
                phatinv * sum(j * pow(wp, (i + 1) * (p - 1 - j), q) for j in range(1, p)) % q
                for i in range(p - 1)])
    v = _kron_vec(pps, per_prime)
    return [x % q for x in v]


# ----------------------------------------------------------------------------
# re-indexing (Tensor.hs:342-379)
# ----------------------------------------------------------------------------

def digit_rev(p: int, e: int, j: int) -> int:
    acc = 0
    for k in range(e - 1, -1, -1):
        j, r = divmod(j, p)
        acc += r * p ** k
    return acc


def index_to_pow(pe, j: int) -> int:
    p, e = pe
    jq, jr = divmod(j, p - 1)
    return p ** (e - 1) * jr + digit_rev(p, e - 1, jq)


def index_to_zms(pe, i: int) -> int:
    p, _ = pe
    i1, i0 = divmod(i, p - 1)
    return p * i1 + i0 + 1


# ----------------------------------------------------------------------------
# ring-extension index tables (Tensor.hs:390-509)
# ----------------------------------------------------------------------------

def merge_pps(pps, pps2):
    """mergePPs: [(p, e, e')] for m | m'."""
    out = []
    a = list(pps)
    for p2, e2 in pps2:
        if a and a[0][0] == p2:
            assert a[0][1] <= e2, "m does not divide m'"
            out.append((p2, a[0][1], e2))
            a.pop(0)
        else:
            assert not a or a[0][0] > p2, "m does not divide m'"
            out.append((p2, 0, e2))
    assert not a, "m does not divide m'"
    return out


def totients(mpps):
    return [(totient_pp((p, e)), totient_pp((p, e2))) for p, e, e2 in mpps]


def to_index_pair(tots, i2: int):
    if not tots:
        assert i2 == 0
        return (0, 0)
    (phi, phi2), rest = tots[0], tots[1:]
    iq, ir = divmod(i2, phi2)
    irq, irr = divmod(ir, phi)
    q1, q0 = to_index_pair(rest, iq)
    return (irq + q1 * (phi2 // phi), irr + q0 * phi)


def from_index_pair(tots, pair) -> int:
    i1, i0 = pair
    if not tots:
        assert (i1, i0) == (0, 0)
        return 0
    (phi, phi2), rest = tots[0], tots[1:]
    i0q, i0r = divmod(i0, phi)
    i1q, i1r = divmod(i1, phi2 // phi)
    i = from_index_pair(rest, (i1q, i0q))
    return (i0r + i1r * phi) + i * phi2


def base_index_dec(mpps, i2: int):
    """baseIndexDec: None or (index, negate?)."""
    if not mpps:
        assert i2 == 0
        return (0, False)
    (p, e, e2), rest = mpps[0], mpps[1:]
    iq, ir = divmod(i2, totient_pp((p, e2)))
    phi = totient_pp((p, e))
    if p > 2 and e == 0 and e2 > 0:
        curr = {0: (0, False), 1: (0, True)}.get(ir)
    else:
        curr = (ir, False) if ir < phi else None
    if curr is None:
        return None
    nxt = base_index_dec(rest, iq)
    if nxt is None:
        return None
    return (curr[0] + phi * nxt[0], curr[1] != nxt[1])


def ext_indices_powdec(pps, pps2):
    tots = totients(merge_pps(pps, pps2))
    return [from_index_pair(tots, (0, i)) for i in range(totient_pps(pps))]


def ext_indices_crt(pps, pps2):
    tots = totients(merge_pps(pps, pps2))
    phi, phi2 = totient_pps(pps), totient_pps(pps2)
    rel = phi2 // phi
    out = []
    for k in range(phi2):
        a, b = divmod(k, rel)  # swap . divMod: pair = (k mod rel, k div rel)
        out.append(from_index_pair(tots, (b, a)))
    return out


def base_indices_pow(pps, pps2):
    tots = totients(merge_pps(pps, pps2))
    return [to_index_pair(tots, i) for i in range(totient_pps(pps2))]


def base_indices_dec(pps, pps2):
    mpps = merge_pps(pps, pps2)
    return [base_index_dec(mpps, i) for i in range(totient_pps(pps2))]


def base_indices_crt(pps, pps2):
    return [p[1] for p in base_indices_pow(pps, pps2)]


def ext_indices_coeffs(pps, pps2):
    tots = totients(merge_pps(pps, pps2))
    phi, phi2 = totient_pps(pps), totient_pps(pps2)
    return [[from_index_pair(tots, (i1, i0)) for i0 in range(phi)] for i1 in range(phi2 // phi)]


# ----------------------------------------------------------------------------
# twace / embed on plain Python lists of one RNS component (Extension.hs:54-129)
# ----------------------------------------------------------------------------

def embed_pow(pps, pps2, arr):
    return [arr[j1] if j0 == 0 else 0 for (j0, j1) in base_indices_pow(pps, pps2)]


def embed_dec(pps, pps2, arr, q: int):
    out = []
    for ent in base_indices_dec(pps, pps2):
        if ent is None:
            out.append(0)
        else:
            sh, neg = ent
            out.append((-arr[sh]) % q if neg else arr[sh])
    return out


def embed_crt(pps, pps2, arr):
    return [arr[i] for i in base_indices_crt(pps, pps2)]


def twace_powdec(pps, pps2, arr):
    return [arr[i] for i in ext_indices_powdec(pps, pps2)]


def twace_crt_tweak(pps, pps2, q: int):
    """tweak = mhat * g' / (m'hat * g) in the CRT basis of O_m' (Extension.hs:110-125)."""
    m, m2 = value_pps(pps), value_pps(pps2)
    gp = g_crt(pps2, q)
    ginv = g_crt(pps, q, inverse=True)
    hat_ratio_inv = mhat_inv(m2, q) * (value_hat(m) % q) % q
    emb = embed_crt(pps, pps2, ginv)
    return [e * g % q * hat_ratio_inv % q for e, g in zip(emb, gp)]


def twace_crt(pps, pps2, arr, q: int):
    tweak = twace_crt_tweak(pps, pps2, q)
    idx = ext_indices_crt(pps, pps2)
    phi, phi2 = totient_pps(pps), totient_pps(pps2)
    rel = phi2 // phi
    prod = [t * a % q for t, a in zip(tweak, arr)]
    v = [prod[i] for i in idx]
    return [sum(v[i * rel:(i + 1) * rel]) % q for i in range(phi)]
