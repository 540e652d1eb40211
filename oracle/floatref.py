"""oracle/floatref.py — numpy restatement of the floating-point members of the Tensor class.
TEST INFRASTRUCTURE (only tests/ may import it).

  crt_c / crtinv_c   the closed form of SURVEY.md Appendix A evaluated over C with
                     omega_m = exp(2 pi i / m): what lol-cpp's tensorCRTC / tensorCRTInvC
                     (crt.cpp:583-598, ppcrt/ppcrtinv at Complex) compute, as a dense n x n matrix
                     (small n only: O(n^2))
  gaussian_dec       tensorGaussianDec (random.cpp:19-64): per odd prime p the real
                     (p-1) x (p-1) matrix of primeD applied along axis k

Pinned against lol-cpp itself (oracle/_ref, run here) and the committed fixtures in
tests/golden/golden_float.npz (tests/test_float.py).
"""
from __future__ import annotations

import numpy as np

from . import lolmath as lm


def _digits(pps):
    """phi_k and the mixed-radix digit tables of the tensor index (k = 1 fastest, tensor.h:46-73)."""
    phis = [(p - 1) * p ** (e - 1) for p, e in pps]
    n = int(np.prod(phis)) if phis else 1
    idx = np.arange(n)
    digs = []
    for ph in phis:
        digs.append(idx % ph)
        idx = idx // ph
    return phis, n, digs


def crt_matrix_c(pps, inverse=False):
    """M[i, j] = prod_k omega_{pp_k}^{pow_k(j_k) * zms_k(i_k)} (Tensor.hs:359-368; Appendix A)."""
    phis, n, digs = _digits(pps)
    M = np.ones((n, n), dtype=np.complex128)
    for (p, e), dig in zip(pps, digs):
        pp = p ** e
        zms = p * (dig // (p - 1)) + dig % (p - 1) + 1                       # units of Z_{p^e}, row index
        hi = dig // (p - 1)
        rev = np.array([lm.digit_rev(p, e - 1, int(h)) for h in hi])         # digit-reversed powerful basis, column index
        pw = p ** (e - 1) * (dig % (p - 1)) + rev
        ex = (zms[:, None] * pw[None, :]) % pp
        M *= np.exp(2j * np.pi * ex / pp)
    return np.linalg.inv(M) if inverse else M


def crt_c(pps, y):
    n = lm.totient_pps(pps)
    return np.asarray(y, dtype=np.complex128).reshape(-1, n) @ crt_matrix_c(pps).T


def crtinv_c(pps, y):
    n = lm.totient_pps(pps)
    return np.asarray(y, dtype=np.complex128).reshape(-1, n) @ crt_matrix_c(pps, inverse=True).T


def gaussian_dec(pps, y):
    """random.cpp:19-64: out[row] = (sum_col 2 c(row*col mod p) y[col-1]) / sqrt 2 on every
    (p-1)-vector of axis k; c = Re omega_p^k for col <= p/2, Im for col > p/2; identity for p = 2."""
    phis, n, _ = _digits(pps)
    y = np.asarray(y, dtype=np.float64).reshape(-1, n).copy()
    B = y.shape[0]
    rts = 1
    for (p, e), ph in zip(pps, phis):
        if p != 2:
            D = np.zeros((p - 1, p - 1))
            for row in range(p - 1):
                for col in range(1, p):
                    ang = 2.0 * np.pi * ((row * col) % p) / p
                    D[row, col - 1] = 2.0 * (np.cos(ang) if col <= p // 2 else np.sin(ang)) / np.sqrt(2.0)
            lts = n // (rts * (p - 1))
            v = y.reshape(B, lts, p - 1, rts)
            y = np.einsum("rc,blcs->blrs", D, v).reshape(B, n)
        rts *= ph
    return y
