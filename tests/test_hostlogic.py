"""CPU tests of the PRODUCT's host side (no GPU, no compute calls):
 - liblolhip.so loads and exports every symbol include/lolhip.h declares;
 - the plan's host tables (ru, ruInv, mhatInv, gCRT, gInvCRT, index tables) equal the
   oracle's independent Python restatement of Lol's rules;
 - error behaviour: no CPU fallback, status codes instead of exit().
"""
import ctypes as C
import os
import re

import numpy as np
import pytest

from oracle import lolmath as lm
from params import BENCH2, TENSOR1, TENSOR2

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(lolhip):
    hdr = open(os.path.join(ROOT, "include", "lolhip.h")).read()
    names = set(re.findall(r"LOLHIP_API\s+[\w\s\*]+?\b(\w+)\s*\(", hdr))
    assert len(names) >= 40
    raw = C.CDLL(lolhip.lib_path())
    for nm in sorted(names):
        assert hasattr(raw, nm), f"liblolhip.so does not export {nm}"
    # the ten reference symbols (lol-cpp/.../CPP/Backend.hs:304-337, Z_q rows)
    for nm in ("tensorCRTRq", "tensorCRTInvRq", "mulRq", "tensorLRq", "tensorLInvRq", "tensorGPowRq",
               "tensorGDecRq", "tensorGInvPowRq", "tensorGInvDecRq"):
        assert nm in names


def test_no_oracle_or_cpu_fallback_in_product():
    """The product must not route through oracle/ or any CPU path."""
    for dp, _dn, fn in os.walk(os.path.join(ROOT, "lol_amd")):
        for f in fn:
            if f.endswith((".py", ".cpp", ".h", ".hip")):
                src = open(os.path.join(dp, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f
                assert "cpu_ref" not in src and "libcpuref" not in src and "libctensor" not in src, f


@pytest.mark.parametrize("m,qs", TENSOR1 + [(1024, [12289]), (2 ** 14, [1073872897]), (15015, [1073842771]),
                                             (64 * 27, [3457]), (64 * 81, [10369]), (64 * 9 * 25, [14401])])
def test_plan_tables_follow_lol_rules(lolhip, m, qs):
    pps = lm.factor_pps(m)
    assert lolhip.factor_pps(m) == pps
    P = lolhip.Plan(pps, qs, host_only=True)
    assert (P.n, P.m, P.T, P.has_crt) == (lm.totient_pps(pps), m, len(qs), True)
    ru, rui = lm.ru_tables(pps, qs), lm.ru_tables(pps, qs, inverse=True)
    for k in range(len(pps)):
        assert list(P.ru(k)) == ru[k]
        assert list(P.ruInv(k)) == rui[k]
    assert list(P.mhatInv()) == [lm.mhat_inv(m, q) for q in qs]
    if P.n <= 2000:
        for t, q in enumerate(qs):
            assert list(P.gCRT()[:, t]) == lm.g_crt(pps, q)
            assert list(P.gInvCRT()[:, t]) == lm.g_crt(pps, q, inverse=True)


def test_good_q_and_61bit_modulus(lolhip):
    for m, lower in ((2 ** 14, 2 ** 60), (2 ** 15, 2 ** 59), (15015, 2 ** 30), (15015, 2 ** 60), (7, 20)):
        assert lolhip.good_q(m, lower) == lm.first_good_q(m, lower)
    q = lolhip.good_q(2 ** 14, 2 ** 60)
    P = lolhip.Plan([(2, 14)], [q], host_only=True)
    assert list(P.ru(0)) == lm.ru_tables([(2, 14)], [q])[0]


@pytest.mark.parametrize("m,m2", [(a, b) for a, b, _ in TENSOR2] + [(a, b) for a, b, _ in BENCH2] + [(12, 60), (9, 45), (1, 1)])
def test_extension_tables(lolhip, m, m2):
    a, b = lm.factor_pps(m), lm.factor_pps(m2)
    q = lm.first_good_q(m2, 1000)
    X = lolhip.Ext(lolhip.Plan(a, [q], host_only=True), lolhip.Plan(b, [q], host_only=True))
    assert list(X.table(0)) == lm.ext_indices_powdec(a, b)
    assert list(X.table(1)) == lm.ext_indices_crt(a, b)
    assert list(X.table(2)) == [j1 if j0 == 0 else -1 for (j0, j1) in lm.base_indices_pow(a, b)]
    assert list(X.table(3)) == [(-1 if e is None else (e[0] | ((1 << 30) if e[1] else 0))) for e in lm.base_indices_dec(a, b)]
    assert list(X.table(4)) == lm.base_indices_crt(a, b)
    assert list(X.table(5)) == [e for row in lm.ext_indices_coeffs(a, b) for e in row]


def test_plan_without_crt_basis(lolhip):
    # Zq 32 with F7 (Default.hs:82): valid for L/G, no CRT
    P = lolhip.Plan([(7, 1)], [32], host_only=True)
    assert not P.has_crt and P.n == 6
    assert P.ru(0).size == 0


def test_error_codes_not_exit(lolhip):
    with pytest.raises(lolhip.LolHipError):
        lolhip.Plan([(4, 1)], [17], host_only=True)         # 4 is not prime
    with pytest.raises(lolhip.LolHipError):
        lolhip.Plan([(3, 1), (2, 2)], [13], host_only=True)  # not ascending
    with pytest.raises(lolhip.LolHipError):
        lolhip.Plan([(2, 3)], [1], host_only=True)           # modulus < 2
    with pytest.raises(lolhip.LolHipError):
        lolhip.Plan([(2, 3)], [2 ** 62 + 1], host_only=True)  # modulus too large
    with pytest.raises(lolhip.LolHipError):
        # wrong root of unity handed in by the caller
        lolhip.Plan([(2, 3)], [17], host_only=True, omega_pp=[16])
    a = lolhip.Plan([(2, 2)], [17], host_only=True)
    b = lolhip.Plan([(3, 1)], [17 if False else 13], host_only=True)
    with pytest.raises(lolhip.LolHipError):
        lolhip.Ext(a, b)                                     # 4 does not divide 3, moduli differ


def test_compute_without_gpu_fails_loudly(lolhip):
    """No CPU fallback anywhere: a host-only plan refuses to compute."""
    P = lolhip.Plan([(2, 3)], [17], host_only=True)
    y = np.arange(4, dtype=np.int64).reshape(1, 4, 1)
    for f in (P.crt, P.crtInv, P.l, P.mulGPow, P.divGDec, P.mulGCRT):
        with pytest.raises(lolhip.NoDeviceError):
            f(y)
    with pytest.raises(lolhip.NoDeviceError):
        P.mul(y, y)
    with pytest.raises(lolhip.NoDeviceError):
        P.polymul(y, y)
    if lolhip.device_count() == 0:
        with pytest.raises(lolhip.NoDeviceError):
            lolhip.Plan([(2, 3)], [17])
        # drop-in symbol: void signature, status retrievable
        raw = lolhip.lib()
        raw.mulRq.argtypes = [C.c_int16, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        raw.mulRq.restype = None
        a = np.array([1, 2, 3], dtype=np.int64)
        qs = np.array([17], dtype=np.int64)
        raw.mulRq(1, a.ctypes.data, a.ctypes.data, 3, qs.ctypes.data)
        assert raw.lolhip_last_status() == -5
        assert list(a) == [1, 2, 3]       # untouched


def test_device_code_has_no_unpadded_wide_store(lolhip):
    """gfx950 needs two wait states between a VMEM store of more than 8 bytes and a VALU write of its data
    registers; hipcc (ROCm 7.2) leaves the pair unpadded when the buffer store takes its offset from an SGPR, and
    a persistent loop then stored the loop counter in place of residues (profiles/r03_store_hazard.txt).  The
    shipped library's device code is disassembled and audited for the pair."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "check_store_hazard.py"), lolhip.lib_path()],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "0 unpadded store-data pair(s)" in r.stdout


def test_class2_plans_merge_small_prime_powers():
    """plan.cpp merge_stages / tile grouping, on host-only plans (no GPU): what a lone crt of the
    reference's benchmark index 64*9*25 (Benchmarks/Default.hs:49-50) launches.  Kinds: 12/13 = 2-power tile
    forward/inverse, 1 = DFT_p, 2 = CRT_p, 3 = CRT_p^-1 (plan.h)."""
    import lol_amd
    def prog(q, inverse=False, m=14400):
        return [tuple(int(v) for v in r) for r in lol_amd.Plan(lm.factor_pps(m), [q], host_only=True).program(inverse)]
    q26, q30, q45 = (lm.first_good_q(14400, 2 ** b) for b in (26, 30, 45))
    assert prog(q26) == [(12, 1, 4, 1), (12, 5, 1, 16), (2, 3, 6, 32), (2, 5, 20, 192)]
    assert prog(q26, True) == [(3, 3, 6, 32), (3, 5, 20, 192), (13, 5, 1, 16), (13, 1, 4, 1)]      # tensor factors commute: same order of primes
    # 20 (q-1)^2 no longer fits 64 bits: 5^2 stays staged, 3^2 (6 terms) still merges
    assert prog(q30) == [(12, 1, 4, 1), (12, 5, 1, 16), (2, 3, 6, 32), (2, 5, 4, 192), (1, 5, 5, 768)]
    # 64-bit residues: the staged form, four levels per tile
    assert prog(q45) == [(12, 1, 4, 1), (12, 5, 1, 16), (2, 3, 2, 32), (1, 3, 3, 64), (2, 5, 4, 192), (1, 5, 5, 768)]
    lol_amd.debug_set("NO_MERGE", True)
    try:
        assert prog(q26) == prog(q45)
    finally:
        lol_amd.debug_set("NO_MERGE", False)
    # adjacent small factors as one Kronecker stage: 3 (x) 5 = an 8-vector (BASELINE config 4's index); not for 64-bit residues
    q30, q60 = lm.first_good_q(15015, 2 ** 30), lm.first_good_q(15015, 2 ** 60)
    assert prog(q30, m=15015) == [(2, 3, 8, 1), (2, 7, 6, 8), (2, 11, 10, 48), (2, 13, 12, 480)]
    assert prog(q30, True, m=15015) == [(3, 3, 8, 1), (3, 7, 6, 8), (3, 11, 10, 48), (3, 13, 12, 480)]
    assert [r[2] for r in prog(q60, m=15015)] == [2, 4, 6, 10, 12]
    # 3^4 is too long for one vector: its two outermost radix-3 stages run as one DFT_9 stage (64 * 81, Benchmarks/Default.hs:42-46)
    q81 = lm.first_good_q(5184, 2 ** 26)
    assert prog(q81, m=5184) == [(12, 1, 4, 1), (12, 5, 1, 16), (2, 3, 2, 32), (1, 3, 3, 64), (2, 3, 9, 192)]
    assert prog(q81, True, m=5184) == [(2, 3, 9, 192), (1, 3, 3, 64), (3, 3, 2, 32), (13, 5, 1, 16), (13, 1, 4, 1)]
