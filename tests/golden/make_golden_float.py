#!/usr/bin/env python3
"""Generates tests/golden/golden_float.npz: inputs and outputs of lol-cpp's floating-point Tensor
symbols (tensorCRTC, tensorCRTInvC, tensorGaussianDec; crt.cpp:583-598, random.cpp:61-64) run
HERE from the reference's own sources (oracle/_ref/libctensor.so, built by oracle/Makefile).
The fixture travels to the GPU box; /root/reference does not.

    python tests/golden/make_golden_float.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import lolmath as lm  # noqa: E402
from oracle.oracle import CTRef  # noqa: E402

INDICES = [8, 9, 15, 21, 45, 64, 63, 128 * 7, 1024, 1728, 15015]


def main():
    ct = CTRef()
    rng = np.random.default_rng(2024)
    out = {"indices": np.array(INDICES, dtype=np.int64)}
    for m in INDICES:
        pps = lm.factor_pps(m)
        n = lm.totient_pps(pps)
        B = 1 if n > 2000 else 3
        z = rng.standard_normal((B, n)) + 1j * rng.standard_normal((B, n))
        g = rng.standard_normal((B, n)) * 3.0
        out[f"m{m}_cin"] = z
        out[f"m{m}_crtc"] = ct.crtc(pps, z)
        out[f"m{m}_crtinvc"] = ct.crtinvc(pps, z)
        out[f"m{m}_gin"] = g
        out[f"m{m}_gauss"] = ct.gaussian_dec(pps, g)
    path = os.path.join(ROOT, "tests", "golden", "golden_float.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
