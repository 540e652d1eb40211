"""Mode A of INTEGRATION.md, proved once at link level (a `not gpu` test; needs /root/reference and binutils):
lol-cpp's own sources, partially linked with the nine Z_q symbols localised (haskell/lol-hip/modeA/Makefile), plus
liblolhip.so satisfy all 29 imports of lol-cpp/Crypto/Lol/Cyclotomic/Tensor/CPP/Backend.hs:304-337 — the Z_q ones
from liblolhip, the other twenty from lol-cpp — with no duplicate definition."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/lol-cpp"


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is not on this machine")
def test_mode_a_links_and_splits_the_29_symbols(tmp_path, lolhip):
    for tool in ("ld", "objcopy", "ar", "g++", "gcc"):
        if shutil.which(tool) is None:
            pytest.skip(f"{tool} not installed")
    out = tmp_path / "modea"
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "haskell", "lol-hip", "modeA"), f"LOLCPP={REF}", f"OUT={out}"], check=True)
    # the partially linked object still CONTAINS the nine Z_q symbols, but as local ones
    nm = subprocess.run(["nm", str(out / "lolcpp_rest.o")], capture_output=True, text=True, check=True).stdout
    kinds = {ln.split()[-1]: ln.split()[-2] for ln in nm.splitlines() if len(ln.split()) >= 2}
    zq = "tensorLRq tensorLInvRq mulRq tensorGPowRq tensorGDecRq tensorGInvPowRq tensorGInvDecRq tensorCRTRq tensorCRTInvRq".split()
    assert all(kinds[s] == "t" for s in zq), {s: kinds.get(s) for s in zq}
    assert all(kinds[s] == "T" for s in ("tensorLR", "tensorGPowC", "tensorCRTC", "tensorGaussianDec", "mulC", "tensorNormSqD"))
    exe = tmp_path / "modea_driver"
    libdir = os.path.dirname(lolhip.lib_path())
    subprocess.run(["gcc", "-O1", "-o", str(exe), os.path.join(ROOT, "tests", "native", "modea_driver.c"),
                    f"-L{out}", "-llolcpp_rest", f"-L{libdir}", "-llolhip", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib",
                    "-lstdc++", "-lm", "-ldl"], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "29 symbols, 9 of them Z_q -> liblolhip.so, 0 misplaced" in r.stdout
    assert "tensorLR (lol-cpp): [5,7] -> [5,12] ok" in r.stdout
    assert "tensorCRTRq (liblolhip)" in r.stdout and "WRONG" not in r.stdout
    # and WITHOUT the localisation the link still succeeds, silently the wrong way round: the executable's own (lol-cpp)
    # definitions of the nine Z_q symbols shadow liblolhip's, and nothing would ever reach the GPU
    exe2 = tmp_path / "modea_shadowed"
    subprocess.run(["gcc", "-O1", "-o", str(exe2), os.path.join(ROOT, "tests", "native", "modea_driver.c"), str(out / "lolcpp_all.o"),
                    f"-L{libdir}", "-llolhip", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-lstdc++", "-lm", "-ldl"], check=True)
    r2 = subprocess.run([str(exe2)], capture_output=True, text=True, timeout=120)
    assert r2.returncode != 0 and "9 misplaced" in r2.stdout
