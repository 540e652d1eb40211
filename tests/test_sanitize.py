"""Sanitizers on the CPU build: the product's HOST C++ (plan.cpp, hostmath.cpp,
wire.cpp, capi.cpp) compiled with -fsanitize=address,undefined and driven through the C ABI by
tests/native/host_sanitize.cpp — plan construction over a grid of indices and moduli, table
queries with short buffers, ring-extension tables, malformed requests, and the protobuf codec on
every truncation and on thousands of random mutations of valid messages.  The device launchers are
stubs (tests/native/device_stubs.cpp): GPU AddressSanitizer is not available on this pool."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "lol_amd", "csrc")


def test_host_code_under_asan_ubsan(tmp_path):
    cxx = shutil.which("g++")
    if cxx is None or not os.path.isdir("/opt/rocm/include"):
        pytest.skip("needs g++ and the ROCm headers")
    exe = tmp_path / "host_sanitize"
    srcs = [os.path.join(CSRC, f) for f in ("plan.cpp", "hostmath.cpp", "wire.cpp", "capi.cpp")]
    srcs += [os.path.join(ROOT, "tests", "native", f) for f in ("device_stubs.cpp", "host_sanitize.cpp")]
    cmd = [cxx, "-std=c++20", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer",
           "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", f"-I{os.path.join(ROOT, 'include')}", f"-I{CSRC}",
           *srcs, "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([str(exe)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "host sanitize ok" in r.stdout
