"""Shared fixtures.  `-m "not gpu"` runs here on CPU; `-m gpu` runs on an MI355X box."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def cpuref():
    from oracle.oracle import CpuRef
    return CpuRef()          # builds oracle/_build/libcpuref.so if it is missing


@pytest.fixture(scope="session")
def ctref():
    """The reference's own lol-cpp C++ (oracle/_ref), when it has been built."""
    from oracle.oracle import CTRef, CTREF_SO, build
    if not os.path.exists(CTREF_SO) and os.path.isdir("/root/reference"):
        build(ref=True)
    if not os.path.exists(CTREF_SO):
        pytest.skip("oracle/_ref/libctensor.so not built (no /root/reference here)")
    return CTRef()


@pytest.fixture(scope="session")
def golden():
    path = os.path.join(ROOT, "tests", "golden", "golden_ct.npz")
    z = np.load(path, allow_pickle=False)
    return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def lolhip():
    """The product library; built in-tree by __graft_entry__.build()."""
    import lol_amd
    if not os.path.exists(lol_amd.lib_path()):
        import __graft_entry__
        __graft_entry__.build()
    lol_amd.lib()
    return lol_amd


@pytest.fixture(scope="session")
def gpu(lolhip):
    if lolhip.device_count() == 0:
        pytest.fail("-m gpu tests need a GPU: liblolhip has no CPU fallback")
    return lolhip


SWITCHES = ("GENERIC_SCALAR", "NO_FUSED2", "NO_POW2_PART", "POLYMUL_UNFUSED", "KEYSWITCH_UNFUSED", "NO_T1", "NO_PIPE", "FORCE_PIPE", "NO_OWN_DIAG", "NO_MERGE", "NO_LAZY", "NO_KRON")


@pytest.fixture(autouse=True)
def _release_forced_paths():
    """A test that forces a launch path (lol_amd.debug_set) and fails must not leave it forced."""
    yield
    import lol_amd.tensor as t
    if t._lib is not None:
        for nm in SWITCHES:
            t._lib.lolhip_debug_set(nm.encode(), 0)
