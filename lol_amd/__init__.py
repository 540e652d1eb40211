"""lol_amd — MI355X-native backend for the Z_q hot path of Lol's `Tensor` class.

The product is `liblolhip.so` (hand-written HIP for gfx950 behind the C ABI in
include/lolhip.h).  This package is the thin host-side mirror of the reference's
operator interface (`Crypto.Lol.Cyclotomic.Tensor`, lol/Crypto/Lol/Cyclotomic/Tensor.hs:86-193,
as instantiated by lol-cpp/Crypto/Lol/Cyclotomic/Tensor/CPP.hs:204-264): same
operation names and argument meaning, a leading batch dimension, and nothing else.
There is no CPU fallback: without the built library, or without a GPU, compute
calls raise.
"""
from .tensor import (  # noqa: F401
    LolHipError, NoDeviceError, Plan, Ext, lib, lib_path, device_count, debug_set, good_q, factor_pps,
    rqproduct_read, rqproduct_write, kshint_read, kshint_write, r_read, secretkey_read, kqproduct_read,
    linearrq_read, tunnelhint_read, r_write, secretkey_write, linearrq_write, tunnelhint_write, chain_read, chain_write,
)

__all__ = ["LolHipError", "NoDeviceError", "Plan", "Ext", "lib", "lib_path", "device_count", "debug_set",
           "good_q", "factor_pps", "rqproduct_read", "rqproduct_write", "kshint_read", "kshint_write", "r_read",
           "secretkey_read", "kqproduct_read", "linearrq_read", "tunnelhint_read", "r_write", "secretkey_write",
           "linearrq_write", "tunnelhint_write", "chain_read", "chain_write"]
