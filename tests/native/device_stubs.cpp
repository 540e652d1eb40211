// tests/native/device_stubs.cpp — the launcher symbols of the .hip translation units, stubbed for the
// CPU-only sanitizer build of the HOST code (plan.cpp, hostmath.cpp, wire.cpp, capi.cpp).  Nothing
// here computes: every launcher reports "no device", which is also what the real library does
// without a GPU.  Test infrastructure only.
#include "kernels.h"
#include "pipeline.h"

namespace lolhip {
hipError_t launch_pow2(const Pow2Launch&, int) { return hipErrorNoDevice; }
hipError_t launch_keyswitch_fused(const KeySwitchLaunch&) { return hipErrorNoDevice; }
hipError_t launch_generic(const GenericLaunch&) { return hipErrorNoDevice; }
bool mixed_ok(i64, const Stage*, int, const u64*, int) { return false; }
hipError_t launch_mixed(const MixedLaunch&) { return hipErrorNoDevice; }
hipError_t launch_mixed_keyswitch(const MixedKeySwitchLaunch&) { return hipErrorNoDevice; }
bool mixed_keyswitch_big_ok(i64) { return false; }
hipError_t launch_cplx(hipStream_t, double*, i64, i64, const Stage*, int, const double*) { return hipErrorNoDevice; }
hipError_t launch_gauss(hipStream_t, double*, i64, i64, const Stage*, int, const double*) { return hipErrorNoDevice; }
hipError_t launch_pointwise_mul(hipStream_t, i64*, const i64*, i64, i64, int, const ModCtx*) { return hipErrorNoDevice; }
hipError_t launch_copy16(hipStream_t, void*, const void*, size_t, int) { return hipErrorNoDevice; }
hipError_t launch_gather(hipStream_t, i64*, const i64*, const int32_t*, i64, i64, i64, int, const ModCtx*, bool) { return hipErrorNoDevice; }
hipError_t launch_twace_crt(hipStream_t, i64*, const i64*, const int32_t*, const i64*, i64, i64, i64, int, const ModCtx*) { return hipErrorNoDevice; }
hipError_t launch_ctmul(hipStream_t, const i64*, const i64*, const i64*, const i64*, i64*, i64*, i64*, const i64*, i64, i64, int, const ModCtx*) { return hipErrorNoDevice; }
hipError_t launch_decompose(hipStream_t, const i64*, i64*, i64, i64, const DecompParams&, const ModCtx*, bool) { return hipErrorNoDevice; }
hipError_t launch_knapsack(hipStream_t, const i64*, int, const i64*, int, const i64*, i64*, i64, i64, int, const ModCtx*, bool) { return hipErrorNoDevice; }
hipError_t launch_rescale(hipStream_t, const i64*, i64*, i64, i64, const RescaleParams&, const ModCtx*) { return hipErrorNoDevice; }
hipError_t launch_coeffs(hipStream_t, i64*, const i64*, const int32_t*, i64, i64, i64, int, const ModCtx*) { return hipErrorNoDevice; }
}  // namespace lolhip
