// lol_amd/csrc/pow2_ar2.hip — the m = 2^k kernels of arithmetic class AR = 2 (see pow2_impl.h)
#include "pow2_impl.h"
namespace lolhip {
template hipError_t launch_pow2_ar<2, false>(const Pow2Launch&, int);
}  // namespace lolhip
#if defined(LOLHIP_STAMPS) && LOLHIP_STAMPS == 2      // diagnostic build only (make ... CXXFLAGS+=-DLOLHIP_STAMPS=2): phase stamps of THIS class
extern "C" __attribute__((visibility("default"))) int lolhip_debug_set_stamps(unsigned long long* dev) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(lolhip::g_stamp_buf), &dev, sizeof(dev));
}
#endif
