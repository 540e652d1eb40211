// lol_amd/csrc/mixed_cls4.hip — the mixed-radix kernels of class 4: class 2 with lazy dense stages, every modulus below 2^27
// (see mixed_impl.h)
#include "mixed_impl.h"
namespace lolhip {
template hipError_t launch_cls<4, 0>(const MixedLaunch&);
template hipError_t launch_cls<4, 2>(const MixedLaunch&);
}  // namespace lolhip
