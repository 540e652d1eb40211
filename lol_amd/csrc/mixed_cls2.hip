// lol_amd/csrc/mixed_cls2.hip — the mixed-radix kernels of arithmetic/storage class 2 (see mixed_impl.h)
#include "mixed_impl.h"
namespace lolhip {
template hipError_t launch_cls<2, 0>(const MixedLaunch&);
template hipError_t launch_cls<2, 2>(const MixedLaunch&);
}  // namespace lolhip
