// lol_amd/csrc/mixed_cls3.hip — the mixed-radix kernels of arithmetic/storage class 3, single program (see mixed_impl.h)
#include "mixed_impl.h"
namespace lolhip {
template hipError_t launch_cls<3, 0>(const MixedLaunch&);
}  // namespace lolhip
