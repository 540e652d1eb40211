// lol_amd/csrc/mixed_cls0f16.hip — class 0, fused poly-mul, 16 coefficients per thread (its own translation unit:
// one instantiation of the 64-bit fused kernel takes over two minutes to compile)
#include "mixed_impl.h"
namespace lolhip {
template hipError_t launch_cls_k<0, 2, 16, false>(const MixedLaunch&);
}  // namespace lolhip
