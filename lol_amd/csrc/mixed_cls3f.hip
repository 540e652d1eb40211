// lol_amd/csrc/mixed_cls3f.hip — class 3, fused poly-mul
#include "mixed_impl.h"
namespace lolhip {
template hipError_t launch_cls<3, 2>(const MixedLaunch&);
}  // namespace lolhip
