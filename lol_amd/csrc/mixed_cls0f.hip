// lol_amd/csrc/mixed_cls0f.hip — class 0, fused poly-mul (its own translation unit: the 64-bit classes take minutes each)
#include "mixed_impl.h"
namespace lolhip {
template hipError_t launch_cls<0, 2>(const MixedLaunch&);
}  // namespace lolhip
