// lol_amd/csrc/pow2_ar3.hip — the m = 2^k kernels of arithmetic class AR = 3 (see pow2_impl.h)
#include "pow2_impl.h"
namespace lolhip {
template hipError_t launch_pow2_ar<3, false>(const Pow2Launch&, int);
}  // namespace lolhip
