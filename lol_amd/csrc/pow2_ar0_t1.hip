// lol_amd/csrc/pow2_ar0_t1.hip — arithmetic class AR = 0 of the m = 2^k kernels, single-modulus launches with
// 16-byte global accesses (k_pow2<..., T1 = true>, pow2_impl.h)
#include "pow2_impl.h"
namespace lolhip {
template hipError_t launch_pow2_ar<0, true>(const Pow2Launch&, int);
}  // namespace lolhip
