// lol_amd/csrc/mixed.hip — dispatcher of the mixed-radix path (kernels: mixed_impl.h, one
// translation unit per arithmetic class).
#include "kernels.h"

namespace lolhip {

template <int CLS, int MODE> hipError_t launch_cls(const MixedLaunch& a);     // MODE 0: one program, 2: fused poly-mul
template <int CLS, int MODE, int KMAX, bool BIG = false> hipError_t launch_cls_k(const MixedLaunch& a);
#define LOLHIP_EXT(C) extern template hipError_t launch_cls<C, 0>(const MixedLaunch&);
LOLHIP_EXT(0) LOLHIP_EXT(1) LOLHIP_EXT(2) LOLHIP_EXT(3)
#undef LOLHIP_EXT
extern template hipError_t launch_cls<1, 2>(const MixedLaunch&);
extern template hipError_t launch_cls<2, 2>(const MixedLaunch&);
extern template hipError_t launch_cls<4, 0>(const MixedLaunch&);
extern template hipError_t launch_cls<4, 2>(const MixedLaunch&);
// the fused poly-mul of the 64-bit classes: one translation unit per coefficients-per-thread variant
#define LOLHIP_EXT(C) extern template hipError_t launch_cls_k<C, 2, 12, false>(const MixedLaunch&); extern template hipError_t launch_cls_k<C, 2, 16, false>(const MixedLaunch&);
LOLHIP_EXT(0) LOLHIP_EXT(3)
#undef LOLHIP_EXT
// same geometry rule as mixed_impl.h mixed_geom: up to ~2048 packed coefficients per workgroup, 128/256/512 threads
static bool fused_k12(const MixedLaunch& a) {
  int ppw = 1;
  while ((size_t)(ppw * 2) * a.n <= 2048 && ppw * 2 <= a.B) ppw *= 2;
  const size_t coeffs = (size_t)ppw * a.n;
  const int threads = coeffs > 4096 ? 512 : (coeffs > 2048 ? 256 : 128);
  return (coeffs + threads - 1) / threads <= 12;
}

bool mixed_ok(i64 n, const Stage* host_stages, int nstages, const u64* qs, int T) {
  if (n > 8192) return false;
  for (int t = 0; t < T; ++t) if (qs[t] <= 16) return false;     // small multipliers (<= 13) must be residues as they stand
  for (int s = 0; s < nstages; ++s) {
    const Stage& st = host_stages[s];
    if (st.kind == ST_DIAG || st.kind == ST_SCALE) continue;
    if (st.kind == ST_POW2F || st.kind == ST_POW2I) { if (st.d < 1 || st.d > 4) return false; continue; }
    const int d = st.d;
    if (!(d == 2 || d == 3 || d == 4 || d == 5 || d == 6 || d == 7 || d == 8 || d == 9 || d == 10 || d == 11 || d == 12 || d == 13 || d == 18 || d == 20)) return false;      // 18, 20: merged prime powers, class 2 / 4 plans only (plan.cpp)
  }
  return true;
}

hipError_t launch_mixed(const MixedLaunch& a) {
  if (a.B == 0) return hipSuccess;
  switch (a.cls) {
    case 0: return a.fused ? (fused_k12(a) ? launch_cls_k<0, 2, 12>(a) : launch_cls_k<0, 2, 16>(a)) : launch_cls<0, 0>(a);
    case 1: return a.fused ? launch_cls<1, 2>(a) : launch_cls<1, 0>(a);
    case 3: return a.fused ? (fused_k12(a) ? launch_cls_k<3, 2, 12>(a) : launch_cls_k<3, 2, 16>(a)) : launch_cls<3, 0>(a);
    case 4: return a.fused ? launch_cls<4, 2>(a) : launch_cls<4, 0>(a);
    default: return a.fused ? launch_cls<2, 2>(a) : launch_cls<2, 0>(a);
  }

}

}  // namespace lolhip
