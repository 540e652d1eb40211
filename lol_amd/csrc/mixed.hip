// lol_amd/csrc/mixed.hip — dispatcher of the mixed-radix path (kernels: mixed_impl.h, one
// translation unit per arithmetic class).
#include "kernels.h"

namespace lolhip {

template <int CLS> hipError_t launch_mixed_cls(const MixedLaunch& a);
extern template hipError_t launch_mixed_cls<0>(const MixedLaunch&);
extern template hipError_t launch_mixed_cls<1>(const MixedLaunch&);
extern template hipError_t launch_mixed_cls<2>(const MixedLaunch&);
extern template hipError_t launch_mixed_cls<3>(const MixedLaunch&);

bool mixed_ok(i64 n, const Stage* host_stages, int nstages, const u64* qs, int T) {
  if (n > 8192) return false;
  for (int t = 0; t < T; ++t) if (qs[t] <= 16) return false;     // small multipliers (<= 13) must be residues as they stand
  for (int s = 0; s < nstages; ++s) {
    const Stage& st = host_stages[s];
    if (st.kind == ST_DIAG || st.kind == ST_SCALE) continue;
    if (st.kind == ST_POW2F || st.kind == ST_POW2I) { if (st.d < 1 || st.d > 4) return false; continue; }
    const int d = st.d;
    if (!(d == 2 || d == 3 || d == 4 || d == 5 || d == 6 || d == 7 || d == 10 || d == 11 || d == 12 || d == 13)) return false;
  }
  return true;
}

hipError_t launch_mixed(const MixedLaunch& a) {
  if (a.B == 0) return hipSuccess;
  switch (a.cls) {
    case 0: return launch_mixed_cls<0>(a);
    case 1: return launch_mixed_cls<1>(a);
    case 3: return launch_mixed_cls<3>(a);
    default: return launch_mixed_cls<2>(a);
  }
}

}  // namespace lolhip
