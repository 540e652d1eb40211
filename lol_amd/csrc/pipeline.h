// lol_amd/csrc/pipeline.h — launcher interface of pipeline.hip (SURVEY.md §8f N1 kernels).
#pragma once
#include <hip/hip_runtime_api.h>

#include "zq_dev.h"

namespace lolhip {

constexpr int PIPE_MAX_T = 16;   // RNS components a pipeline call accepts (parameter structs travel by value)

struct DecompParams {
  int T;                 // components
  int L;                 // total digits = sum k[t]
  i64 base;              // 0: TrivGad (ZqBasic.hs:227-232); >= 2: BaseBGad b (ZqBasic.hs:258-264)
  u64 magic;             // floor(2^64 (2^l - base) / base) + 1,  l = ceil(log2 base)
  int sh1, sh2;          // min(l,1), max(l-1,0)
  int k[PIPE_MAX_T];     // digits of component t: gadlen(base, q_t) (1 for TrivGad)
};

struct RescaleParams {
  int T;                       // components of the input; component 0 is dropped
  u64 qa_inv[PIPE_MAX_T];      // q_0^-1 mod q_s, s >= 1
};

hipError_t launch_ctmul(hipStream_t s, const i64* c0, const i64* c1, const i64* d0, const i64* d1, i64* e0, i64* e1,
                        i64* e2, const i64* gcrt, i64 B, i64 n, int T, const ModCtx* mod);
hipError_t launch_decompose(hipStream_t s, const i64* c, i64* digits, i64 B, i64 n, const DecompParams& p,
                            const ModCtx* mod, bool q32 = false);   // q32: every modulus below 2^31
hipError_t launch_knapsack(hipStream_t s, const i64* xs, int L, const i64* hint, int K, const i64* addend, i64* out,
                           i64 B, i64 n, int T, const ModCtx* mod, bool q32 = false);   // q32: every modulus below 2^29 (64-bit accumulators)
hipError_t launch_rescale(hipStream_t s, const i64* c, i64* out, i64 B, i64 n, const RescaleParams& p,
                          const ModCtx* mod);

// coeffs (Extension.hs:90-93): out[i1][b][i0][t] = in[b][idx[i1*n_lo + i0]][t]
hipError_t launch_coeffs(hipStream_t s, i64* out, const i64* in, const int32_t* idx, i64 B, i64 n_lo, i64 n_hi, int T,
                         const ModCtx* mod);

}  // namespace lolhip
