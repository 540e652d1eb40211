// lol_amd/csrc/kernels.h — launcher interface between the C ABI (capi.cpp) and kernels.hip.
#pragma once
#include <hip/hip_runtime_api.h>

#include "pipeline.h"
#include "plan.h"

namespace lolhip {

static const int32_t EMBED_NEG_FLAG_DEV = 1 << 30;   // == hostmath.h EMBED_NEG_FLAG

struct Pow2Launch {
  hipStream_t stream;
  i64* y;            // in-place operand (modes 0,1) or output c (mode 2)
  const i64* a;      // mode 2 inputs
  const i64* b;
  i64 B;
  int T;
  int L;
  const void *tw_fwd, *tw_inv, *scale;   // Shoup-pair tables: u64 pairs (arith 0/1) or u32 pairs (arith 2)
  const ModCtx* mod;
  int arith;         // 4: every modulus < 2^27; 2: < 2^30; 3: < 2^31 (32-bit paths); 1: every modulus < 2^61; 0: exact 64-bit
};
// mode 0 = crt, 1 = crtInv, 2 = fused poly-mul
hipError_t launch_pow2(const Pow2Launch& a, int mode);

// mode 2 only: the persistent, LDS-DMA-pipelined fused poly-mul of the 32-bit classes (pow2_pipe.hip)
bool pow2_pipe_ok(const Pow2Launch& a, bool forced);
hipError_t launch_pow2_pipe(const Pow2Launch& a);

// fused key switch (m = 2^k, 32-bit arithmetic class, two hint coefficients)
struct KeySwitchLaunch {
  hipStream_t stream;
  const i64* c2;        // [B][n][T] powerful basis
  const i64* hint;      // [L][2][n][T] CRT basis
  const i64* addend;    // [2][B][n][T] CRT basis or null
  i64* out;             // [2][B][n][T]
  i64 B;
  int T;
  int L;                // n = 2^L
  const uint32_t* tw_fwd32;
  const ModCtx* mod;
  DecompParams dp;
  uint32_t magic32;     // 32-bit invariant-divisor constant of dp.base
  int arith;            // 2 or 4 (Pow2Launch::arith)
};
hipError_t launch_keyswitch_fused(const KeySwitchLaunch& a);

struct GenericLaunch {
  hipStream_t stream;
  i64* y;
  i64 B;
  int T;
  i64 n;
  const Stage* stages;
  int nstages;
  const u64* consts;
  int cpc;
  const ModCtx* mod;
  u64* scratch;
  size_t scratch_bytes;
  bool q32;          // every modulus < 2^32: 32-bit operand products in the dot-product stages
};
hipError_t launch_generic(const GenericLaunch& a);

// vector-per-thread interpreter + fused mixed-radix poly-mul (mixed.hip); every prime <= 13, n <= 8192
struct MixedLaunch {
  hipStream_t stream;
  i64* y;                // output
  const i64* a;          // input (may equal y) / first operand
  const i64* b;          // second operand of the fused poly-mul
  i64 B;
  int T;
  i64 n;
  const Stage* st_a; int n_a;     // the program, or crt for the fused poly-mul
  const Stage* st_b; int n_b;     // crtInv for the fused poly-mul
  bool big = false;               // a program holds 18- or 20-element vectors (merged prime powers, class 2): the BIG kernels
  const u64* consts;              // the pool (classes 2 and 3: the Montgomery copy)
  const uint32_t* consts32;       // its 32-bit copy (classes 1 and 2), else null
  int cpc;
  const ModCtx* mod;
  int cls;               // Plan::mixed_cls (plan.cpp): 0/1 exact division, 2/3 Montgomery with `consts` = the
                         // pool pre-scaled by 2^32 / 2^64
  bool fused;
};
bool mixed_ok(i64 n, const Stage* host_stages, int nstages, const u64* qs, int T);
// fused key switch on the vector interpreter, arithmetic class 2 (mixed_ks.hip): any m with every prime <= 13, n <= 8192
struct MixedKeySwitchLaunch {
  hipStream_t stream;
  const i64* c2;        // [B][n][T] powerful basis
  const i64* hint;      // [L][2][n][T] CRT basis
  const i64* addend;    // [2][B][n][T] CRT basis or null
  i64* out;             // [2][B][n][T]
  i64 B;
  int T;
  i64 n;
  const Stage* st_crt; int n_crt;   // the plan's forward program (device copy)
  const uint32_t* consts32;         // class 2's 32-bit Montgomery pool
  int cpc;
  const ModCtx* mod;
  DecompParams dp;
  uint32_t magic32;     // 32-bit invariant-divisor constant of dp.base
  bool big = false;     // st_crt holds 18-/20-element vectors: only where mixed_keyswitch_big_ok(n)
};
bool mixed_keyswitch_big_ok(i64 n);
hipError_t launch_mixed_keyswitch(const MixedKeySwitchLaunch& a);
hipError_t launch_mixed(const MixedLaunch& a);

// floating-point side (floatpath.hip): y [B][n] complex doubles (re, im interleaved) / doubles, in place
hipError_t launch_cplx(hipStream_t s, double* y, i64 B, i64 n, const Stage* stages, int nstages, const double* cconsts);
hipError_t launch_gauss(hipStream_t s, double* y, i64 B, i64 n, const Stage* stages, int nstages, const double* rconsts);

// 16-byte-per-lane copy of a slab: the measured ceiling of a read-once/write-once kernel (bench.py's yardstick)
hipError_t launch_copy16(hipStream_t s, void* dst, const void* src, size_t bytes, int variant);
hipError_t launch_pointwise_mul(hipStream_t s, i64* a, const i64* b, i64 total, i64 bperiod, int T, const ModCtx* mod);
// replicating: every output has a source (embedCRT) — worth staging the source polynomial in LDS
hipError_t launch_gather(hipStream_t s, i64* out, const i64* in, const int32_t* idx, i64 B, i64 n_out, i64 n_in,
                         int T, const ModCtx* mod, bool replicating = false);
hipError_t launch_twace_crt(hipStream_t s, i64* out, const i64* in, const int32_t* idx, const i64* tweak, i64 B,
                            i64 n_out, i64 n_in, int T, const ModCtx* mod);

}  // namespace lolhip
