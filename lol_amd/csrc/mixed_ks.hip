// lol_amd/csrc/mixed_ks.hip — k_mixed_keyswitch: the fused key switch for ANY index the vector interpreter takes
// (every prime of m <= 13, n <= 8192) in arithmetic class 2 (every modulus odd and below 2^30.15 — where the
// reference's own key-switch benchmark lives: F64*F9*F25 with Zq (1008001 ** 1065601),
// lol-apps/Crypto/Lol/Applications/Benchmarks/Default.hs:49-50).
//
//   out_k = addend_k + sum_j crt(reduce(digit_j(c2))) * hint_jk,   k = 0, 1
// — `switch` of SymmSHE.hs:312-314 over Cyc.hs:592-604 (decompose), Gadget.hs:96-101, ZqBasic.hs:227-264 — in ONE pass.
// The three-launch path (k_decompose -> L*B polynomials through the interpreter -> k_knapsack) moves 3 + 4L slabs
// through a digit slab in HBM (1.03 ms per 2048 ciphertexts at m = 14400, base 256, round 2); here one work item =
// (ciphertext, target component s) keeps everything on chip, like k_keyswitch does for m = 2^k:
//   * the thread that owns position x lifts c2's component t at x (centred), peels its base-b digits in registers,
//     reduces each into Z_{q_s} and writes it to the LDS polynomial;
//   * the plan's forward stage program (the SAME run_stages the poly-mul uses) transforms it in place;
//   * the thread multiply-accumulates its positions with the two hint coefficients (L2-resident, shared by the batch)
//     into 64-bit sums of raw products (each below 2^60.3: folded every 8 digits), reduced once at the end.
// HBM sees c2 once per target component (L2 hits after the first), the addends once, the outputs once: 5 slabs.
#include "mixed_impl.h"

namespace lolhip {

// low dword of an int64 residue: for q < 2^31 a representative in (-q, q) is determined by it (bit 31 = sign)
__device__ __forceinline__ u32 load_lo32(rsrc_t r, u32 voff, u32 soff) { return __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0); }
__device__ __forceinline__ u32 canon_lo32(u32 x, u32 q) { return x + (q & (u32)((int)x >> 31)); }

// BIG: the stage program may hold the 18-/20-element vectors of merged prime powers (mixed_impl.h run_stages);
// instantiated for <= 12 coefficients per thread only (the accumulator sets of 16 leave no room for a 20-vector)
template <int KMAX, bool BIG = false>
__global__ void __launch_bounds__(512, 4)
k_mixed_keyswitch(const i64* __restrict__ c2, const i64* __restrict__ hint, const i64* addend, i64* out, i64 B, int T, int n,
                  const Stage* __restrict__ st_crt, int n_crt, const u32* __restrict__ consts32, int cpc,
                  const ModCtx* __restrict__ mod, DecompParams dp, u32 magic32) {
  constexpr int CLS = 2;
  using V = MV<CLS>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  V* buf = reinterpret_cast<V*>(smem);
  const u64 n_magic = (((u64)1 << 40) / (u64)n) + 1;
  const int nthr = blockDim.x, tid = threadIdx.x;
  const i64 items = B * T;
  const u64 slab = (u64)B * n * T;
  const u32 uT8 = (u32)T * 8u;
  const u32 step = (u32)nthr * uT8;
  const int shift = (int)(dp.base / 2);
  const u32 base = (u32)dp.base;
  for (i64 item = blockIdx.x; item < items; item += gridDim.x) {
    const i64 b = item / T;
    const int s = (int)(item % T);
    const ModCtx ms = mod[s];
    const u32 qs = (u32)ms.q, mu32 = (u32)(ms.mu >> 32);
    const PT<CLS>* cst = consts32 + (size_t)s * cpc;
    const size_t pbase = (size_t)b * n * T;                       // element (x, t) of this ciphertext at pbase + x * T + t
    const u32 pbytes = (u32)n * uT8;
    const rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void*)(c2 + pbase), 0, pbytes, 0x00020000);
    const rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc((void*)hint, 0, (u32)dp.L * 2u * pbytes, 0x00020000);
    const u32 o0 = (u32)fresh(tid) * uT8;
    auto fold = [&](u64 x) -> u64 {                               // x mod q_s, any 64-bit x
      const u64 r = x - __umul64hi(x, ms.mu) * ms.q;             // [0, 2q)
      return (u64)min((u32)r, (u32)r - qs);
    };
    u64 acc0[KMAX], acc1[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) { acc0[k] = 0; acc1[k] = 0; }
    int j = 0;
    for (int t = 0; t < T; ++t) {
      const u32 qt = (u32)mod[t].q;
      int vv[KMAX];                                               // centred lift of component t, then its running quotient
      {
        u32 raw[KMAX];
#pragma unroll
        for (int k = 0; k < KMAX; ++k) raw[k] = (k * nthr < n) ? load_lo32(rc, o0, (u32)k * step + (u32)t * 8u) : 0;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
          const u32 x = canon_lo32(raw[k], qt);                              // (-q,q) -> [0,q)
          vv[k] = (2 * x < qt) ? (int)x : (int)x - (int)qt;                  // ZqBasic.hs:92-94
        }
      }
      const int kt = dp.k[t];
      for (int kd = 0; kd < kt; ++kd, ++j) {
        {
          const int x0 = fresh(tid);
#pragma unroll
          for (int k = 0; k < KMAX; ++k) {
            int d;
            if (kd + 1 < kt) {                                    // centred remainder, quotient carries on (Numeric.hs:202-205,227-234)
              const int a = vv[k] + shift;
              const u32 nn = a >= 0 ? (u32)a : (u32)(-a - 1);
              const u32 t1 = __umulhi(magic32, nn);
              const u32 qq = (t1 + ((nn - t1) >> dp.sh1)) >> dp.sh2;
              const int qd = a >= 0 ? (int)qq : -(int)qq - 1;
              d = a - qd * (int)base - shift;
              vv[k] = qd;
            } else {
              d = vv[k];
            }
            const u32 ad = d >= 0 ? (u32)d : (u32)(-d);           // reduce into component s
            u32 r = ad - __umulhi(ad, mu32) * qs;                 // [0, 2q)
            r = min(r, r - qs);
            const int x = x0 + k * nthr;
            if (x < n) buf[x] = (d < 0 && r != 0) ? qs - r : r;
          }
        }
        __syncthreads();
        run_stages<CLS, false, BIG>(buf, n, n, n_magic, st_crt, n_crt, cst, ms);      // ends with a barrier
        {
          const u32 hoff = (u32)j * 2u * pbytes + (u32)s * 8u;
          const int x0 = fresh(tid);
          u32 h0[KMAX], h1[KMAX];
#pragma unroll
          for (int k = 0; k < KMAX; ++k) {
            const bool in = k * nthr < n;
            h0[k] = in ? load_lo32(rh, o0, (u32)k * step + hoff) : 0;
            h1[k] = in ? load_lo32(rh, o0, (u32)k * step + hoff + pbytes) : 0;
          }
#pragma unroll
          for (int k = 0; k < KMAX; ++k) {
            const int x = x0 + k * nthr;
            const u32 vc = x < n ? buf[x] : 0;
            // hint residues may be any representative in (-q_s, q_s), like every other input
            acc0[k] += (u64)canon_lo32(h0[k], qs) * vc;
            acc1[k] += (u64)canon_lo32(h1[k], qs) * vc;
          }
        }
        if ((j & 7) == 7) {
#pragma unroll
          for (int k = 0; k < KMAX; ++k) { acc0[k] = fold(acc0[k]); acc1[k] = fold(acc1[k]); }
        }
        __syncthreads();                                          // the buffer takes the next digit
      }
    }
    const rsrc_t ro0 = __builtin_amdgcn_make_buffer_rsrc((void*)(out + pbase), 0, pbytes, 0x00020000);
    const rsrc_t ro1 = __builtin_amdgcn_make_buffer_rsrc((void*)(out + slab + pbase), 0, pbytes, 0x00020000);
    const rsrc_t ra0 = __builtin_amdgcn_make_buffer_rsrc((void*)(addend ? addend + pbase : nullptr), 0, addend ? pbytes : 0, 0x00020000);
    const rsrc_t ra1 = __builtin_amdgcn_make_buffer_rsrc((void*)(addend ? addend + slab + pbase : nullptr), 0, addend ? pbytes : 0, 0x00020000);
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      if (k * nthr >= n) continue;
      const u32 eo = (u32)k * step + (u32)s * 8u;
      u32 r0 = (u32)fold(acc0[k]), r1 = (u32)fold(acc1[k]);
      if (addend) {
        r0 = m_add<CLS>(r0, canon_lo32(load_lo32(ra0, o0, eo), qs), ms.q);
        r1 = m_add<CLS>(r1, canon_lo32(load_lo32(ra1, o0, eo), qs), ms.q);
      }
      store_u64(ro0, o0, eo, (u64)r0);                             // lanes past n: dropped by the range check
      store_u64(ro1, o0, eo, (u64)r1);
    }
  }
}

// does the fused kernel take this key switch?  (class-2 plan of the vector interpreter, two hint coefficients,
// a base whose digits fit 32-bit arithmetic, hints addressable through one buffer descriptor)
static int ks_threads(i64 n) { return n > 2048 ? 512 : (n > 1024 ? 256 : 128); }
// may this key switch run a program with 18-/20-element vectors?  (at most 12 coefficients per thread)
bool mixed_keyswitch_big_ok(i64 n) { return ((size_t)n + ks_threads(n) - 1) / ks_threads(n) <= 12; }

hipError_t launch_mixed_keyswitch(const MixedKeySwitchLaunch& a) {
  if (a.B == 0) return hipSuccess;
  const size_t lds_bytes = (size_t)a.n * sizeof(u32);
  // few coefficients per thread: the two accumulator sets (4 VGPRs per coefficient) stay in registers across the stage program
  const int threads = ks_threads(a.n);
  i64 grid = a.B * a.T;
  if (grid > 65536) grid = 65536;
  const size_t per_thread = ((size_t)a.n + threads - 1) / threads;
#define LOLHIP_KS_LAUNCH(K, BIG)                                                                                           \
  hipLaunchKernelGGL((k_mixed_keyswitch<K, BIG>), dim3((unsigned)grid), dim3(threads), lds_bytes, a.stream, a.c2, a.hint,  \
                     a.addend, a.out, a.B, a.T, (int)a.n, a.st_crt, a.n_crt, a.consts32, a.cpc, a.mod, a.dp, a.magic32)
  if (a.big && per_thread > 12) return hipErrorInvalidValue;      // the caller picks the program (mixed_keyswitch_big_ok)
  if (per_thread <= 8) { if (a.big) LOLHIP_KS_LAUNCH(8, true); else LOLHIP_KS_LAUNCH(8, false); }
  else if (per_thread <= 12) { if (a.big) LOLHIP_KS_LAUNCH(12, true); else LOLHIP_KS_LAUNCH(12, false); }
  else LOLHIP_KS_LAUNCH(16, false);
#undef LOLHIP_KS_LAUNCH
  return hipGetLastError();
}

}  // namespace lolhip
