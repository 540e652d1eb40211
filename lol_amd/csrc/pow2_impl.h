// lol_amd/csrc/pow2_impl.h — device templates of the m = 2^k path (k_pow2 and everything it
// is made of) plus its launcher templates.  Included by one translation unit per arithmetic
// class (pow2_ar{0,1,2,3}.hip, each an explicit instantiation of launch_pow2_ar<AR>, so the
// classes compile in parallel) and by kernels.hip (the fused key switch reuses the transforms).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <mutex>
#include <type_traits>
#include <utility>

#include "kernels.h"
#include "zq_dev.h"

namespace lolhip {

// =============================================================================
// power-of-two path
// =============================================================================
// Index convention (Tensor.hs:359-368, verified against the reference):
//   crt:  Y[i] = sum_j a[j] * psi^(bitrev(j) * (2i+1)),  psi = omega_m, m = 2n.
// i.e. the stored powerful basis is bit-reversed, the CRT output is in natural
// order.  That is exactly an in-place decimation-in-time network: level s = 1..L
// combines positions x and x + 2^(s-1) inside blocks of N = 2^s with the twiddle
// psi_N^(2i+1), i = x mod N/2, psi_N = psi^(n/N).  The inverse runs the levels
// backwards with Gentleman-Sande butterflies and folds mhat^-1 into level 1.
//
// Data movement.  A polynomial is owned by n/16 threads, 16 coefficients each in
// registers.  Which coefficient sits where is a compile-time LAYOUT: every bit of the
// position x is assigned either to one of the 4 register-index bits or to a thread-index
// bit.  A butterfly level on position bit beta needs beta on a register bit, so the
// transform is a sequence of
//   - 4 levels on the current register bits,
//   - an LDS transpose to the next layout (write at x in layout A, read at x in layout B),
//   - for the 1-2 levels left over when 4 does not divide log2 n: a cross-lane
//     v_permlane32_swap / v_permlane16_swap that exchanges a LANE bit with a register bit
//     (no LDS, no barrier) instead of a fourth transpose.
// Global loads/stores always use a layout whose lane bits are the low position bits
// (consecutive lanes touch consecutive coefficients).

constexpr int R = 4;
constexpr int E = 1 << R;
// levels whose eight twiddles are all distinct are fetched and consumed in this many parts
// (2 or 4) in the register-lean schedule: 16 or 8 twiddle VGPRs live instead of 32
#ifndef LOLHIP_LEVEL_PARTS
#define LOLHIP_LEVEL_PARTS 4
#endif

// Diagnostic build (-DLOLHIP_STAMPS): per-wave s_memtime stamps at phase boundaries, written
// to a side buffer nobody else reads.  Never enabled in the shipped library.
#ifdef LOLHIP_STAMPS
static __device__ unsigned long long* g_stamp_buf = nullptr;   // per translation unit (development builds only)
#define LH_STAMP(i) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
    if (g_stamp_buf && (threadIdx.x & 63) == 0) g_stamp_buf[((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 32 + (i)] = t_; } while (0)
#else
#define LH_STAMP(i) do {} while (0)
#endif

struct Lay {
  int reg[R];     // position bit held by register-index bit k
  int thr[12];    // position bit held by thread-index bit j
  int ntb;        // number of thread bits (log2 n - 4)
};
constexpr bool lay_eq(const Lay& a, const Lay& b) {
  if (a.ntb != b.ntb) return false;
  for (int k = 0; k < R; ++k) if (a.reg[k] != b.reg[k]) return false;
  for (int j = 0; j < a.ntb; ++j) if (a.thr[j] != b.thr[j]) return false;
  return true;
}
// registers hold bits lo..lo+3, threads the remaining bits in ascending order
constexpr Lay lay_std(int L, int lo) {
  Lay a{};
  a.ntb = L - R;
  for (int k = 0; k < R; ++k) a.reg[k] = lo + k;
  int j = 0;
  for (int x = 0; x < L; ++x) if (x < lo || x >= lo + R) a.thr[j++] = x;
  return a;
}
// registers hold bits lo..lo+3; the `nhi` bits above them sit on lane bits lb0 (, lb1);
// the bits below fill the other thread bits in ascending order
constexpr Lay lay_lane_hi(int L, int lo, int nhi, int lb0, int lb1) {
  Lay a{};
  a.ntb = L - R;
  for (int k = 0; k < R; ++k) a.reg[k] = lo + k;
  int x = 0;
  for (int j = 0; j < a.ntb; ++j) {
    if (j == lb0) a.thr[j] = lo + R;
    else if (nhi == 2 && j == lb1) a.thr[j] = lo + R + 1;
    else a.thr[j] = x++;
  }
  return a;
}
constexpr Lay lay_swap(Lay a, int tb, int rk) {
  const int t = a.thr[tb];
  a.thr[tb] = a.reg[rk];
  a.reg[rk] = t;
  return a;
}
constexpr int xreg(const Lay& a, int e) {
  int x = 0;
  for (int k = 0; k < R; ++k) x |= ((e >> k) & 1) << a.reg[k];
  return x;
}
// position bits contributed by the thread index (runs of consecutive bits move together).
// The run decomposition is forced through a constexpr object: left as a loop over the
// template-parameter object, hipcc emits a RUNTIME loop of dependent global loads.
struct LayRuns { int n; int src[12]; int len[12]; int dst[12]; };
constexpr LayRuns lay_runs(const Lay& a) {
  LayRuns r{};
  for (int j = 0; j < a.ntb;) {
    int len = 1;
    while (j + len < a.ntb && a.thr[j + len] == a.thr[j] + len) ++len;
    r.src[r.n] = j; r.len[r.n] = len; r.dst[r.n] = a.thr[j];
    ++r.n;
    j += len;
  }
  return r;
}
template <Lay A, int I>
__device__ __forceinline__ int xthr_run(int tau) {
  constexpr LayRuns r = lay_runs(A);
  if constexpr (I < r.n) return (((tau >> r.src[I]) & ((1 << r.len[I]) - 1)) << r.dst[I]) | xthr_run<A, I + 1>(tau);
  else return 0;
}
template <Lay A>
__device__ __forceinline__ int xthr(int tau) { return xthr_run<A, 0>(tau); }

// per-layout compile-time tables (register part of x, twiddle index and slot per butterfly)
template <Lay A> struct LayTab {
  int xr[E];
  constexpr LayTab() : xr{} { for (int e = 0; e < E; ++e) xr[e] = xreg(A, e); }
};
template <Lay A> inline constexpr LayTab<A> lay_tab{};

// one padding word per 16: lpad(a|b) = lpad(a) + lpad(b) for bit-disjoint a, b, so the
// register part of every LDS address is an immediate offset
constexpr int lpad(int x) { return x + (x >> 4); }

// ---- arithmetic flavours, chosen per plan on the host (template parameter AR) -------------
//  AR = 2 (every q_t < 2^30): 32-bit residues, 32-bit Shoup products (3 multiplies per butterfly),
//         Harvey's lazy ranges [0,4q) forward / [0,2q) inverse.  This is the reference's own
//         correct domain (its Zq overflows beyond ~2^31.5, types.h:79-84) and the HBM-bound case.
//  AR = 3 (every q_t < 2^31, some >= 2^30): 32-bit residues again, but 4q no longer fits a word:
//         forward values in [0,2q), inverse values canonical, two conditional subtractions per
//         butterfly (10 instructions against 8 for AR = 2 and ~25 for AR = 1).  Covers the rest
//         of the reference's correct domain (goodQs above 2^30, e.g. 1073872897, 2148249601 is out).
//  AR = 1 (every q_t < 2^61): 64-bit residues, Shoup products with the 9-multiply approximate
//         quotient (shoup_acc, result in [0,4q)); forward values in [0,8q), inverse in [0,4q).
//  AR = 0 (2^61 <= q_t < 2^62): 10-multiply exact quotient, ranges [0,4q) / [0,2q).
//  AR = 4 (every q_t < 2^27 — the reference's own benchmark moduli: 12289, 1017857, 1032193 ...):
//         as AR = 2, but the forward transform does NO conditional subtraction: a level grows the
//         bound by 2q, so after 14 levels values are below 29 q < 2^32; one Barrett step at the
//         end.  7 instructions per forward butterfly against 9.
template <int AR> using VT = std::conditional_t<AR >= 2, u32, u64>;

// Per-modulus constants of the lazy butterflies (wave-uniform, live in SGPRs).
struct QK {
  u64 q, nq, q2, nq2, q4, nq4;
  // UNIFORM: the modulus is the same for every lane of the wave (SGPRs), else per lane (VGPRs)
  template <bool UNIFORM>
  __device__ __forceinline__ QK(const ModCtx& mc, std::bool_constant<UNIFORM> u) : QK(mc.q, u) {}
  template <bool UNIFORM>
  __device__ __forceinline__ QK(u64 q_, std::bool_constant<UNIFORM>) : q(q_), nq(0 - q_), q2(2 * q_), nq2(0 - 2 * q_), q4(4 * q_), nq4(0 - 4 * q_) {
    // opaque to the optimiser: x + nq4 must stay ONE v_lshl_add_u64, not be rewritten as the
    // two-instruction borrow chain x - q4
    if constexpr (UNIFORM) asm volatile("" : "+s"(nq), "+s"(nq2), "+s"(nq4));
    else asm volatile("" : "+v"(nq), "+v"(nq2), "+v"(nq4));
  }
};
struct QK32 {
  u32 q, q2, mu;       // mu = floor(2^32 / q): one-step Barrett of the wide lazy range of AR = 4
  u32 r, rp, nqinv;    // Shoup pair of 2^32 mod q and -q^-1 mod 2^32: the fused poly-mul's Montgomery pointwise product
  template <bool UNIFORM>
  __device__ __forceinline__ QK32(const ModCtx& mc, std::bool_constant<UNIFORM>)
      : q((u32)mc.q), q2(2 * (u32)mc.q), mu((u32)(mc.mu >> 32)), r(mc.r32), rp(mc.r32p), nqinv((u32)mc.nqinv) {}
};
template <int AR> using QKT = std::conditional_t<AR >= 2, QK32, QK>;

__device__ __forceinline__ u32 csub32(u32 x, u32 m) { return min(x, x - m); }        // x < 2m
// w*y mod q in [0,2q) for any 32-bit y
__device__ __forceinline__ u32 shoup32(u32 y, u32 w, u32 wp, u32 q) { return w * y - __umulhi(wp, y) * q; }

// ---- pair/mad form of the 32-bit lazy butterflies (LOLHIP_PAIR_BFLY: pow2_pipe.hip only) -----------------------
// A residue sits in the LOW half of a 64-bit register pair (the high half is never read), so v_mad_u64_u32 does the
// multiply AND the add: x + w y - Q q as two chained mads (nq = -q mod 2^32), 2x + 2q - X' as a mad by -1.  Forward 5
// instructions instead of 7, inverse 7 instead of 9; the same values mod 2^32 as the 32-bit forms below, so the same
// lazy ranges.  Twice the data registers: for the persistent kernel (4 waves per SIMD), not the 8-wave ones.
// tools/microbench_bfly32.hip: -13 % / -6 % per butterfly, exact.
#ifndef LOLHIP_PAIR_BFLY
#define LOLHIP_PAIR_BFLY 0
#endif
__device__ __forceinline__ u64 pair_of(u32 lo) { typedef u32 u32p __attribute__((ext_vector_type(2))); u32p t; t.x = lo; return __builtin_bit_cast(u64, t); }
// X' = x + w Y - Q q,  Y' = 2x + 2q - X'   (x, Y any 32-bit values)
template <bool WS>
__device__ __forceinline__ void fwd_pair(u32& X, u32& Y, u32 x, u32 w, u32 wp, u32 q, u32 q2) {
  const u64 xp = pair_of(x), q2p = (u64)q2;
  const u32 nq = 0u - q;
  u64 Xn, Z, Yn, cy; u32 Q;
#define LH_FWD_PAIR(WC)                                                                                     \
  asm("v_mul_hi_u32 %[Q], %[wp], %[yl]\n\t"                                                                  \
      "v_mad_u64_u32 %[Xn], %[cy], %[w], %[yl], %[X]\n\t"                                                    \
      "v_lshl_add_u64 %[Z], %[X], 1, %[q2]\n\t"                                                              \
      "v_mad_u64_u32 %[Xn], %[cy], %[Q], %[nq], %[Xn]"                                                        \
      : [Q] "=&v"(Q), [Xn] "=&v"(Xn), [Z] "=&v"(Z), [cy] "=&s"(cy)                                            \
      : [wp] WC(wp), [yl] "v"(Y), [w] WC(w), [X] "v"(xp), [q2] "s"(q2p), [nq] "s"(nq))
  if constexpr (WS) { LH_FWD_PAIR("s"); } else { LH_FWD_PAIR("v"); }
#undef LH_FWD_PAIR
  asm("v_mad_u64_u32 %[Yn], %[cy], %[xl], -1, %[Z]" : [Yn] "=v"(Yn), [cy] "=s"(cy) : [xl] "v"((u32)Xn), [Z] "v"(Z));
  X = (u32)Xn; Y = (u32)Yn;
}
// w d - Q q for any 32-bit d
template <bool WS>
__device__ __forceinline__ u32 shoup_pair(u32 d, u32 w, u32 wp, u32 q) {
  const u32 nq = 0u - q;
  const u32 Q = __umulhi(wp, d);
  u64 T, Yn, cy;
#define LH_SH_PAIR(WC)                                                                                      \
  asm("v_mad_u64_u32 %[T], %[cy], %[w], %[d], 0\n\t"                                                         \
      "v_mad_u64_u32 %[Yn], %[cy], %[Q], %[nq], %[T]"                                                         \
      : [T] "=&v"(T), [Yn] "=&v"(Yn), [cy] "=&s"(cy) : [w] WC(w), [d] "v"(d), [Q] "v"(Q), [nq] "s"(nq))
  if constexpr (WS) { LH_SH_PAIR("s"); } else { LH_SH_PAIR("v"); }
#undef LH_SH_PAIR
  return (u32)Yn;
}

// forward (Cooley-Tukey) butterfly:  X' = X + w*Y,  Y' = X - w*Y
// WS: the twiddle is wave-uniform (SGPR operands in the 64-bit class's multiply chains, zq_dev.h)
template <int AR, bool WS = false>
__device__ __forceinline__ void bfly_fwd(VT<AR>& X, VT<AR>& Y, VT<AR> w, VT<AR> wp, const QKT<AR>& k) {
  if constexpr (AR == 3) {
    const u32 x = csub32(X, k.q);                     // [0,2q) -> [0,q)
    const u32 t = csub32(shoup32(Y, w, wp, k.q), k.q);
    X = x + t;                                        // [0,2q)
    Y = x - t + k.q;                                  // (0,2q)
  } else if constexpr (AR == 2) {
    const u32 x = csub32(X, k.q2);                    // [0,4q) -> [0,2q)
#if LOLHIP_PAIR_BFLY
    fwd_pair<WS>(X, Y, x, w, wp, k.q, k.q2);
#else
    const u32 t = shoup32(Y, w, wp, k.q);
    X = x + t;
    Y = x - t + k.q2;
#endif
  } else if constexpr (AR == 4) {
#if LOLHIP_PAIR_BFLY
    fwd_pair<WS>(X, Y, X, w, wp, k.q, k.q2);
#else
    const u32 t = shoup32(Y, w, wp, k.q);             // [0,2q) for ANY 32-bit Y
    const u32 x = X;                                  // < B: both outputs < B + 2q, never wrapping below 29 q
    X = x + t;
    Y = x - t + k.q2;
#endif
  } else if constexpr (AR == 1) {
#ifdef LH_ABL_NO_FWD_CSUB
    const u64 x = X;                                  // timing-only ablation: results are garbage
#else
    const u64 x = csubn(X, k.nq4);                    // [0,8q) -> [0,4q)
#endif
    const u64 xn = shoup_acc<WS>(Y, w, wp, k.nq, x);  // x + t, t in [0,4q)
    const u64 z = shl1_add64u(x, k.q4);                // 2x + 4q
    X = xn;
    Y = z - xn;                                       // x - t + 4q
  } else {
    const u64 x = csub(X, k.q2);
    const u64 t = shoup_lazy(Y, w, wp, k.q);
    X = x + t;
    Y = x - t + k.q2;
  }
}
// inverse (Gentleman-Sande) butterfly:  X' = X + Y,  Y' = (X - Y) * w
template <int AR, bool WS = false>
__device__ __forceinline__ void bfly_inv(VT<AR>& X, VT<AR>& Y, VT<AR> w, VT<AR> wp, const QKT<AR>& k) {
  if constexpr (AR == 3) {                            // canonical in, canonical out
    const u32 s = X + Y;
    const u32 d = X - Y + k.q;
    X = csub32(s, k.q);
    Y = csub32(shoup32(d, w, wp, k.q), k.q);
  } else if constexpr (AR == 2 || AR == 4) {
    const u32 s = X + Y;
    const u32 d = X - Y + k.q2;
    X = csub32(s, k.q2);
#if LOLHIP_PAIR_BFLY
    Y = shoup_pair<WS>(d, w, wp, k.q);
#else
    Y = shoup32(d, w, wp, k.q);
#endif
  } else if constexpr (AR == 1) {
    const u64 s = add64(X, Y);                        // [0,8q)
    const u64 d = add64u(X, k.q4) - Y;                 // (0,8q)
    X = csubn(s, k.nq4);
    Y = shoup_acc<WS>(d, w, wp, k.nq, 0);
  } else {
    const u64 s = X + Y;
    const u64 d = X - Y + k.q2;
    X = csub(s, k.q2);
    Y = shoup_lazy(d, w, wp, k.q);
  }
}
// last inverse level: both outputs additionally scaled by mhat^-1 (crt.cpp:573-579).
// (s0,s1) = Shoup pair of mhat^-1; (w, wp) = Shoup pair of psi_2^-1 * mhat^-1.
template <int AR, bool WS = false>
__device__ __forceinline__ void bfly_inv_last(VT<AR>& X, VT<AR>& Y, VT<AR> w, VT<AR> wp, VT<AR> s0, VT<AR> s1, const QKT<AR>& k) {
  if constexpr (AR == 3) {
    const u32 s = X + Y;
    const u32 d = X - Y + k.q;
    X = csub32(shoup32(s, s0, s1, k.q), k.q);
    Y = csub32(shoup32(d, w, wp, k.q), k.q);
  } else if constexpr (AR == 2 || AR == 4) {
    const u32 s = X + Y;
    const u32 d = X - Y + k.q2;
    X = shoup32(s, s0, s1, k.q);
    Y = shoup32(d, w, wp, k.q);
  } else if constexpr (AR == 1) {
    const u64 s = add64(X, Y);
    const u64 d = add64u(X, k.q4) - Y;
    X = shoup_acc<WS>(s, s0, s1, k.nq, 0);            // WS: the scale constants are wave-uniform
    Y = shoup_acc<WS>(d, w, wp, k.nq, 0);
  } else {
    const u64 s = X + Y;
    const u64 d = X - Y + k.q2;
    X = shoup_lazy(s, s0, s1, k.q);
    Y = shoup_lazy(d, w, wp, k.q);
  }
}
template <int AR> __device__ __forceinline__ VT<AR> canon_fwd(VT<AR> v, const QKT<AR>& k) {
  if constexpr (AR == 3) return csub32(v, k.q);
  else if constexpr (AR == 2) return csub32(csub32(v, k.q2), k.q);
  else if constexpr (AR == 4) return csub32(v - __umulhi(v, k.mu) * k.q, k.q);     // any 32-bit v: v - floor(v mu / 2^32) q in [0,2q)
  else {
    if constexpr (AR == 1) v = csubn(v, k.nq4);
    return csubn(csubn(v, k.nq2), k.nq);
  }
}
template <int AR> __device__ __forceinline__ VT<AR> canon_inv(VT<AR> v, const QKT<AR>& k) {
  if constexpr (AR == 3) return v;
  else if constexpr (AR == 2 || AR == 4) return csub32(v, k.q);
  else {
    if constexpr (AR == 1) v = csubn(v, k.nq2);
    return csubn(v, k.nq);
  }
}
// reference-style input in (-q, q) -> [0, q)
template <int AR> __device__ __forceinline__ VT<AR> from_i64(i64 x, const QKT<AR>& k) {
  if constexpr (AR >= 2) return (u32)x + (k.q & (u32)(x >> 63));
  else return canon_in(x, k.q);
}
// the same on the way INTO a forward transform, whose lazy range takes more than [0,q): one instruction.
// 64-bit class: any value in [0,8q), and x + 4q is in (3q,5q) for every x in (-q,q) (one v_lshl_add_u64 instead
// of a sign test and a masked add).  32-bit classes: x + q is in (0,2q), inside [0,2q) (class 3), [0,4q) (class 2)
// and the wide range of class 4 (inputs below 2q: 2q + 14 * 2q = 30q < 2^32 for q < 2^27).
template <int AR> __device__ __forceinline__ VT<AR> from_i64_fwd(i64 x, const QKT<AR>& k) {
  if constexpr (AR == 1) return add64u((u64)x, k.q4);
  else if constexpr (AR >= 2) return (u32)x + k.q;
  else return from_i64<AR>(x, k);
}
// and into an inverse transform: 64-bit class range [0,4q): x + 2q is in (q,3q); classes 2 and 4 [0,2q): x + q;
// class 3's inverse works on canonical values
template <int AR> __device__ __forceinline__ VT<AR> from_i64_inv(i64 x, const QKT<AR>& k) {
  if constexpr (AR == 1) return add64u((u64)x, k.q2);
  else if constexpr (AR == 2 || AR == 4) return (u32)x + k.q;
  else return from_i64<AR>(x, k);
}
// the operand of the fused poly-mul that waits in registers: canonical in the 64-bit classes; in the
// 32-bit ones canonical AND multiplied by 2^32 (one Shoup product, valid for any lazy 32-bit value),
// so that the pointwise product below is a bare Montgomery reduction
template <int AR> __device__ __forceinline__ VT<AR> park_fwd(VT<AR> v, const QKT<AR>& k) {
  if constexpr (AR >= 2) return csub32(shoup32(v, k.r, k.rp, k.q), k.q);
  else if constexpr (AR == 1) return csubn(csubn(v, k.nq4), k.nq2);      // [0,8q) -> [0,2q): all the Montgomery product needs
  else return canon_fwd<AR>(v, k);
}
// pointwise product of a parked a-hat and a lazy b-hat (forward range), any range the
// inverse transform accepts
template <int AR> __device__ __forceinline__ VT<AR> pmul(VT<AR> a, VT<AR> b, const ModCtx& mc, const QKT<AR>& k) {
  if constexpr (AR >= 2) {
    // a = a-hat 2^32 mod q < q; b < 4q (AR = 2), 2q (AR = 3), 29q (AR = 4): x = a b < q 2^32.
    // REDC: m = x (-q^-1) mod 2^32; (x + m q) / 2^32 = a-hat b-hat mod q, below 2q; 4 instructions
    // (the 64-bit Barrett step this replaces: ~16)
    const u64 x = (u64)a * b;
    const u32 m = (u32)x * k.nqinv;
    const u32 t = (u32)((x + (u64)m * k.q) >> 32);
    if constexpr (AR == 3) return csub32(t, k.q);      // this class's inverse takes canonical values
    else return t;                                     // [0,2q): the inverse's lazy range
  } else if constexpr (AR == 1) {
#ifdef LOLHIP_ABL_PMUL      // ablation: what the pointwise product costs (0.028 of 0.52 ms with the exact division)
    return a + b;
#endif
    // a < 2q, b < 8q, q < 2^61: T = a b < 16 q^2 < 2^126.  Montgomery: m = T (-q^-1) mod 2^64;
    // (T + m q) / 2^64 = hi(T) + hi(m q) + [lo(T) != 0]  <  16 q^2 / 2^64 + q + 1 < 3q + 1: inside the inverse's
    // lazy range [0,4q) as it stands.  The factor 2^-64 is taken back by the last inverse level, whose
    // two constants the plan pre-multiplies by 2^64 mod q.  ~22 instructions against ~40 for the exact division.
    const unsigned __int128 T = (unsigned __int128)a * b;
    const u64 lo = (u64)T;
    return (u64)(T >> 64) + __umul64hi(lo * mc.nqinv, mc.q) + (lo != 0);
  } else {
    return mulmod(a, canon_fwd<0>(b, k), mc);
  }
}

// Global memory goes through buffer descriptors: address = base (SGPRs) + one 32-bit
// per-lane offset + a wave-uniform offset, so no 64-bit address lives in VGPRs.
typedef u32 u32x2 __attribute__((ext_vector_type(2)));
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

__device__ __forceinline__ void load_tw(rsrc_t tw, u32 voff, u32 const_idx, u64& w, u64& wp) {
  const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(tw, voff, const_idx * 16u, 0);
  w = ((u64)r.y << 32) | r.x;
  wp = ((u64)r.w << 32) | r.z;
}
__device__ __forceinline__ void load_tw(rsrc_t tw, u32 voff, u32 const_idx, u32& w, u32& wp) {
  const u32x2 r = __builtin_amdgcn_raw_buffer_load_b64(tw, voff, const_idx * 8u, 0);
  w = r.x;
  wp = r.y;
}
__device__ __forceinline__ u64 load_u64(rsrc_t r, u32 voff, u32 soff) {
#ifdef LOLHIP_ABL_NO_IO       // ablation: compute-only timing, results are garbage
  return (u64)voff * 0x9E3779B97F4A7C15ull + soff;
#endif
  const u32x2 x = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
  return ((u64)x.y << 32) | x.x;
}
__device__ __forceinline__ void store_u64(rsrc_t r, u32 voff, u32 soff, u64 val) {
  u32x2 x;
  x.x = (u32)val; x.y = (u32)(val >> 32);
#ifdef LOLHIP_ABL_NO_IO
  if (val != 0x1234567ull) return;
#endif
  __builtin_amdgcn_raw_buffer_store_b64(x, r, voff, soff, 0);
}

// 16 bytes per lane: coefficients x, x+1 of a single-modulus slab
__device__ __forceinline__ void load_u64x2(rsrc_t r, u32 voff, u32 soff, u64& x0, u64& x1) {
#ifdef LOLHIP_ABL_NO_IO
  x0 = (u64)voff * 0x9E3779B97F4A7C15ull + soff; x1 = x0 ^ soff; return;
#endif
  const u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
  x0 = ((u64)x.y << 32) | x.x;
  x1 = ((u64)x.w << 32) | x.z;
}
// The constant part of the address goes into voffset/the 12-bit immediate, NEVER into an SGPR soffset:
// hipcc (ROCm 7.2) pads "store of more than 8 bytes -> VALU write of its data registers" (2 wait states on
// gfx950) only when soffset is not a register, but MI355X needs the pad in the SGPR form too — a persistent
// loop's last store picked up the loop-bound compare's v_mov in 16 lanes (profiles/r03_store_hazard.txt;
// tools/check_store_hazard.py audits the assembly of every translation unit for the pair).
__device__ __forceinline__ void store_u64x2(rsrc_t r, u32 voff, u32 coff, u64 x0, u64 x1) {
  u32x4 x;
  x.x = (u32)x0; x.y = (u32)(x0 >> 32); x.z = (u32)x1; x.w = (u32)(x1 >> 32);
#ifdef LOLHIP_ABL_NO_IO
  if (x0 != 0x1234567ull) return;
#endif
  __builtin_amdgcn_raw_buffer_store_b128(x, r, voff + coff, 0, 0);
}
// whole-polynomial register I/O in layout A; element stride `ebytes` (= 8 T).  Layouts whose
// register bit 0 is position bit 0 (Sched<L, true>) move pairs with 16-byte accesses.
template <Lay A, bool W16, typename F>
__device__ __forceinline__ void load_poly(rsrc_t r, u32 off, u32 ebytes, F&& put) {
  if constexpr (W16 && A.reg[0] == 0) {
#pragma unroll
    for (int e = 0; e < E; e += 2) {
      u64 x0, x1;
      load_u64x2(r, off, (u32)lay_tab<A>.xr[e] * 8u, x0, x1);
      put(e, x0); put(e + 1, x1);
    }
  } else {
#pragma unroll
    for (int e = 0; e < E; ++e) put(e, load_u64(r, off, (u32)lay_tab<A>.xr[e] * ebytes));
  }
}
template <Lay A, bool W16, typename F>
__device__ __forceinline__ void store_poly(rsrc_t r, u32 off, u32 ebytes, F&& get) {
  if constexpr (W16 && A.reg[0] == 0) {
#pragma unroll
    for (int e = 0; e < E; e += 2) store_u64x2(r, off, (u32)lay_tab<A>.xr[e] * 8u, get(e), get(e + 1));
  } else {
#pragma unroll
    for (int e = 0; e < E; ++e) store_u64(r, off, (u32)lay_tab<A>.xr[e] * ebytes, get(e));
  }
}

// Where a level's twiddles come from:
//  * index bits all on registers (the thread part of x mod 2^beta is empty): wave-uniform,
//    read with SCALAR loads — no texture-path traffic at all;
//  * beta <= 8: from a per-workgroup LDS copy of table entries [16, 512) (levels 5..9) —
//    the texture path costs 16 clk per 1 KiB wave-load even when 64 lanes hit 16 addresses,
//    and at 60 such loads per wave it, not the ALU, was bounding the transform;
//  * otherwise (the top levels, whose tables are 8 KiB..64 KiB): buffer loads, L2-served.
constexpr int TWL_LO = 16, TWL_HI = 512;              // LDS-resident table entries [lo, hi)
constexpr int TWL_MIN_L = 11;                         // smaller polynomials: the copy costs more than it saves
constexpr int twl_words(int n) { return n >= (1 << TWL_MIN_L) ? 2 * (TWL_HI - TWL_LO) : 0; }

template <typename V> struct TwCtxT {
  rsrc_t fwd, inv;
  const V *pf, *pi;     // this component's tables as plain pointers (scalar loads)
  const V* lds_tw;      // LDS copy of entries [16,512) of the table currently in use
  u32 comp;             // byte offset of this RNS component's table
  V sc0, sc1;           // Shoup pair of mhat^-1 (times 2^64 in the 64-bit fused poly-mul)
  V l1w, l1wp;          // Shoup pair of the level-1 inverse twiddle times the same factor
};
template <Lay A, int K> constexpr bool tw_uniform() {
  for (int j = 0; j < A.ntb; ++j) if (A.thr[j] < A.reg[K]) return false;
  return true;
}

// ---- twiddles: fetched a whole register pass ahead ------------------------------------
// Level on register bit K of layout A: butterflies pair e and e|1<<K; the twiddle index is
// x mod 2^beta (beta = A.reg[K]), whose register part is a compile-time constant.  Threads
// issue the (deduplicated) loads for ALL levels of a pass before the LDS transpose that
// precedes it, so L2 latency overlaps the exchange and the barrier.
constexpr int tw_cidx(const Lay& a, int k, int e) {
  const int beta = a.reg[k];
  return (1 << beta) + (xreg(a, e) & ((1 << beta) - 1));
}
// ordinal (0..7) of the first butterfly of level k that uses the same twiddle as butterfly e
// number of distinct twiddles of level k in layout a
constexpr int tw_distinct(const Lay& a, int k) {
  int cnt = 0;
  for (int e = 0; e < E; ++e) {
    if (e & (1 << k)) continue;
    bool first = true;
    for (int f = 0; f < e; ++f) if (!(f & (1 << k)) && tw_cidx(a, k, f) == tw_cidx(a, k, e)) first = false;
    cnt += first ? 1 : 0;
  }
  return cnt;
}
constexpr int tw_slot(const Lay& a, int k, int e) {
  int ord = 0;
  for (int f = 0; f < E; ++f) {
    if (f & (1 << k)) continue;
    if (tw_cidx(a, k, f) == tw_cidx(a, k, e)) return ord;
    ++ord;
  }
  return 0;
}
template <Lay A, int K> struct LevelTab {
  int cidx[E], slot[E];
  constexpr LevelTab() : cidx{}, slot{} { for (int e = 0; e < E; ++e) { cidx[e] = tw_cidx(A, K, e); slot[e] = tw_slot(A, K, e); } }
};
template <Lay A, int K> inline constexpr LevelTab<A, K> level_tab{};
template <typename V> struct LevelTwT { V w[8], wp[8]; };

template <bool INV, Lay A, int K, int HALF = -1, typename V>
__device__ __forceinline__ void tw_fetch(LevelTwT<V>& t, const TwCtxT<V>& tw, int xt) {
  constexpr int beta = A.reg[K];
#ifdef LOLHIP_ABL_NO_TW       // ablation: no twiddle traffic
#pragma unroll
  for (int s = 0; s < 8; ++s) { t.w[s] = tw.sc0 + s; t.wp[s] = tw.sc1 + K; }
  return;
#endif
  const u32 voff = tw.comp + (u32)(xt & ((1 << beta) - 1)) * (u32)(2 * sizeof(V));
  const V* sp = INV ? tw.pi : tw.pf;
  int ord = 0;
#pragma unroll
  for (int e = 0; e < E; ++e) {
    if (e & (1 << K)) continue;
    if (level_tab<A, K>.slot[e] == ord && (HALF < 0 || ord / (8 / LOLHIP_LEVEL_PARTS) == HALF)) {
      const int cidx = level_tab<A, K>.cidx[e];
      if constexpr (tw_uniform<A, K>()) {
        // through the CONSTANT address space: a uniform load from it is an s_load whatever else the kernel does.
        // As a plain global load hipcc demotes it to a per-lane vector load as soon as it cannot prove the
        // table unclobbered (seen in pow2_pipe.hip, where it then queued behind the LDS-DMA in vmcnt order)
        if constexpr (sizeof(V) == 4) {
          typedef const __attribute__((address_space(4))) V* cptr_t;
          const cptr_t cp = (cptr_t)(sp + 2 * cidx);
          t.w[ord] = cp[0]; t.wp[ord] = cp[1];
        } else {        // the 64-bit classes always got s_load; through the constant space they spill 26 SGPRs instead of 6
          t.w[ord] = sp[2 * cidx]; t.wp[ord] = sp[2 * cidx + 1];
        }
      } else if constexpr (A.ntb + R >= TWL_MIN_L && (2 << beta) <= TWL_HI && (1 << beta) >= TWL_LO) {
        const V* lp = tw.lds_tw + 2 * (cidx - TWL_LO + (xt & ((1 << beta) - 1)));
        if constexpr (sizeof(V) == 8) { const ulonglong2 r = *reinterpret_cast<const ulonglong2*>(lp); t.w[ord] = r.x; t.wp[ord] = r.y; }
        else { const uint2 r = *reinterpret_cast<const uint2*>(lp); t.w[ord] = r.x; t.wp[ord] = r.y; }
      } else {
        load_tw(INV ? tw.inv : tw.fwd, voff, (u32)cidx, t.w[ord], t.wp[ord]);
      }
    }
    ++ord;
  }
}
// copy entries [16, 512) of one component's table into LDS (NT threads of one polynomial)
template <int NT, typename V>
__device__ __forceinline__ void tw_fill_lds(V* dst, rsrc_t src, u32 comp, int n, int tau) {
  const int cnt = (n < TWL_HI ? n : TWL_HI) - TWL_LO;
  for (int i = tau; i < cnt; i += NT) {
    if constexpr (sizeof(V) == 8) {
      const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(src, comp + (u32)(TWL_LO + i) * 16u, 0, 0);
      *reinterpret_cast<u32x4*>(dst + 2 * i) = r;
    } else {
      const u32x2 r = __builtin_amdgcn_raw_buffer_load_b64(src, comp + (u32)(TWL_LO + i) * 8u, 0, 0);
      *reinterpret_cast<u32x2*>(dst + 2 * i) = r;
    }
  }
}
#ifndef LOLHIP_WS
#define LOLHIP_WS 1      // wave-uniform twiddles as SGPR operands (A/B switch)
#endif
template <int AR, bool INV, Lay A, int K, int HALF = -1>
__device__ __forceinline__ void level(VT<AR> (&v)[E], const LevelTwT<VT<AR>>& t, const TwCtxT<VT<AR>>& tw, const QKT<AR>& qk) {
  constexpr int beta = A.reg[K];
#ifdef LOLHIP_ABL_NO_BFLY     // ablation: keep the twiddles live, skip the arithmetic
#pragma unroll
  for (int s = 0; s < 8; ++s) asm volatile("" :: "v"(t.w[s]), "v"(t.wp[s]));
  return;
#endif
  int ordb = -1;
#pragma unroll
  for (int e = 0; e < E; ++e) {
    if (e & (1 << K)) continue;
    ++ordb;
    if (HALF >= 0 && ordb / (8 / LOLHIP_LEVEL_PARTS) != HALF) continue;
    const int s = level_tab<A, K>.slot[e];
    // a wave holds one polynomial component (so per-component constants are wave-uniform) from n = 1024 up
    constexpr bool WU = LOLHIP_WS && (A.ntb + R >= 10);
    if constexpr (!INV) bfly_fwd<AR, WU && tw_uniform<A, K>()>(v[e], v[e | (1 << K)], t.w[s], t.wp[s], qk);
    else if constexpr (beta == 0) bfly_inv_last<AR, WU>(v[e], v[e | (1 << K)], tw.l1w, tw.l1wp, tw.sc0, tw.sc1, qk);
    else bfly_inv<AR, WU && tw_uniform<A, K>()>(v[e], v[e | (1 << K)], t.w[s], t.wp[s], qk);
  }
}

// exchange lane bit TB (4 or 5) with register bit RK: 16 v_permlane*_swap, no LDS
template <int TB, int RK>
__device__ __forceinline__ void lane_swap(u32 (&v)[E]) {
  static_assert(TB == 4 || TB == 5, "only lane bits 4 and 5 have swap instructions");
#pragma unroll
  for (int e = 0; e < E; ++e) {
    if (e & (1 << RK)) continue;
    const int f = e | (1 << RK);
    u32x2 r;
    if constexpr (TB == 5) r = __builtin_amdgcn_permlane32_swap(v[e], v[f], false, false);
    else r = __builtin_amdgcn_permlane16_swap(v[e], v[f], false, false);
    v[e] = r.x;
    v[f] = r.y;
  }
}
template <int TB, int RK>
__device__ __forceinline__ void lane_swap(u64 (&v)[E]) {
  static_assert(TB == 4 || TB == 5, "only lane bits 4 and 5 have swap instructions");
#pragma unroll
  for (int e = 0; e < E; ++e) {
    if (e & (1 << RK)) continue;
    const int f = e | (1 << RK);
    u32x2 lo, hi;
    if constexpr (TB == 5) {
      lo = __builtin_amdgcn_permlane32_swap((u32)v[e], (u32)v[f], false, false);
      hi = __builtin_amdgcn_permlane32_swap((u32)(v[e] >> 32), (u32)(v[f] >> 32), false, false);
    } else {
      lo = __builtin_amdgcn_permlane16_swap((u32)v[e], (u32)v[f], false, false);
      hi = __builtin_amdgcn_permlane16_swap((u32)(v[e] >> 32), (u32)(v[f] >> 32), false, false);
    }
    v[e] = ((u64)hi.x << 32) | lo.x;
    v[f] = ((u64)hi.y << 32) | lo.y;
  }
}

// LDS transpose A -> B, split so that independent work (twiddle fetches) sits between the
// halves.  WAVE-LOCAL transposes (both layouts keep a wave inside its own 1024-coefficient
// block) need no s_barrier at all: a wave's LDS instructions execute in order.  Only the one
// transpose per transform that crosses waves pays two workgroup barriers; the one protecting
// the previous reads comes FIRST, when every wave has long finished them.
template <Lay A, Lay B, bool CROSS_WAVE, typename V>
__device__ __forceinline__ void transpose_put(V (&v)[E], V* lds, int tau) {
#ifdef LOLHIP_ABL_NO_XPOSE
  return;
#endif
  if constexpr (!lay_eq(A, B)) {
    if constexpr (CROSS_WAVE) __syncthreads(); else __builtin_amdgcn_wave_barrier();
    V* wp = lds + lpad(xthr<A>(tau));
#pragma unroll
    for (int e = 0; e < E; ++e) wp[lpad(lay_tab<A>.xr[e])] = v[e];
  }
}
template <Lay A, Lay B, bool CROSS_WAVE, typename V>
__device__ __forceinline__ void transpose_get(V (&v)[E], V* lds, int tau) {
#ifdef LOLHIP_ABL_NO_XPOSE
  return;
#endif
  if constexpr (!lay_eq(A, B)) {
    if constexpr (CROSS_WAVE) __syncthreads(); else __builtin_amdgcn_wave_barrier();
    const V* rp = lds + lpad(xthr<B>(tau));
#pragma unroll
    for (int e = 0; e < E; ++e) v[e] = rp[lpad(lay_tab<B>.xr[e])];
  }
}

// layout with explicit register bits; thread bits listed lowest first
constexpr Lay lay_make(int L, int r0, int r1, int r2, int r3, const int* thr_bits) {
  Lay a{};
  a.ntb = L - R;
  a.reg[0] = r0; a.reg[1] = r1; a.reg[2] = r2; a.reg[3] = r3;
  for (int j = 0; j < a.ntb; ++j) a.thr[j] = thr_bits[j];
  return a;
}

// Compile-time schedule for n = 2^L.  LW = min(L,10) levels run WAVE-LOCALLY (a wave owns a
// block of 1024 consecutive coefficients, or 2^(10-L) whole polynomials when L < 10):
//   W0: registers = bits 0..3                      -> levels 1..4
//   W1: registers = bits 4..7, lane bits 4,5 = 8,9  -> levels 5..8, lane swaps -> levels 9, 10
//       (for L < 8 an overlapping window of the top four bits instead)
// and the remaining L-10 levels after ONE cross-wave transpose into
//   G:  registers = bits L-4..L-1                   -> levels 11..L
// G is also the coalesced global-memory layout (consecutive lanes = consecutive coefficients).
// W16 (single-modulus launches, 16-byte-aligned slabs): the two layouts that touch global memory
// keep position bit 0 on REGISTER bit 0, so a lane moves coefficients x, x+1 with one
// buffer_load/store_dwordx4 (16 B per lane, 1 KiB per wave-instruction: half the vector-memory
// instructions of the 8 B form).  G has a spare register bit for that up to L = 13 (level 10 is
// reached by a lane swap, so G's lowest register bit carries no level).
template <int L, bool W16 = false> struct Sched {
  static constexpr int NTB = L - R;
  static constexpr int LW = L < 10 ? L : 10;
  static constexpr bool HAS_G = L > 10;
  static constexpr Lay w0() { return lay_std(L, 0); }
  static constexpr bool W1_WINDOW = (LW > 4 && LW < 8);          // overlapping window, no swaps
  static constexpr bool HAS_W1 = LW > 4;
  static constexpr Lay w1() {
    if (W1_WINDOW) return lay_std(L, L - R);
    int thr[12] = {0, 1, 2, 3, 8, 9, 10, 11, 12, 13, 14, 15};    // t0-3 = bits 0-3, t4 = 8, t5 = 9, waves = 10+
    return lay_make(L, 4, 5, 6, 7, thr);
  }
  static constexpr int W1_K0 = W1_WINDOW ? (R - (LW - 4)) : 0;   // first register bit with work in W1
  static constexpr int NSWAP = LW >= 8 ? LW - 8 : 0;             // levels reached through lane swaps (0..2)
  static constexpr Lay w1a() { return lay_swap(w1(), 4, 3); }    // lane bit 4 <-> register bit 3: bit 8
  static constexpr Lay w1b() { return lay_swap(w1a(), 5, 2); }   // lane bit 5 <-> register bit 2: bit 9
  static constexpr Lay wave_end() { return NSWAP == 2 ? w1b() : NSWAP == 1 ? w1a() : HAS_W1 ? w1() : w0(); }
  static constexpr bool G16 = W16 && L > 10 && L <= 13;
  static constexpr Lay g() {
    if (!G16) return lay_std(L, L - R);
    int thr[12] = {1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12};       // bits 1..L-4 (only the first L-4 are used)
    return lay_make(L, 0, L - 3, L - 2, L - 1, thr);
  }
  static constexpr int G_K0 = R - (L - 10);                      // first register bit with work in G
  // coalesced load/store layout for powerful-basis data that keeps a wave inside its block
  static constexpr Lay io() {
    if (W16) {
      if (L < 10) { int thr[12] = {1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12}; return lay_make(L, 0, L - 3, L - 2, L - 1, thr); }
      int thr[12] = {1, 2, 3, 4, 5, 6, 10, 11, 12, 13, 14, 15};
      return lay_make(L, 0, 7, 8, 9, thr);
    }
    if (L < 10) return lay_std(L, L - R);
    int thr[12] = {0, 1, 2, 3, 4, 5, 10, 11, 12, 13, 14, 15};
    return lay_make(L, 6, 7, 8, 9, thr);
  }
  static constexpr Lay final_layout() { return HAS_G ? g() : wave_end(); }
};

template <int AR, bool INV, Lay A, int K0>
__device__ __forceinline__ void fetch4(LevelTwT<VT<AR>> (&t)[R], const TwCtxT<VT<AR>>& tw, int xt) {
  if constexpr (K0 <= 0) tw_fetch<INV, A, 0>(t[0], tw, xt);
  if constexpr (K0 <= 1) tw_fetch<INV, A, 1>(t[1], tw, xt);
  if constexpr (K0 <= 2) tw_fetch<INV, A, 2>(t[2], tw, xt);
  if constexpr (K0 <= 3) tw_fetch<INV, A, 3>(t[3], tw, xt);
}
// register-lean variant: fetch each level's twiddles right before the level (<= 32 VGPRs live)
template <int AR, bool INV, Lay A, int K>
__device__ __forceinline__ void level_jit(VT<AR> (&v)[E], const TwCtxT<VT<AR>>& tw, int xt, const QKT<AR>& qk) {
  using LevelTw = LevelTwT<VT<AR>>;
  if constexpr (tw_distinct(A, K) == 8 && sizeof(VT<AR>) == 8) {
    // all eight twiddles distinct (32 VGPRs): LOLHIP_LEVEL_PARTS parts keep the live set small
    [&]<int... PART>(std::integer_sequence<int, PART...>) {
      (([&] { LevelTw t; tw_fetch<INV, A, K, PART>(t, tw, xt); level<AR, INV, A, K, PART>(v, t, tw, qk); }()), ...);
    }(std::make_integer_sequence<int, LOLHIP_LEVEL_PARTS>{});
  } else {
    LevelTw t; tw_fetch<INV, A, K>(t, tw, xt); level<AR, INV, A, K>(v, t, tw, qk);
  }
}
template <int AR, Lay A, int K0>
__device__ __forceinline__ void levels4_jit(VT<AR> (&v)[E], const TwCtxT<VT<AR>>& tw, int xt, const QKT<AR>& qk) {
  if constexpr (K0 <= 0) level_jit<AR, false, A, 0>(v, tw, xt, qk);
  if constexpr (K0 <= 1) level_jit<AR, false, A, 1>(v, tw, xt, qk);
  if constexpr (K0 <= 2) level_jit<AR, false, A, 2>(v, tw, xt, qk);
  if constexpr (K0 <= 3) level_jit<AR, false, A, 3>(v, tw, xt, qk);
}
template <int AR, bool INV, Lay A, int K0>
__device__ __forceinline__ void levels4(VT<AR> (&v)[E], const LevelTwT<VT<AR>> (&t)[R], const TwCtxT<VT<AR>>& tw, const QKT<AR>& qk) {
  if constexpr (!INV) {
    if constexpr (K0 <= 0) level<AR, false, A, 0>(v, t[0], tw, qk);
    if constexpr (K0 <= 1) level<AR, false, A, 1>(v, t[1], tw, qk);
    if constexpr (K0 <= 2) level<AR, false, A, 2>(v, t[2], tw, qk);
    if constexpr (K0 <= 3) level<AR, false, A, 3>(v, t[3], tw, qk);
  } else {
    if constexpr (K0 <= 3) level<AR, true, A, 3>(v, t[3], tw, qk);
    if constexpr (K0 <= 2) level<AR, true, A, 2>(v, t[2], tw, qk);
    if constexpr (K0 <= 1) level<AR, true, A, 1>(v, t[1], tw, qk);
    if constexpr (K0 <= 0) level<AR, true, A, 0>(v, t[0], tw, qk);
  }
}

// forward transform; data arrives in registers in layout PREV, leaves in Sched<L>::final_layout().
// LEAN: the caller keeps 32 more VGPRs live (a-hat during b's transform in the fused poly-mul),
// so twiddles are fetched level by level instead of a pass ahead.
// An opaque copy of a value: addresses derived from the copy cannot be CSE'd with (and kept
// live since) an earlier pass's; recomputing a few address adds per pass is far cheaper than
// a dozen VGPRs held across all three transforms of the fused poly-mul.
__device__ __forceinline__ int fresh(int x) {
  asm volatile("" : "+v"(x));
  return x;
}

// The forward transform's vector-memory twiddles (the second lane-swap level and the levels after the
// cross-wave exchange: tables of 4 KiB and more, L2-served) may be supplied by the CALLER through a provider TOP
// with members l10, g[R] and wait_l10() / wait_g(), called right before the first use.  vmcnt retires in issue
// order: a kernel that keeps an LDS-DMA in flight under a transform (pow2_pipe.hip) fetches these BEFORE it
// issues the DMA, or the first such twiddle waits for the whole DMA to land.
struct NoTop {};
template <int AR, int L, Lay PREV, int SB = 0, bool LEAN = false, bool W16 = false, typename TOP = NoTop>
__device__ __forceinline__ void fwd_transform(VT<AR> (&v)[E], VT<AR>* lds, const TwCtxT<VT<AR>>& tw, int tau_in, const QKT<AR>& qk,
                                              TOP* pre = nullptr) {
  constexpr bool PRE = !std::is_same_v<TOP, NoTop>;
  using LevelTw = LevelTwT<VT<AR>>;
  using S = Sched<L, W16>;
  {   // W0: levels 1..4
    constexpr Lay A = S::w0();
    const int tau = fresh(tau_in);
    // the previous transform's cross-wave reads may still be in flight in other waves
    if constexpr (S::HAS_G && !lay_eq(PREV, A)) __syncthreads();
    transpose_put<PREV, A, false>(v, lds, tau);
    if constexpr (LEAN) {
      transpose_get<PREV, A, false>(v, lds, tau);
      levels4_jit<AR, A, 0>(v, tw, xthr<A>(tau), qk);
    } else {
      LevelTw t[R];
      fetch4<AR, false, A, 0>(t, tw, xthr<A>(tau));
      transpose_get<PREV, A, false>(v, lds, tau);
      LH_STAMP(SB + 2);
      levels4<AR, false, A, 0>(v, t, tw, qk);
      LH_STAMP(SB + 3);
    }
  }
  if constexpr (S::HAS_W1) {   // W1: levels 5..8 (or the window), then lane swaps for 9, 10
    constexpr Lay A = S::w1();
    const int tau = fresh(tau_in);
    if constexpr (LEAN) {
      transpose_put<S::w0(), A, false>(v, lds, tau);
      transpose_get<S::w0(), A, false>(v, lds, tau);
      levels4_jit<AR, A, S::W1_K0>(v, tw, xthr<A>(tau), qk);
      if constexpr (S::NSWAP >= 1) { lane_swap<4, 3>(v); level_jit<AR, false, S::w1a(), 3>(v, tw, xthr<S::w1a()>(tau), qk); }
      if constexpr (S::NSWAP >= 2) {
        lane_swap<5, 2>(v);
        if constexpr (PRE) { pre->wait_l10(); level<AR, false, S::w1b(), 2>(v, pre->l10, tw, qk); }
        else level_jit<AR, false, S::w1b(), 2>(v, tw, xthr<S::w1b()>(tau), qk);
      }
    } else {
      LevelTw t[R];
      transpose_put<S::w0(), A, false>(v, lds, tau);
      fetch4<AR, false, A, S::W1_K0>(t, tw, xthr<A>(tau));
      transpose_get<S::w0(), A, false>(v, lds, tau);
      LH_STAMP(SB + 4);
      // levels 5..8, with the twiddles of the two lane-swap levels fetched as registers free up
      if constexpr (S::W1_K0 <= 0) level<AR, false, A, 0>(v, t[0], tw, qk);
      if constexpr (S::W1_K0 <= 1) level<AR, false, A, 1>(v, t[1], tw, qk);
      if constexpr (S::W1_K0 <= 2) level<AR, false, A, 2>(v, t[2], tw, qk);
      LevelTw ua, ub;
      if constexpr (S::NSWAP >= 1) tw_fetch<false, S::w1a(), 3>(ua, tw, xthr<S::w1a()>(tau));
      level<AR, false, A, 3>(v, t[3], tw, qk);
      LH_STAMP(SB + 5);
      if constexpr (S::NSWAP >= 2 && !PRE) tw_fetch<false, S::w1b(), 2>(ub, tw, xthr<S::w1b()>(tau));
      if constexpr (S::NSWAP >= 1) { lane_swap<4, 3>(v); level<AR, false, S::w1a(), 3>(v, ua, tw, qk); }
      if constexpr (S::NSWAP >= 2) {
        lane_swap<5, 2>(v);
        if constexpr (PRE) { pre->wait_l10(); level<AR, false, S::w1b(), 2>(v, pre->l10, tw, qk); }
        else level<AR, false, S::w1b(), 2>(v, ub, tw, qk);
      }
      LH_STAMP(SB + 6);
    }
  }
  if constexpr (S::HAS_G) {    // the one cross-wave exchange, then levels 11..L
    constexpr Lay A = S::g();
    const int tau = fresh(tau_in);
    transpose_put<S::wave_end(), A, false>(v, lds, tau);   // writes stay inside the wave's own block
    if constexpr (PRE) {
      transpose_get<S::wave_end(), A, true>(v, lds, tau);  // barrier, then read across blocks
      LH_STAMP(SB + 7);
      pre->wait_g();
      levels4<AR, false, A, S::G_K0>(v, pre->g, tw, qk);
      LH_STAMP(SB + 8);
    } else if constexpr (LEAN) {
      transpose_get<S::wave_end(), A, true>(v, lds, tau);  // barrier, then read across blocks
      levels4_jit<AR, A, S::G_K0>(v, tw, xthr<A>(tau), qk);
    } else {
      LevelTw t[R];
      fetch4<AR, false, A, S::G_K0>(t, tw, xthr<A>(tau));
      transpose_get<S::wave_end(), A, true>(v, lds, tau);
      LH_STAMP(SB + 7);
      levels4<AR, false, A, S::G_K0>(v, t, tw, qk);
      LH_STAMP(SB + 8);
    }
  }
}

// inverse transform; data arrives in Sched<L>::final_layout(), leaves in layout NEXT
template <int AR, int L, Lay NEXT, bool W16 = false>
__device__ __forceinline__ void inv_transform(VT<AR> (&v)[E], VT<AR>* lds, const TwCtxT<VT<AR>>& tw, int tau_in, const QKT<AR>& qk) {
  using LevelTw = LevelTwT<VT<AR>>;
  using S = Sched<L, W16>;
  if constexpr (S::HAS_G) {
    constexpr Lay A = S::g();
    const int tau = fresh(tau_in);
    LevelTw t[R];
    fetch4<AR, true, A, S::G_K0>(t, tw, xthr<A>(tau));
    levels4<AR, true, A, S::G_K0>(v, t, tw, qk);
    transpose_put<A, S::wave_end(), true>(v, lds, tau);    // barrier (earlier readers), write across blocks
  }
  if constexpr (S::HAS_W1) {
    constexpr Lay A = S::w1();
    const int tau = fresh(tau_in);
    const int xt = xthr<A>(tau);
    LevelTw ua, ub, t[R];
    if constexpr (S::NSWAP >= 2) tw_fetch<true, S::w1b(), 2>(ub, tw, xthr<S::w1b()>(tau));
    if constexpr (S::NSWAP >= 1) tw_fetch<true, S::w1a(), 3>(ua, tw, xthr<S::w1a()>(tau));
    if constexpr (S::NSWAP == 0) fetch4<AR, true, A, S::W1_K0>(t, tw, xt);
    if constexpr (S::HAS_G) transpose_get<S::g(), S::wave_end(), true>(v, lds, tau);   // barrier, read own block
    if constexpr (S::NSWAP >= 2) { level<AR, true, S::w1b(), 2>(v, ub, tw, qk); lane_swap<5, 2>(v); }
    if constexpr (S::NSWAP >= 1) {
      tw_fetch<true, A, 3>(t[3], tw, xt);
      level<AR, true, S::w1a(), 3>(v, ua, tw, qk);
      lane_swap<4, 3>(v);
      tw_fetch<true, A, 2>(t[2], tw, xt);
      tw_fetch<true, A, 1>(t[1], tw, xt);
      tw_fetch<true, A, 0>(t[0], tw, xt);
    }
    levels4<AR, true, A, S::W1_K0>(v, t, tw, qk);
    transpose_put<A, S::w0(), false>(v, lds, tau);
  }
  {
    constexpr Lay A = S::w0();
    const int tau = fresh(tau_in);
    LevelTw t[R];
    fetch4<AR, true, A, 0>(t, tw, xthr<A>(tau));
    if constexpr (S::HAS_W1) transpose_get<S::w1(), A, false>(v, lds, tau);
    levels4<AR, true, A, 0>(v, t, tw, qk);
    transpose_put<A, NEXT, false>(v, lds, tau);
    transpose_get<A, NEXT, false>(v, lds, tau);
  }
}

constexpr int pow2_threads(int L) { return (1 << (L - R)) >= 256 ? (1 << (L - R)) : 256; }

// MODE 0: crt in place, 1: crtInv in place, 2: c = crtInv(crt(a) * crt(b))
// T1: the launch has a single modulus (T = 1) and 16-byte-aligned slabs: t = 0 for every lane even
// when a wave holds several short polynomials — the per-modulus constants stay in SGPRs — and
// global memory moves 16 bytes per lane (Sched<L, true>)
template <int L, int MODE, int AR, bool T1 = false>
__global__ void __launch_bounds__(pow2_threads(L), AR >= 2 ? 8 : 4)
k_pow2(i64* y, const i64* a_in, const i64* b_in, i64 B, int T,
       const VT<AR>* __restrict__ tw_fwd, const VT<AR>* __restrict__ tw_inv, const VT<AR>* __restrict__ scale,
       const ModCtx* __restrict__ mod, int xcd_map) {
  using S = Sched<L, T1>;
  using V = VT<AR>;
  constexpr int n = 1 << L;
  constexpr int NT = 1 << (L - R);                  // threads per polynomial
  constexpr int PPW = NT >= 256 ? 1 : 256 / NT;     // polynomials per workgroup
  constexpr int LDSW = n + n / 16 + twl_words(n);   // padded coefficients + twiddle copy, per polynomial (V words)
  constexpr u32 TWB = 2 * sizeof(V);                // bytes per twiddle entry
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  V* lds = reinterpret_cast<V*>(smem) + (threadIdx.x / NT) * LDSW;
  V* lds_tw = lds + n + n / 16;
  const int tau = threadIdx.x % NT;
  if constexpr (T1) T = 1;

  // work item -> (b, t); with xcd_map the T components of one polynomial land on
  // workgroups that share an XCD (equal blockIdx % 8) so its cache lines are
  // fetched from HBM once.  Placement only affects speed.
  const i64 item0 = (i64)blockIdx.x * PPW;             // wave-uniform
  const int slot = (PPW == 1) ? 0 : (int)(threadIdx.x / NT);
  const i64 item = item0 + slot;
  i64 b, b0; int t;
  if (xcd_map) { i64 g = item / (8 * (i64)T); int r = (int)(item % (8 * T)); b = g * 8 + (r & 7); t = r >> 3; b0 = b; }
  else { b = item / T; t = (int)(item % T); b0 = item0 / T; }
  if constexpr (T1) { b = item; t = 0; b0 = item0; }

  if constexpr (NT >= 64) {           // a wave never straddles two polynomials: make that provable to hipcc
    t = __builtin_amdgcn_readfirstlane(t);
    b = (i64)(((u64)(u32)__builtin_amdgcn_readfirstlane((int)(b >> 32)) << 32) | (u32)__builtin_amdgcn_readfirstlane((int)b));
  }
  b0 = (i64)(((u64)(u32)__builtin_amdgcn_readfirstlane((int)(b0 >> 32)) << 32) | (u32)__builtin_amdgcn_readfirstlane((int)b0));
  const QKT<AR> qk(mod[t], std::bool_constant<(NT >= 64 || T1)>{});
  // Buffer descriptors (wave-uniform): data windows start at the workgroup's first polynomial
  // and end at the end of the batch, so tail lanes of a packed launch read zeros and their
  // stores are dropped by the hardware range check.
  const u64 win = (u64)(B - b0) * n * T * 8;
  const u32 wbytes = win > 0xFFFFFFFFull ? 0xFFFFFFFFu : (u32)win;
  const size_t wbase = (size_t)b0 * n * T;
  const rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)(y + wbase), 0, wbytes, 0x00020000);
  const rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)(a_in + (MODE == 2 ? wbase : 0)), 0, MODE == 2 ? wbytes : 0, 0x00020000);
  const rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)(b_in + (MODE == 2 ? wbase : 0)), 0, MODE == 2 ? wbytes : 0, 0x00020000);
  TwCtxT<V> tw;
  tw.fwd = __builtin_amdgcn_make_buffer_rsrc((void*)tw_fwd, 0, (u32)T * n * TWB, 0x00020000);
  tw.inv = __builtin_amdgcn_make_buffer_rsrc((void*)tw_inv, 0, (u32)T * n * TWB, 0x00020000);
  tw.comp = (u32)t * (u32)n * TWB;
  tw.pf = tw_fwd + (size_t)t * n * 2;
  tw.pi = tw_inv + (size_t)t * n * 2;
  tw.lds_tw = lds_tw;
  // levels 5..9 read their twiddles from LDS; (re)filled before the transform direction changes.
  // Visibility: for L > 10 a workgroup barrier follows before first use; for L <= 10 the
  // polynomial's own wave does both the fill and the reads.
  if constexpr (L >= TWL_MIN_L) tw_fill_lds<NT>(lds_tw, (MODE == 1) ? tw.inv : tw.fwd, tw.comp, n, tau);
  constexpr int SC = (MODE == 2 && AR == 1) ? 4 : 0;       // the 2^64-scaled pairs: see pmul
  tw.sc0 = scale[(size_t)t * 8 + SC];
  tw.sc1 = scale[(size_t)t * 8 + SC + 1];
  tw.l1w = scale[(size_t)t * 8 + SC + 2];
  tw.l1wp = scale[(size_t)t * 8 + SC + 3];

  constexpr Lay LIO = S::io();              // global I/O of powerful-basis data
  constexpr Lay LFIN = S::final_layout();   // where the forward transform leaves the CRT coefficients
  const u32 uT8 = (u32)T * 8u;
  const u32 pofs = ((u32)(b - b0) * (u32)n * (u32)T + (u32)t) * 8u;       // this polynomial inside the window
  const u32 off_io = pofs + (u32)xthr<LIO>(tau) * uT8;
  const u32 off_fin = pofs + (u32)xthr<LFIN>(tau) * uT8;

  V v[E];
  LH_STAMP(0);
#ifdef LOLHIP_STAMPS
  if (g_stamp_buf && (threadIdx.x & 63) == 0) {
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    g_stamp_buf[((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 32 + 30] = ((unsigned long long)xcc << 32) | hwid;
  }
#endif
  if constexpr (MODE == 0 || MODE == 2) {
    const rsrc_t src = (MODE == 2) ? ra : ry;
    load_poly<LIO, T1>(src, off_io, uT8, [&](int e, u64 x) { v[e] = from_i64_fwd<AR>((i64)x, qk); });
    LH_STAMP(1);
    fwd_transform<AR, L, LIO, 0, false, T1>(v, lds, tw, tau, qk);
    if constexpr (MODE == 0) LH_STAMP(20);
  }
  if constexpr (MODE == 2) {
    // a-hat stays in registers (canonical) while b is transformed with the register-lean
    // twiddle schedule; nothing is parked in HBM, and c may alias a and/or b freely because
    // both operands are fully read before the first store to c.
    V va[E];
#pragma unroll
    for (int e = 0; e < E; ++e) va[e] = park_fwd<AR>(v[e], qk);
    LH_STAMP(9);
    const bool square = (a_in == b_in);
    if (!square) {
      // (issuing these with a's loads and holding the 32 raw registers across a's transform
      // was measured: no gain at q ~ 2^30, 7% slower in class 4 — 6 instead of 8 waves/SIMD)
      u64 raw[E];
      load_poly<LIO, T1>(rb, off_io, uT8, [&](int e, u64 x) { raw[e] = x; });
      // keep the 16 loads back to back: at this register pressure the scheduler otherwise
      // sinks each load to its use and the wave pays 16 serial HBM round trips
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int e = 0; e < E; ++e) v[e] = from_i64_fwd<AR>((i64)raw[e], qk);
      LH_STAMP(11);
      fwd_transform<AR, L, LIO, 10, true, T1>(v, lds, tw, tau, qk);
    }
    LH_STAMP(19);
    const ModCtx mc = mod[t];     // re-read here: keeping it live across the transforms costs registers
#pragma unroll
    for (int e = 0; e < E; ++e) v[e] = pmul<AR>(va[e], (AR < 2 && square) ? va[e] : v[e], mc, qk);   // squaring: v still holds a-hat (lazy)
    LH_STAMP(22);
  }
  if constexpr (MODE == 2 && L >= TWL_MIN_L) {
    // every wave is past its last forward use of the LDS twiddles (the cross-wave barrier of
    // the final forward stage, or program order inside a wave): switch the copy to the inverse table
    if constexpr (L > 10) __syncthreads();
    tw_fill_lds<NT>(lds_tw, tw.inv, tw.comp, n, tau);
  }
  if constexpr (MODE == 1) {
    load_poly<LFIN, T1>(ry, off_fin, uT8, [&](int e, u64 x) { v[e] = from_i64_inv<AR>((i64)x, qk); });
  }
  if constexpr (MODE == 0) {
    store_poly<LFIN, T1>(ry, off_fin, uT8, [&](int e) { return (u64)canon_fwd<AR>(v[e], qk); });
    LH_STAMP(21);
  } else {
    LH_STAMP(23);
    inv_transform<AR, L, LIO, T1>(v, lds, tw, tau, qk);
    LH_STAMP(24);
    store_poly<LIO, T1>(ry, off_io, uT8, [&](int e) { return (u64)canon_inv<AR>(v[e], qk); });
    LH_STAMP(25);
  }
}

// Per-device launch state of one kernel instantiation.  hipFuncSetAttribute is per DEVICE: a
// process that drives several GPUs has to repeat it on each of them (a per-process flag made the
// second GPU's first > 64 KiB-LDS launch fail).  state 0 = not set up, 2 = ready.
constexpr int MAX_DEV = 64;
struct KernelDev { std::atomic<int> state{0}; };
inline std::mutex& kernel_dev_mutex() { static std::mutex m; return m; }
template <typename Setup>
static hipError_t kernel_dev_setup(KernelDev (&tab)[MAX_DEV], Setup&& setup) {
  int dev = -1;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev < 0 || dev >= MAX_DEV) return hipErrorInvalidDevice;
  KernelDev& d = tab[dev];
  if (d.state.load(std::memory_order_acquire) != 2) {
    std::lock_guard<std::mutex> g(kernel_dev_mutex());
    if (d.state.load(std::memory_order_relaxed) != 2) {
      e = setup();
      if (e != hipSuccess) return e;
      d.state.store(2, std::memory_order_release);
    }
  }
  return hipSuccess;
}

template <int L, int MODE, int AR, bool T1>
static hipError_t launch_pow2_L(const Pow2Launch& a) {
  constexpr int n = 1 << L;
  constexpr int NT = 1 << (L - R);
  constexpr int PPW = NT >= 256 ? 1 : 256 / NT;
  constexpr int LDSW = n + n / 16 + twl_words(n);
  const size_t lds_bytes = (size_t)PPW * LDSW * sizeof(VT<AR>);
  const i64 items = a.B * a.T;
  const int xcd_map = (a.T > 1 && PPW == 1 && a.B % 8 == 0) ? 1 : 0;
  const i64 grid = (items + PPW - 1) / PPW;
  if (grid == 0) return hipSuccess;
  if (lds_bytes > 64 * 1024) {
    static KernelDev tab[MAX_DEV];
    hipError_t e = kernel_dev_setup(tab, [&]() -> hipError_t {
      return hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pow2<L, MODE, AR, T1>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    });
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL((k_pow2<L, MODE, AR, T1>), dim3((unsigned)grid), dim3(NT * PPW), lds_bytes, a.stream,
                     a.y, a.a, a.b, a.B, a.T, static_cast<const VT<AR>*>(a.tw_fwd), static_cast<const VT<AR>*>(a.tw_inv),
                     static_cast<const VT<AR>*>(a.scale), a.mod, xcd_map);
  return hipGetLastError();
}

template <int MODE, int AR, bool T1>
static hipError_t launch_pow2_mode(const Pow2Launch& a) {
  switch (a.L) {
    case 4: return launch_pow2_L<4, MODE, AR, T1>(a);
    case 5: return launch_pow2_L<5, MODE, AR, T1>(a);
    case 6: return launch_pow2_L<6, MODE, AR, T1>(a);
    case 7: return launch_pow2_L<7, MODE, AR, T1>(a);
    case 8: return launch_pow2_L<8, MODE, AR, T1>(a);
    case 9: return launch_pow2_L<9, MODE, AR, T1>(a);
    case 10: return launch_pow2_L<10, MODE, AR, T1>(a);
    case 11: return launch_pow2_L<11, MODE, AR, T1>(a);
    case 12: return launch_pow2_L<12, MODE, AR, T1>(a);
    case 13: return launch_pow2_L<13, MODE, AR, T1>(a);
    case 14: return launch_pow2_L<14, MODE, AR, T1>(a);
    default: return hipErrorInvalidValue;
  }
}

// T1: single-modulus launches whose slabs are 16-byte aligned (checked by launch_pow2, kernels.hip)
template <int AR, bool T1>
hipError_t launch_pow2_ar(const Pow2Launch& a, int mode) {
  switch (mode) {
    case 0: return launch_pow2_mode<0, AR, T1>(a);
    case 1: return launch_pow2_mode<1, AR, T1>(a);
    case 2: return launch_pow2_mode<2, AR, T1>(a);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace lolhip
