// lol_amd/csrc/kernels.hip — hand-written gfx950 kernels for the Lol Tensor hot path.
//
// One polynomial (one RNS component of it) is owned by one workgroup, staged
// once into registers/LDS, taken through EVERY stage of the transform there and
// written back once: HBM sees 8 bytes in + 8 bytes out per coefficient, where the
// reference makes one full sweep over memory per stage per prime power
// (crt.cpp:459-538, tensor.h:76-95).
//
//   k_pow2<L,MODE>   m = 2^(L+1): negacyclic NTT (CRT), inverse, fused poly-mul
//   k_generic        any m: interpreter of the plan's stage program (CRT/CRT^-1 via
//                    dense p-point stages, L/L^-1, mulG/divG in Pow and Dec bases)
//   k_pointwise_mul  mulRq (mul.cpp:14-30) and mulGCRT/divGCRT (CPP.hs:230-231)
//   k_gather / k_twace_crt   twace*/embed* (Extension.hs:54-129)
//
// Layout at the boundary is the reference's: y[(b*n + j)*T + t] (tensor.h:69).
#include <hip/hip_runtime.h>

#include <cstdlib>
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
__device__ unsigned long long* g_stamp_buf = nullptr;
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
//  AR = 1 (every q_t < 2^61): 64-bit residues, Shoup products with the 9-multiply approximate
//         quotient (shoup_acc, result in [0,4q)); forward values in [0,8q), inverse in [0,4q).
//  AR = 0 (2^61 <= q_t < 2^62): 10-multiply exact quotient, ranges [0,4q) / [0,2q).
template <int AR> using VT = std::conditional_t<AR == 2, u32, u64>;

// Per-modulus constants of the lazy butterflies (wave-uniform, live in SGPRs).
struct QK {
  u64 q, nq, q2, nq2, q4, nq4;
  __device__ __forceinline__ explicit QK(u64 q_) : q(q_), nq(0 - q_), q2(2 * q_), nq2(0 - 2 * q_), q4(4 * q_), nq4(0 - 4 * q_) {}
};
struct QK32 {
  u32 q, q2;
  __device__ __forceinline__ explicit QK32(u64 q_) : q((u32)q_), q2(2 * (u32)q_) {}
};
template <int AR> using QKT = std::conditional_t<AR == 2, QK32, QK>;

__device__ __forceinline__ u32 csub32(u32 x, u32 m) { return min(x, x - m); }        // x < 2m
// w*y mod q in [0,2q) for any 32-bit y
__device__ __forceinline__ u32 shoup32(u32 y, u32 w, u32 wp, u32 q) { return w * y - __umulhi(wp, y) * q; }

// forward (Cooley-Tukey) butterfly:  X' = X + w*Y,  Y' = X - w*Y
template <int AR>
__device__ __forceinline__ void bfly_fwd(VT<AR>& X, VT<AR>& Y, VT<AR> w, VT<AR> wp, const QKT<AR>& k) {
  if constexpr (AR == 2) {
    const u32 x = csub32(X, k.q2);                    // [0,4q) -> [0,2q)
    const u32 t = shoup32(Y, w, wp, k.q);
    X = x + t;
    Y = x - t + k.q2;
  } else if constexpr (AR == 1) {
    const u64 x = csubn(X, k.nq4);                    // [0,8q) -> [0,4q)
    const u64 xn = shoup_acc(Y, w, wp, k.nq, x);      // x + t, t in [0,4q)
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
template <int AR>
__device__ __forceinline__ void bfly_inv(VT<AR>& X, VT<AR>& Y, VT<AR> w, VT<AR> wp, const QKT<AR>& k) {
  if constexpr (AR == 2) {
    const u32 s = X + Y;
    const u32 d = X - Y + k.q2;
    X = csub32(s, k.q2);
    Y = shoup32(d, w, wp, k.q);
  } else if constexpr (AR == 1) {
    const u64 s = add64(X, Y);                        // [0,8q)
    const u64 d = add64u(X, k.q4) - Y;                 // (0,8q)
    X = csubn(s, k.nq4);
    Y = shoup_acc(d, w, wp, k.nq, 0);
  } else {
    const u64 s = X + Y;
    const u64 d = X - Y + k.q2;
    X = csub(s, k.q2);
    Y = shoup_lazy(d, w, wp, k.q);
  }
}
// last inverse level: both outputs additionally scaled by mhat^-1 (crt.cpp:573-579).
// (s0,s1) = Shoup pair of mhat^-1; (w, wp) = Shoup pair of psi_2^-1 * mhat^-1.
template <int AR>
__device__ __forceinline__ void bfly_inv_last(VT<AR>& X, VT<AR>& Y, VT<AR> w, VT<AR> wp, VT<AR> s0, VT<AR> s1, const QKT<AR>& k) {
  if constexpr (AR == 2) {
    const u32 s = X + Y;
    const u32 d = X - Y + k.q2;
    X = shoup32(s, s0, s1, k.q);
    Y = shoup32(d, w, wp, k.q);
  } else if constexpr (AR == 1) {
    const u64 s = add64(X, Y);
    const u64 d = add64u(X, k.q4) - Y;
    X = shoup_acc(s, s0, s1, k.nq, 0);
    Y = shoup_acc(d, w, wp, k.nq, 0);
  } else {
    const u64 s = X + Y;
    const u64 d = X - Y + k.q2;
    X = shoup_lazy(s, s0, s1, k.q);
    Y = shoup_lazy(d, w, wp, k.q);
  }
}
template <int AR> __device__ __forceinline__ VT<AR> canon_fwd(VT<AR> v, const QKT<AR>& k) {
  if constexpr (AR == 2) return csub32(csub32(v, k.q2), k.q);
  else {
    if constexpr (AR == 1) v = csubn(v, k.nq4);
    return csubn(csubn(v, k.nq2), k.nq);
  }
}
template <int AR> __device__ __forceinline__ VT<AR> canon_inv(VT<AR> v, const QKT<AR>& k) {
  if constexpr (AR == 2) return csub32(v, k.q);
  else {
    if constexpr (AR == 1) v = csubn(v, k.nq2);
    return csubn(v, k.nq);
  }
}
// reference-style input in (-q, q) -> [0, q)
template <int AR> __device__ __forceinline__ VT<AR> from_i64(i64 x, const QKT<AR>& k) {
  if constexpr (AR == 2) return (u32)x + (k.q & (u32)(x >> 63));
  else return canon_in(x, k.q);
}
// pointwise product of a canonical a-hat and a lazy b-hat (forward range), any range the
// inverse transform accepts
template <int AR> __device__ __forceinline__ VT<AR> pmul(VT<AR> a, VT<AR> b, const ModCtx& mc, const QKT<AR>& k) {
  if constexpr (AR == 2) {
    const u64 x = (u64)a * b;                          // < q * 4q < 2^62
    const u64 Q = __umul64hi(x, mc.mu);                // floor(x/q) or one less
    return csub32((u32)(x - Q * mc.q), k.q);           // [0,2q) -> [0,q)
  } else if constexpr (AR == 1) {
    return mulmod(a, b, mc);                           // a < q, b < 8q: a*b < q * 2^64
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
  V sc0, sc1;           // Shoup pair of mhat^-1
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
        t.w[ord] = sp[2 * cidx]; t.wp[ord] = sp[2 * cidx + 1];
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
    if constexpr (!INV) bfly_fwd<AR>(v[e], v[e | (1 << K)], t.w[s], t.wp[s], qk);
    else if constexpr (beta == 0) bfly_inv_last<AR>(v[e], v[e | (1 << K)], t.w[s], t.wp[s], tw.sc0, tw.sc1, qk);
    else bfly_inv<AR>(v[e], v[e | (1 << K)], t.w[s], t.wp[s], qk);
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
template <int L> struct Sched {
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
  static constexpr Lay g() { return lay_std(L, L - R); }
  static constexpr int G_K0 = R - (L - 10);                      // first register bit with work in G
  // coalesced load/store layout for powerful-basis data that keeps a wave inside its block
  static constexpr Lay io() {
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

template <int AR, int L, Lay PREV, int SB = 0, bool LEAN = false>
__device__ __forceinline__ void fwd_transform(VT<AR> (&v)[E], VT<AR>* lds, const TwCtxT<VT<AR>>& tw, int tau_in, const QKT<AR>& qk) {
  using LevelTw = LevelTwT<VT<AR>>;
  using S = Sched<L>;
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
      if constexpr (S::NSWAP >= 2) { lane_swap<5, 2>(v); level_jit<AR, false, S::w1b(), 2>(v, tw, xthr<S::w1b()>(tau), qk); }
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
      if constexpr (S::NSWAP >= 2) tw_fetch<false, S::w1b(), 2>(ub, tw, xthr<S::w1b()>(tau));
      if constexpr (S::NSWAP >= 1) { lane_swap<4, 3>(v); level<AR, false, S::w1a(), 3>(v, ua, tw, qk); }
      if constexpr (S::NSWAP >= 2) { lane_swap<5, 2>(v); level<AR, false, S::w1b(), 2>(v, ub, tw, qk); }
      LH_STAMP(SB + 6);
    }
  }
  if constexpr (S::HAS_G) {    // the one cross-wave exchange, then levels 11..L
    constexpr Lay A = S::g();
    const int tau = fresh(tau_in);
    transpose_put<S::wave_end(), A, false>(v, lds, tau);   // writes stay inside the wave's own block
    if constexpr (LEAN) {
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
template <int AR, int L, Lay NEXT>
__device__ __forceinline__ void inv_transform(VT<AR> (&v)[E], VT<AR>* lds, const TwCtxT<VT<AR>>& tw, int tau_in, const QKT<AR>& qk) {
  using LevelTw = LevelTwT<VT<AR>>;
  using S = Sched<L>;
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
// TU ("T uniform"): the launch has a single modulus, so t = 0 for every lane even when a wave
// holds several short polynomials — the per-modulus constants stay in SGPRs (instantiated for
// n <= 512 only; longer polynomials own whole waves and are uniform anyway)
template <int L, int MODE, int AR, bool TU = false>
__global__ void __launch_bounds__(pow2_threads(L), AR == 2 ? 8 : 4)
k_pow2(i64* y, const i64* a_in, const i64* b_in, i64 B, int T,
       const VT<AR>* __restrict__ tw_fwd, const VT<AR>* __restrict__ tw_inv, const VT<AR>* __restrict__ scale,
       const ModCtx* __restrict__ mod, int xcd_map) {
  using S = Sched<L>;
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

  // work item -> (b, t); with xcd_map the T components of one polynomial land on
  // workgroups that share an XCD (equal blockIdx % 8) so its cache lines are
  // fetched from HBM once.  Placement only affects speed.
  const i64 item0 = (i64)blockIdx.x * PPW;             // wave-uniform
  const int slot = (PPW == 1) ? 0 : (int)(threadIdx.x / NT);
  const i64 item = item0 + slot;
  i64 b, b0; int t;
  if (xcd_map) { i64 g = item / (8 * (i64)T); int r = (int)(item % (8 * T)); b = g * 8 + (r & 7); t = r >> 3; b0 = b; }
  else { b = item / T; t = (int)(item % T); b0 = item0 / T; }
  if constexpr (TU) { b = item; t = 0; b0 = item0; }

  if constexpr (NT >= 64) {           // a wave never straddles two polynomials: make that provable to hipcc
    t = __builtin_amdgcn_readfirstlane(t);
    b = (i64)(((u64)(u32)__builtin_amdgcn_readfirstlane((int)(b >> 32)) << 32) | (u32)__builtin_amdgcn_readfirstlane((int)b));
  }
  b0 = (i64)(((u64)(u32)__builtin_amdgcn_readfirstlane((int)(b0 >> 32)) << 32) | (u32)__builtin_amdgcn_readfirstlane((int)b0));
  const QKT<AR> qk(mod[t].q);
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
  tw.sc0 = scale[(size_t)t * 2];
  tw.sc1 = scale[(size_t)t * 2 + 1];

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
#pragma unroll
    for (int e = 0; e < E; ++e) v[e] = from_i64<AR>((i64)load_u64(src, off_io, (u32)lay_tab<LIO>.xr[e] * uT8), qk);
    LH_STAMP(1);
    fwd_transform<AR, L, LIO>(v, lds, tw, tau, qk);
    if constexpr (MODE == 0) LH_STAMP(20);
  }
  if constexpr (MODE == 2) {
    // a-hat stays in registers (canonical) while b is transformed with the register-lean
    // twiddle schedule; nothing is parked in HBM, and c may alias a and/or b freely because
    // both operands are fully read before the first store to c.
    V va[E];
#pragma unroll
    for (int e = 0; e < E; ++e) va[e] = canon_fwd<AR>(v[e], qk);
    LH_STAMP(9);
    const bool square = (a_in == b_in);
    if (!square) {
      u64 raw[E];
#pragma unroll
      for (int e = 0; e < E; ++e) raw[e] = load_u64(rb, off_io, (u32)lay_tab<LIO>.xr[e] * uT8);
      // keep the 16 loads back to back: at this register pressure the scheduler otherwise
      // sinks each load to its use and the wave pays 16 serial HBM round trips
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int e = 0; e < E; ++e) v[e] = from_i64<AR>((i64)raw[e], qk);
      LH_STAMP(11);
      fwd_transform<AR, L, LIO, 10, true>(v, lds, tw, tau, qk);
    }
    LH_STAMP(19);
    const ModCtx mc = mod[t];     // re-read here: keeping it live across the transforms costs registers
#pragma unroll
    for (int e = 0; e < E; ++e) v[e] = pmul<AR>(va[e], square ? va[e] : v[e], mc, qk);
    LH_STAMP(22);
  }
  if constexpr (MODE == 2 && L >= TWL_MIN_L) {
    // every wave is past its last forward use of the LDS twiddles (the cross-wave barrier of
    // the final forward stage, or program order inside a wave): switch the copy to the inverse table
    if constexpr (L > 10) __syncthreads();
    tw_fill_lds<NT>(lds_tw, tw.inv, tw.comp, n, tau);
  }
  if constexpr (MODE == 1) {
#pragma unroll
    for (int e = 0; e < E; ++e) v[e] = from_i64<AR>((i64)load_u64(ry, off_fin, (u32)lay_tab<LFIN>.xr[e] * uT8), qk);
  }
  if constexpr (MODE == 0) {
#pragma unroll
    for (int e = 0; e < E; ++e) store_u64(ry, off_fin, (u32)lay_tab<LFIN>.xr[e] * uT8, (u64)canon_fwd<AR>(v[e], qk));
    LH_STAMP(21);
  } else {
    LH_STAMP(23);
    inv_transform<AR, L, LIO>(v, lds, tw, tau, qk);
    LH_STAMP(24);
#pragma unroll
    for (int e = 0; e < E; ++e) store_u64(ry, off_io, (u32)lay_tab<LIO>.xr[e] * uT8, (u64)canon_inv<AR>(v[e], qk));
    LH_STAMP(25);
  }
}

template <int L, int MODE, int AR>
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
  static bool attr_set = false;
  if (!attr_set && lds_bytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pow2<L, MODE, AR>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  if constexpr (NT < 64 && AR != 2) {
    if (a.T == 1) {
      hipLaunchKernelGGL((k_pow2<L, MODE, AR, true>), dim3((unsigned)grid), dim3(NT * PPW), lds_bytes, a.stream,
                         a.y, a.a, a.b, a.B, a.T, static_cast<const VT<AR>*>(a.tw_fwd), static_cast<const VT<AR>*>(a.tw_inv),
                         static_cast<const VT<AR>*>(a.scale), a.mod, xcd_map);
      return hipGetLastError();
    }
  }
  hipLaunchKernelGGL((k_pow2<L, MODE, AR>), dim3((unsigned)grid), dim3(NT * PPW), lds_bytes, a.stream,
                     a.y, a.a, a.b, a.B, a.T, static_cast<const VT<AR>*>(a.tw_fwd), static_cast<const VT<AR>*>(a.tw_inv),
                     static_cast<const VT<AR>*>(a.scale), a.mod, xcd_map);
  return hipGetLastError();
}

template <int MODE, int AR>
static hipError_t launch_pow2_mode(const Pow2Launch& a) {
  switch (a.L) {
    case 4: return launch_pow2_L<4, MODE, AR>(a);
    case 5: return launch_pow2_L<5, MODE, AR>(a);
    case 6: return launch_pow2_L<6, MODE, AR>(a);
    case 7: return launch_pow2_L<7, MODE, AR>(a);
    case 8: return launch_pow2_L<8, MODE, AR>(a);
    case 9: return launch_pow2_L<9, MODE, AR>(a);
    case 10: return launch_pow2_L<10, MODE, AR>(a);
    case 11: return launch_pow2_L<11, MODE, AR>(a);
    case 12: return launch_pow2_L<12, MODE, AR>(a);
    case 13: return launch_pow2_L<13, MODE, AR>(a);
    case 14: return launch_pow2_L<14, MODE, AR>(a);
    default: return hipErrorInvalidValue;
  }
}

#ifdef LOLHIP_STAMPS
extern "C" __attribute__((visibility("default"))) int lolhip_debug_set_stamps(unsigned long long* dev) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_buf), &dev, sizeof(dev));
}
#endif

template <int AR>
static hipError_t launch_pow2_ar(const Pow2Launch& a, int mode) {
  switch (mode) {
    case 0: return launch_pow2_mode<0, AR>(a);
    case 1: return launch_pow2_mode<1, AR>(a);
    case 2: return launch_pow2_mode<2, AR>(a);
    default: return hipErrorInvalidValue;
  }
}
hipError_t launch_pow2(const Pow2Launch& a, int mode) {
  switch (a.arith) {
    case 0: return launch_pow2_ar<0>(a, mode);
    case 1: return launch_pow2_ar<1>(a, mode);
    case 2: return launch_pow2_ar<2>(a, mode);
    default: return hipErrorInvalidValue;
  }
}

// =============================================================================
// fused key switch (m = 2^k, every modulus < 2^30):
//   out_k = addend_k + sum_j crt(reduce(digit_j(c2))) * hint_jk,  k = 0, 1
// `switch` of lol-apps SymmSHE.hs:312-314 in ONE pass: what the reference does as
// decompose -> reduce -> adviseCRT -> knapsack over boxed vectors, and the unfused path here as
// three launches through an [L][B][n][T] digit slab (3 + 4L slab passes), reads c2 once per
// target component (L2 hits after the first), the addends once, and writes the two outputs
// once: 5 slab passes.  One workgroup item = (ciphertext b, target component s); the digit
// loop is a run-time loop around one register-resident forward transform; the two
// accumulators (64-bit sums of raw products, 64 VGPRs) stay in registers in the transform's
// output layout.
// =============================================================================
template <int L>
__global__ void __launch_bounds__(pow2_threads(L), 4)
k_keyswitch(const i64* __restrict__ c2, const i64* __restrict__ hint, const i64* addend, i64* out, i64 B, int T,
            const u32* __restrict__ tw_fwd, const ModCtx* __restrict__ mod, DecompParams dp, u32 magic32, int xcd_map) {
  constexpr int AR = 2;
  constexpr int K = 2;
  using S = Sched<L>;
  using V = u32;
  constexpr int n = 1 << L;
  constexpr int NT = 1 << (L - R);
  constexpr int PPW = NT >= 256 ? 1 : 256 / NT;
  constexpr int LDSW = n + n / 16 + twl_words(n);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  V* lds = reinterpret_cast<V*>(smem) + (threadIdx.x / NT) * LDSW;
  V* lds_tw = lds + n + n / 16;
  const int tau = threadIdx.x % NT;

  const i64 item0 = (i64)blockIdx.x * PPW;
  const int slot = (PPW == 1) ? 0 : (int)(threadIdx.x / NT);
  const i64 item = item0 + slot;
  i64 b, b0; int s;
  if (xcd_map) { i64 g = item / (8 * (i64)T); int r = (int)(item % (8 * T)); b = g * 8 + (r & 7); s = r >> 3; b0 = b; }
  else { b = item / T; s = (int)(item % T); b0 = item0 / T; }
  if constexpr (NT >= 64) {
    s = __builtin_amdgcn_readfirstlane(s);
    b = (i64)(((u64)(u32)__builtin_amdgcn_readfirstlane((int)(b >> 32)) << 32) | (u32)__builtin_amdgcn_readfirstlane((int)b));
  }
  b0 = (i64)(((u64)(u32)__builtin_amdgcn_readfirstlane((int)(b0 >> 32)) << 32) | (u32)__builtin_amdgcn_readfirstlane((int)b0));
  const ModCtx ms = mod[s];
  const QK32 qk(ms.q);
  const u32 mu32 = (u32)(ms.mu >> 32);                   // floor(2^32 / q_s)

  const u64 slab = (u64)B * n * T;                       // elements per [B][n][T] slab
  const u64 win = (u64)(B - b0) * n * T * 8;
  const u32 wbytes = win > 0xFFFFFFFFull ? 0xFFFFFFFFu : (u32)win;
  const size_t wbase = (size_t)b0 * n * T;
  const rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void*)(c2 + wbase), 0, wbytes, 0x00020000);
  const rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc((void*)hint, 0, (u32)dp.L * K * n * (u32)T * 8u, 0x00020000);
  const rsrc_t ro0 = __builtin_amdgcn_make_buffer_rsrc((void*)(out + wbase), 0, wbytes, 0x00020000);
  const rsrc_t ro1 = __builtin_amdgcn_make_buffer_rsrc((void*)(out + slab + wbase), 0, wbytes, 0x00020000);
  const rsrc_t ra0 = __builtin_amdgcn_make_buffer_rsrc((void*)(addend ? addend + wbase : nullptr), 0, addend ? wbytes : 0, 0x00020000);
  const rsrc_t ra1 = __builtin_amdgcn_make_buffer_rsrc((void*)(addend ? addend + slab + wbase : nullptr), 0, addend ? wbytes : 0, 0x00020000);
  TwCtxT<V> tw;
  tw.fwd = __builtin_amdgcn_make_buffer_rsrc((void*)tw_fwd, 0, (u32)T * n * 8u, 0x00020000);
  tw.inv = tw.fwd;
  tw.comp = (u32)s * (u32)n * 8u;
  tw.pf = tw_fwd + (size_t)s * n * 2;
  tw.pi = tw.pf;
  tw.lds_tw = lds_tw;
  tw.sc0 = 0; tw.sc1 = 0;
  if constexpr (L >= TWL_MIN_L) tw_fill_lds<NT>(lds_tw, tw.fwd, tw.comp, n, tau);

  constexpr Lay LIO = S::io();
  constexpr Lay LFIN = S::final_layout();
  const u32 uT8 = (u32)T * 8u;
  const u32 pofs0 = (u32)(b - b0) * (u32)n * (u32)T * 8u;                 // this ciphertext inside the window, component 0
  const u32 off_io = pofs0 + (u32)xthr<LIO>(tau) * uT8;                   // + t*8 per source component
  const u32 off_fin = pofs0 + (u32)s * 8u + (u32)xthr<LFIN>(tau) * uT8;   // outputs / addends
  const u32 off_h = (u32)s * 8u + (u32)xthr<LFIN>(tau) * uT8;             // hint polynomials (no batch axis)
  const u32 hstride = (u32)n * uT8;                                       // bytes per hint polynomial

  // 64-bit accumulators of raw products: digit-hat < q and hint < q, so 16 products of < 2^60
  // fit before a reduction is due — one multiply-add per (digit, hint coefficient) instead of a
  // modular product
  u64 acc0[E], acc1[E];
#pragma unroll
  for (int e = 0; e < E; ++e) { acc0[e] = 0; acc1[e] = 0; }
  auto fold = [&](u64 x) -> u64 {                       // x mod q, any 64-bit x
    const u64 r = x - __umul64hi(x, ms.mu) * ms.q;     // [0, 2q)
    return (u64)csub32((u32)r, qk.q);
  };
  const int shift = (int)(dp.base / 2);
  const u32 base = (u32)dp.base;
  int j = 0;
  for (int t = 0; t < T; ++t) {
    const u32 qt = (u32)mod[t].q;
    int vv[E];                                   // centred lift of component t, then its running quotient
    {
      u64 raw[E];
#pragma unroll
      for (int e = 0; e < E; ++e) raw[e] = load_u64(rc, off_io, (u32)lay_tab<LIO>.xr[e] * uT8 + (u32)t * 8u);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const u32 x = (u32)raw[e] + (qt & (u32)((i64)raw[e] >> 63));      // (-q,q) -> [0,q)
        vv[e] = (2 * x < qt) ? (int)x : (int)x - (int)qt;                 // ZqBasic.hs:92-94
      }
    }
    const int kt = dp.k[t];
    for (int kd = 0; kd < kt; ++kd, ++j) {
      V v[E];
#pragma unroll
      for (int e = 0; e < E; ++e) {
        int d;
        if (kd + 1 < kt) {                       // centred remainder, quotient carries on (Numeric.hs:202-205,227-234)
          const int a = vv[e] + shift;
          const u32 nn = a >= 0 ? (u32)a : (u32)(-a - 1);
          const u32 t1 = __umulhi(magic32, nn);
          const u32 qq = (t1 + ((nn - t1) >> dp.sh1)) >> dp.sh2;
          const int qd = a >= 0 ? (int)qq : -(int)qq - 1;
          d = a - qd * (int)base - shift;
          vv[e] = qd;
        } else {
          d = vv[e];
        }
        const u32 ad = d >= 0 ? (u32)d : (u32)(-d);                       // reduce into component s
        u32 r = ad - __umulhi(ad, mu32) * qk.q;                           // [0, 2q)
        r = csub32(r, qk.q);
        v[e] = (d < 0 && r != 0) ? qk.q - r : r;
      }
      fwd_transform<AR, L, LIO, 10, true>(v, lds, tw, tau, qk);
      const u32 hoff = (u32)j * (u32)K * hstride;
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const u32 eo = (u32)lay_tab<LFIN>.xr[e] * uT8;
        const u32 h0 = (u32)load_u64(rh, off_h, hoff + eo);
        const u32 h1 = (u32)load_u64(rh, off_h, hoff + hstride + eo);
        const u32 vc = canon_fwd<AR>(v[e], qk);                        // [0,4q) -> [0,q)
        acc0[e] += (u64)h0 * vc;
        acc1[e] += (u64)h1 * vc;
      }
      if ((j & 15) == 15) {
#pragma unroll
        for (int e = 0; e < E; ++e) { acc0[e] = fold(acc0[e]); acc1[e] = fold(acc1[e]); }
      }
    }
  }
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const u32 eo = (u32)lay_tab<LFIN>.xr[e] * uT8;
    u32 r0 = (u32)fold(acc0[e]), r1 = (u32)fold(acc1[e]);
    if (addend) {
      r0 = csub32(r0 + from_i64<AR>((i64)load_u64(ra0, off_fin, eo), qk), qk.q);
      r1 = csub32(r1 + from_i64<AR>((i64)load_u64(ra1, off_fin, eo), qk), qk.q);
    }
    store_u64(ro0, off_fin, eo, (u64)r0);
    store_u64(ro1, off_fin, eo, (u64)r1);
  }
}

template <int L>
static hipError_t launch_keyswitch_L(const KeySwitchLaunch& a) {
  constexpr int n = 1 << L;
  constexpr int NT = 1 << (L - R);
  constexpr int PPW = NT >= 256 ? 1 : 256 / NT;
  constexpr int LDSW = n + n / 16 + twl_words(n);
  const size_t lds_bytes = (size_t)PPW * LDSW * sizeof(u32);
  const i64 items = a.B * a.T;
  const int xcd_map = (a.T > 1 && PPW == 1 && a.B % 8 == 0) ? 1 : 0;
  const i64 grid = (items + PPW - 1) / PPW;
  if (grid == 0) return hipSuccess;
  static bool attr_set = false;
  if (!attr_set && lds_bytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_keyswitch<L>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  hipLaunchKernelGGL((k_keyswitch<L>), dim3((unsigned)grid), dim3(NT * PPW), lds_bytes, a.stream, a.c2, a.hint,
                     a.addend, a.out, a.B, a.T, a.tw_fwd32, a.mod, a.dp, a.magic32, xcd_map);
  return hipGetLastError();
}

hipError_t launch_keyswitch_fused(const KeySwitchLaunch& a) {
  switch (a.L) {
    case 4: return launch_keyswitch_L<4>(a);
    case 5: return launch_keyswitch_L<5>(a);
    case 6: return launch_keyswitch_L<6>(a);
    case 7: return launch_keyswitch_L<7>(a);
    case 8: return launch_keyswitch_L<8>(a);
    case 9: return launch_keyswitch_L<9>(a);
    case 10: return launch_keyswitch_L<10>(a);
    case 11: return launch_keyswitch_L<11>(a);
    case 12: return launch_keyswitch_L<12>(a);
    case 13: return launch_keyswitch_L<13>(a);
    case 14: return launch_keyswitch_L<14>(a);
    default: return hipErrorInvalidValue;
  }
}

// =============================================================================
// generic path: stage-program interpreter
// =============================================================================

__device__ __forceinline__ u64 dot_reduce(unsigned __int128 acc, const ModCtx& mc) {
  return reduce128((u64)(acc >> 64), (u64)acc, mc);
}

// k mod q for a small constant k (almost always k < q: skip the 64-bit division)
__device__ __forceinline__ u64 smallmod(u64 k, u64 q) { return k < q ? k : k % q; }
// x / v for x < 2^20 with M = floor(2^40/v)+1 (integer division is ~40 instructions on this ISA)
template <bool MAGIC> __device__ __forceinline__ int fdiv(int x, u64 M, int v) {
  if constexpr (MAGIC) return (int)(((u64)(u32)x * M) >> 40); else return x / v;
}
// a*b mod q for the generic path; Q32: both operands (and q) are below 2^32, one 32x32 product
template <bool Q32> __device__ __forceinline__ u64 gmul(u64 a, u64 b, const ModCtx& mc) {
  if constexpr (Q32) return rem128(0, (u64)(u32)a * (u32)b, mc);
  else return mulmod(a, b, mc);
}
// out-of-place evaluation of one output element of one stage.
// Q32 (every modulus of the plan < 2^32, the reference's own valid domain and below): the
// dot products multiply 32-bit operands — one v_mad_u64_u32 per term instead of a 64x64->128
// product — still accumulated exactly in 128 bits and reduced once.
template <bool MAGIC, bool Q32>
__device__ __forceinline__ u64 stage_eval(const Stage& st, const u64* __restrict__ in, int x,
                                          const u64* __restrict__ cst, const ModCtx& mc) {
  const u64 q = mc.q;
  u64 out;
  if (st.kind == ST_DIAG) {
    out = in[x];
  } else if (st.kind == ST_SCALE) {
    return gmul<Q32>(in[x], cst[st.tw_off], mc);
  } else {
    const int rts = st.rts, d = st.d, p = st.p;
    const int xb = fdiv<MAGIC>(x, st.m_rts, rts);
    const int i = xb - fdiv<MAGIC>(xb, st.m_d, d) * d;
    const u64* vin = in + (x - i * rts);
    switch (st.kind) {
      case ST_DFTP:
      case ST_CRTP:
      case ST_CRTPINV: {
        const u64* row = cst + st.mat_off + i * d;      // dense matrix row (plan.cpp)
        unsigned __int128 acc = 0;
        u64 part = 0;
        int cnt = 0;
        for (int c = 0; c < d; ++c) {
          if constexpr (Q32) acc += (unsigned __int128)((u64)(u32)vin[c * rts] * (u32)row[c]);
          else acc += (unsigned __int128)vin[c * rts] * row[c];
          if (++cnt == 8) {   // 8 products of < 2^124 fit in 128 bits
            part = addmod(part, dot_reduce(acc, mc), q); acc = 0;
            cnt = 0;
          }
        }
        out = cnt ? addmod(part, dot_reduce(acc, mc), q) : part;
        break;
      }
      case ST_L: {
        u64 s = 0;
        for (int c = 0; c <= i; ++c) s = addmod(s, vin[c * rts], q);
        out = s;
        break;
      }
      case ST_LINV:
        out = (i == 0) ? vin[0] : submod(vin[i * rts], vin[(i - 1) * rts], q);
        break;
      case ST_GPOW: {
        const u64 last = vin[(d - 1) * rts];
        out = addmod(vin[i * rts], last, q);
        if (i > 0) out = submod(out, vin[(i - 1) * rts], q);
        break;
      }
      case ST_GDEC: {
        if (i > 0) {
          out = submod(vin[i * rts], vin[(i - 1) * rts], q);
        } else {
          u64 s = vin[0];
          for (int c = 0; c < d; ++c) s = addmod(s, vin[c * rts], q);
          out = s;
        }
        break;
      }
      case ST_GINVPOW: {
        u64 le = 0, re = 0;
        for (int c = 0; c < d; ++c) {
          if (c <= i) le = addmod(le, vin[c * rts], q); else re = addmod(re, vin[c * rts], q);
        }
        out = submod(gmul<Q32>(smallmod((u64)(p - 1 - i), q), le, mc), gmul<Q32>(smallmod((u64)(i + 1), q), re, mc), q);
        break;
      }
      case ST_GINVDEC: {
        u64 s = 0, hi = 0;
        for (int c = 0; c < d; ++c) {
          s = addmod(s, gmul<Q32>(smallmod((u64)(c + 1), q), vin[c * rts], mc), q);
          if (c > i) hi = addmod(hi, vin[c * rts], q);
        }
        out = submod(s, gmul<Q32>(smallmod((u64)p, q), hi, mc), q);
        break;
      }
      default:
        out = in[x];
    }
  }
  if (st.tw_off >= 0) {
    const int xd = fdiv<MAGIC>(x, st.m_twdiv, st.tw_div);
    out = gmul<Q32>(out, cst[st.tw_off + xd - fdiv<MAGIC>(xd, st.m_twmod, st.tw_mod) * st.tw_mod], mc);
  }
  return out;
}

template <bool MAGIC, bool Q32>
__global__ void __launch_bounds__(1024)
k_generic(i64* __restrict__ y, i64 B, int T, int n, const Stage* __restrict__ stages, int nstages,
          const u64* __restrict__ consts, int cpc, const ModCtx* __restrict__ mod, int ppw,
          u64* __restrict__ scratch, i64 ngroups) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const i64 items = ngroups * T;
  for (i64 item = blockIdx.x; item < items; item += gridDim.x) {
    const i64 g = item / T;
    const int t = (int)(item % T);
    const i64 b0 = g * ppw;
    const int np = (int)((B - b0) < ppw ? (B - b0) : ppw);   // polynomials in this group
    const int tot = np * n;
    const ModCtx mc = mod[t];
    const u64* cst = consts + (size_t)t * cpc;
    const u64 n_magic = (((u64)1 << 40) / (u64)n) + 1;
    u64* bufA;
    u64* bufB;
    if (scratch) {
      bufA = scratch + (size_t)blockIdx.x * 2 * n;   // ppw == 1 on this path
      bufB = bufA + n;
    } else {
      bufA = reinterpret_cast<u64*>(smem);
      bufB = bufA + (size_t)ppw * n;
    }
    for (int x = threadIdx.x; x < tot; x += blockDim.x)
      bufA[x] = canon_in(y[((size_t)b0 * n + x) * T + t], mc.q);
    __syncthreads();
    for (int s = 0; s < nstages; ++s) {
      const Stage st = stages[s];
      for (int x = threadIdx.x; x < tot; x += blockDim.x) {
        const int pi = fdiv<MAGIC>(x, n_magic, n), xi = x - pi * n;
        bufB[x] = stage_eval<MAGIC, Q32>(st, bufA + pi * n, xi, cst, mc);
      }
      __syncthreads();
      u64* tmp = bufA; bufA = bufB; bufB = tmp;
    }
    for (int x = threadIdx.x; x < tot; x += blockDim.x)
      y[((size_t)b0 * n + x) * T + t] = (i64)bufA[x];
    __syncthreads();
  }
}

// -----------------------------------------------------------------------------
// vector-per-thread interpreter for small primes (d <= 13): one thread owns one d-vector of a
// stage (I (x) A_p (x) I_rts), reads it from LDS once, applies A_p in registers and writes it
// back IN PLACE — one LDS buffer instead of ping-pong (more polynomials per CU), one LDS read
// per coefficient instead of d, matrix entries through scalar loads (they are the same for
// every vector), and the O(d) forms of L, L^-1, G, G^-1 (l.cpp:28-98, g.cpp:16-123) instead
// of per-output sums.  Same stage program, same results as k_generic.
// -----------------------------------------------------------------------------
template <int D, bool Q32>
__device__ __forceinline__ void stage_vec(const Stage& st, u64* __restrict__ buf, int vec, int n, u64 n_magic,
                                          const u64* __restrict__ cst, const ModCtx& mc) {
  const u64 q = mc.q;
  const int rts = st.rts;
  const int blk = fdiv<true>(vec, st.m_rts, rts), r = vec - blk * rts;
  const int x0 = blk * D * rts + r;                       // position of element 0 in the (packed) buffer
  u64* base = buf + x0;
  u64 v[D], o[D];
#pragma unroll
  for (int i = 0; i < D; ++i) v[i] = base[i * rts];
  switch (st.kind) {
    case ST_DFTP:
    case ST_CRTP:
    case ST_CRTPINV: {
      const u64* M = cst + st.mat_off;
      static_assert(D <= 16, "16 products below 2^124 (q < 2^62) fit in 128 bits");
#pragma unroll
      for (int i = 0; i < D; ++i) {
        unsigned __int128 acc = 0;
#pragma unroll
        for (int c = 0; c < D; ++c) {
          if constexpr (Q32) acc += (unsigned __int128)((u64)(u32)v[c] * (u32)M[i * D + c]);
          else acc += (unsigned __int128)v[c] * M[i * D + c];
        }
        // Q32: the sum is below 16 * 2^64, so its high word is below q whenever q > 16: one division step
        if (Q32 && q > 16) o[i] = rem128((u64)(acc >> 64), (u64)acc, mc);
        else o[i] = dot_reduce(acc, mc);
      }
      break;
    }
    case ST_L: {                         // prefix sums (l.cpp:28-57)
      u64 s = 0;
#pragma unroll
      for (int i = 0; i < D; ++i) { s = addmod(s, v[i], q); o[i] = s; }
      break;
    }
    case ST_LINV: {                      // adjacent differences (l.cpp:67-98)
      o[0] = v[0];
#pragma unroll
      for (int i = 1; i < D; ++i) o[i] = submod(v[i], v[i - 1], q);
      break;
    }
    case ST_GPOW: {                      // g.cpp:16-35
      const u64 last = v[D - 1];
      o[0] = addmod(v[0], last, q);
#pragma unroll
      for (int i = 1; i < D; ++i) o[i] = submod(addmod(v[i], last, q), v[i - 1], q);
      break;
    }
    case ST_GDEC: {                      // g.cpp:37-58
      u64 s = v[0];
#pragma unroll
      for (int c = 0; c < D; ++c) s = addmod(s, v[c], q);
      o[0] = s;
#pragma unroll
      for (int i = 1; i < D; ++i) o[i] = submod(v[i], v[i - 1], q);
      break;
    }
    case ST_GINVPOW: {                   // g.cpp:60-90: (p-1-i) * sum_{c<=i} - (i+1) * sum_{c>i}
      u64 tot = 0;
#pragma unroll
      for (int c = 0; c < D; ++c) tot = addmod(tot, v[c], q);
      u64 le = 0;
#pragma unroll
      for (int i = 0; i < D; ++i) {
        le = addmod(le, v[i], q);
        const u64 re = submod(tot, le, q);
        o[i] = submod(gmul<Q32>(smallmod((u64)(st.p - 1 - i), q), le, mc), gmul<Q32>(smallmod((u64)(i + 1), q), re, mc), q);
      }
      break;
    }
    case ST_GINVDEC: {                   // g.cpp:92-123: sum_c (c+1) v_c - p * sum_{c>i} v_c
      u64 s = 0, tot = 0;
#pragma unroll
      for (int c = 0; c < D; ++c) {
        s = addmod(s, gmul<Q32>(smallmod((u64)(c + 1), q), v[c], mc), q);
        tot = addmod(tot, v[c], q);
      }
      const u64 pm = smallmod((u64)st.p, q);
      u64 le = 0;
#pragma unroll
      for (int i = 0; i < D; ++i) {
        le = addmod(le, v[i], q);
        o[i] = submod(s, gmul<Q32>(pm, submod(tot, le, q), mc), q);
      }
      break;
    }
    default:
#pragma unroll
      for (int i = 0; i < D; ++i) o[i] = v[i];
  }
  if (st.tw_off >= 0) {
    const int pi = fdiv<true>(x0, n_magic, n);
    const int xi0 = x0 - pi * n;                          // position inside its polynomial
#pragma unroll
    for (int i = 0; i < D; ++i) {
      const int xd = fdiv<true>(xi0 + i * rts, st.m_twdiv, st.tw_div);
      o[i] = gmul<Q32>(o[i], cst[st.tw_off + xd - fdiv<true>(xd, st.m_twmod, st.tw_mod) * st.tw_mod], mc);
    }
  }
#pragma unroll
  for (int i = 0; i < D; ++i) base[i * rts] = o[i];
}

// every vector length the vector interpreter instantiates (p-1 and p for p = 3..13)
__host__ __device__ constexpr bool vec_len_ok(int d) {
  return d == 2 || d == 3 || d == 4 || d == 5 || d == 6 || d == 7 || d == 10 || d == 11 || d == 12 || d == 13;
}

template <bool Q32>
__global__ void __launch_bounds__(512)
k_generic_vec(i64* y, const i64* src, i64 B, int T, int n, const Stage* __restrict__ stages, int nstages,
              const u64* __restrict__ consts, int cpc, const ModCtx* __restrict__ mod, int ppw, i64 ngroups) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u64* buf = reinterpret_cast<u64*>(smem);
  const i64 items = ngroups * T;
  const u64 n_magic = (((u64)1 << 40) / (u64)n) + 1;
  for (i64 item = blockIdx.x; item < items; item += gridDim.x) {
    const i64 g = item / T;
    const int t = (int)(item % T);
    const i64 b0 = g * ppw;
    const int np = (int)((B - b0) < ppw ? (B - b0) : ppw);
    const int tot = np * n;
    const ModCtx mc = mod[t];
    const u64* cst = consts + (size_t)t * cpc;
    for (int x = threadIdx.x; x < tot; x += blockDim.x) buf[x] = canon_in(src[((size_t)b0 * n + x) * T + t], mc.q);
    __syncthreads();
    for (int s = 0; s < nstages; ++s) {
      const Stage st = stages[s];
      if (st.kind == ST_DIAG || st.kind == ST_SCALE) {
        for (int x = threadIdx.x; x < tot; x += blockDim.x) {
          const int pi = fdiv<true>(x, n_magic, n);
          buf[x] = stage_eval<true, Q32>(st, buf + pi * n, x - pi * n, cst, mc);   // element-wise: in place is safe
        }
      } else {
        const int nvec = tot / st.d;
        for (int vec = threadIdx.x; vec < nvec; vec += blockDim.x) {
          switch (st.d) {
            case 2: stage_vec<2, Q32>(st, buf, vec, n, n_magic, cst, mc); break;
            case 3: stage_vec<3, Q32>(st, buf, vec, n, n_magic, cst, mc); break;
            case 4: stage_vec<4, Q32>(st, buf, vec, n, n_magic, cst, mc); break;
            case 5: stage_vec<5, Q32>(st, buf, vec, n, n_magic, cst, mc); break;
            case 6: stage_vec<6, Q32>(st, buf, vec, n, n_magic, cst, mc); break;
            case 7: stage_vec<7, Q32>(st, buf, vec, n, n_magic, cst, mc); break;
            case 10: stage_vec<10, Q32>(st, buf, vec, n, n_magic, cst, mc); break;
            case 11: stage_vec<11, Q32>(st, buf, vec, n, n_magic, cst, mc); break;
            case 12: stage_vec<12, Q32>(st, buf, vec, n, n_magic, cst, mc); break;
            case 13: stage_vec<13, Q32>(st, buf, vec, n, n_magic, cst, mc); break;
            default: break;   // excluded on the host (vec_len_ok)
          }
        }
      }
      __syncthreads();
    }
    for (int x = threadIdx.x; x < tot; x += blockDim.x) y[((size_t)b0 * n + x) * T + t] = (i64)buf[x];
    __syncthreads();
  }
}

hipError_t launch_generic(const GenericLaunch& a) {
  if (a.B == 0) return hipSuccess;
  const size_t lds_budget = 152 * 1024;
  const size_t per_poly = 2 * (size_t)a.n * sizeof(u64);
  int ppw = 1;
  u64* scratch = nullptr;
  size_t lds_bytes;
  i64 grid;
  if (a.vec_ok && (size_t)a.n * sizeof(u64) <= 64 * 1024) {
    // vector interpreter: one LDS buffer; pack small polynomials up to ~2048 coefficients
    while ((size_t)(ppw * 2) * a.n <= 2048 && ppw * 2 <= a.B) ppw *= 2;
    lds_bytes = (size_t)ppw * a.n * sizeof(u64);
    const i64 ngroups = (a.B + ppw - 1) / ppw;
    grid = ngroups * a.T;
    if (grid > 65536) grid = 65536;
    // measured at m = 15015 / 1728 / 14336: 46 KiB polynomials want 512 threads (three groups per
    // CU = 6 waves/SIMD), 1-2 K coefficient groups 128
    const size_t coeffs = (size_t)ppw * a.n;
    const int vthreads = coeffs >= 4096 ? 512 : (coeffs >= 2048 ? 256 : 128);
    if (a.q32)
      hipLaunchKernelGGL((k_generic_vec<true>), dim3((unsigned)grid), dim3(vthreads), lds_bytes, a.stream, a.y,
                         a.src ? a.src : a.y, a.B, a.T, (int)a.n, a.stages, a.nstages, a.consts, a.cpc, a.mod, ppw, ngroups);
    else
      hipLaunchKernelGGL((k_generic_vec<false>), dim3((unsigned)grid), dim3(vthreads), lds_bytes, a.stream, a.y,
                         a.src ? a.src : a.y, a.B, a.T, (int)a.n, a.stages, a.nstages, a.consts, a.cpc, a.mod, ppw, ngroups);
    return hipGetLastError();
  }
  if (per_poly <= lds_budget) {
    // pack small polynomials: aim for >= 2048 coefficients per workgroup, <= 32 KiB per buffer
    while ((size_t)(ppw * 2) * a.n <= 2048 && ppw * 2 <= a.B) ppw *= 2;
    lds_bytes = (size_t)ppw * per_poly;
    const i64 ngroups = (a.B + ppw - 1) / ppw;
    grid = ngroups * a.T;
    if (grid > 65536) grid = 65536;
    static bool attr_set = false;
    if (!attr_set) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_generic<true, false>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_budget);
      if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_generic<true, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_budget);
      if (e != hipSuccess) return e;
      attr_set = true;
    }
    // a workgroup that fills a CU's LDS on its own should also fill its SIMDs
    const int threads = ((size_t)ppw * a.n >= 4096) ? 1024 : ((size_t)ppw * a.n >= 1024 ? 512 : 256);
    if (a.q32)
      hipLaunchKernelGGL((k_generic<true, true>), dim3((unsigned)grid), dim3(threads), lds_bytes, a.stream, a.y, a.B, a.T,
                         (int)a.n, a.stages, a.nstages, a.consts, a.cpc, a.mod, ppw, scratch, ngroups);
    else
      hipLaunchKernelGGL((k_generic<true, false>), dim3((unsigned)grid), dim3(threads), lds_bytes, a.stream, a.y, a.B, a.T,
                         (int)a.n, a.stages, a.nstages, a.consts, a.cpc, a.mod, ppw, scratch, ngroups);
  } else {
    if (!a.scratch) return hipErrorInvalidValue;
    grid = a.B * a.T;
    const i64 maxg = (i64)(a.scratch_bytes / (2 * (size_t)a.n * sizeof(u64)));
    if (grid > maxg) grid = maxg;
    if (grid < 1) return hipErrorInvalidValue;
    if (a.n < (1 << 20))
      hipLaunchKernelGGL((k_generic<true, false>), dim3((unsigned)grid), dim3(1024), 0, a.stream, a.y, a.B, a.T, (int)a.n,
                         a.stages, a.nstages, a.consts, a.cpc, a.mod, 1, a.scratch, a.B);
    else
      hipLaunchKernelGGL((k_generic<false, false>), dim3((unsigned)grid), dim3(1024), 0, a.stream, a.y, a.B, a.T, (int)a.n,
                         a.stages, a.nstages, a.consts, a.cpc, a.mod, 1, a.scratch, a.B);
  }
  return hipGetLastError();
}

// =============================================================================
// streaming kernels
// =============================================================================

// a[i] = a[i] * b[i mod bperiod]  (bperiod = total length for mulRq, n*T for g vectors)
__global__ void __launch_bounds__(256)
k_pointwise_mul(i64* __restrict__ a, const i64* __restrict__ b, i64 total, i64 bperiod, int T,
                const ModCtx* __restrict__ mod) {
  const i64 stride = (i64)gridDim.x * blockDim.x;
  for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const ModCtx mc = mod[i % T];
    const u64 x = canon_in(a[i], mc.q), z = canon_in(b[i % bperiod], mc.q);
    a[i] = (i64)mulmod(x, z, mc);
  }
}

hipError_t launch_pointwise_mul(hipStream_t s, i64* a, const i64* b, i64 total, i64 bperiod, int T, const ModCtx* mod) {
  if (total == 0) return hipSuccess;
  i64 blocks = (total + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(k_pointwise_mul, dim3((unsigned)blocks), dim3(256), 0, s, a, b, total, bperiod, T, mod);
  return hipGetLastError();
}

// out[b][i][t] = +-in[b][idx[i]][t] or 0  (embedPow/Dec/CRT, twacePowDec; Extension.hs:54-101)
__global__ void __launch_bounds__(256)
k_gather(i64* __restrict__ out, const i64* __restrict__ in, const int32_t* __restrict__ idx, i64 B, i64 n_out,
         i64 n_in, int T, const ModCtx* __restrict__ mod) {
  const i64 total = B * n_out * T;
  const i64 stride = (i64)gridDim.x * blockDim.x;
  for (i64 g = (i64)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += stride) {
    const int t = (int)(g % T);
    const i64 r = g / T;
    const i64 i = r % n_out, b = r / n_out;
    const int32_t e = idx[i];
    i64 val = 0;
    if (e >= 0) {
      const u64 q = mod[t].q;
      const u64 x = canon_in(in[(b * n_in + (e & (EMBED_NEG_FLAG_DEV - 1))) * T + t], q);
      val = (i64)((e & EMBED_NEG_FLAG_DEV) ? (x == 0 ? 0 : q - x) : x);
    }
    out[g] = val;
  }
}

hipError_t launch_gather(hipStream_t s, i64* out, const i64* in, const int32_t* idx, i64 B, i64 n_out, i64 n_in,
                         int T, const ModCtx* mod) {
  const i64 total = B * n_out * T;
  if (total == 0) return hipSuccess;
  i64 blocks = (total + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(k_gather, dim3((unsigned)blocks), dim3(256), 0, s, out, in, idx, B, n_out, n_in, T, mod);
  return hipGetLastError();
}

// twaceCRT (Extension.hs:110-129): out[b][i][t] = sum_{r<rel} tweak[e]*in[b][e][t], e = idx[i*rel + r]
__global__ void __launch_bounds__(256)
k_twace_crt(i64* __restrict__ out, const i64* __restrict__ in, const int32_t* __restrict__ idx,
            const i64* __restrict__ tweak, i64 B, i64 n_out, i64 n_in, int T, const ModCtx* __restrict__ mod) {
  const i64 total = B * n_out * T;
  const i64 stride = (i64)gridDim.x * blockDim.x;
  const int rel = (int)(n_in / n_out);
  for (i64 g = (i64)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += stride) {
    const int t = (int)(g % T);
    const i64 r = g / T;
    const i64 i = r % n_out, b = r / n_out;
    const ModCtx mc = mod[t];
    u64 acc = 0;
    for (int k = 0; k < rel; ++k) {
      const i64 e = idx[i * rel + k];
      const u64 x = canon_in(in[(b * n_in + e) * T + t], mc.q);
      acc = addmod(acc, mulmod(x, (u64)tweak[e * T + t], mc), mc.q);
    }
    out[g] = (i64)acc;
  }
}

hipError_t launch_twace_crt(hipStream_t s, i64* out, const i64* in, const int32_t* idx, const i64* tweak, i64 B,
                            i64 n_out, i64 n_in, int T, const ModCtx* mod) {
  const i64 total = B * n_out * T;
  if (total == 0) return hipSuccess;
  i64 blocks = (total + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(k_twace_crt, dim3((unsigned)blocks), dim3(256), 0, s, out, in, idx, tweak, B, n_out, n_in, T, mod);
  return hipGetLastError();
}

}  // namespace lolhip
