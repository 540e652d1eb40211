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
// Each thread keeps E = 16 coefficients in registers and does R = 4 levels per
// pass; between passes the polynomial is transposed through LDS.  Pass with base
// bit `lo` holds positions  x = (tau >> lo) << (lo+R) | e << lo | tau & (2^lo-1).

constexpr int R = 4;
constexpr int E = 1 << R;

__device__ __forceinline__ int xpos(int tau, int e, int lo) {
  return ((tau >> lo) << (lo + R)) | (e << lo) | (tau & ((1 << lo) - 1));
}
// one padding word per 16 keeps both the stride-16 and the stride-1 side of every
// transpose conflict-free for ds_write_b64 / ds_read_b64
__device__ __forceinline__ int lpad(int x) { return x + (x >> 4); }

// forward (Cooley-Tukey / Harvey) butterfly: X,Y in [0,4q) -> [0,4q)
__device__ __forceinline__ void bfly_fwd(u64& X, u64& Y, u64 w, u64 wp, u64 q, u64 q2) {
  u64 x = csub(X, q2);
  u64 t = shoup_lazy(Y, w, wp, q);
  X = x + t;
  Y = x - t + q2;
}
// inverse (Gentleman-Sande) butterfly: X,Y in [0,2q) -> [0,2q)
__device__ __forceinline__ void bfly_inv(u64& X, u64& Y, u64 w, u64 wp, u64 q, u64 q2) {
  u64 s = X + Y;
  u64 d = X - Y + q2;
  X = csub(s, q2);
  Y = shoup_lazy(d, w, wp, q);
}

template <int LO, int K0, int K1>
__device__ __forceinline__ void fwd_levels(u64 (&v)[E], const u64* __restrict__ tw, int tau_low, u64 q, u64 q2) {
#pragma unroll
  for (int k = K0; k < K1; ++k) {
#pragma unroll
    for (int e = 0; e < E; ++e) {
      if (e & (1 << k)) continue;
      const int elow = e & ((1 << k) - 1);
      const int idx = (1 << (LO + k)) + (elow << LO) + tau_low;
      const ulonglong2 W = *reinterpret_cast<const ulonglong2*>(tw + 2 * (size_t)idx);
      bfly_fwd(v[e], v[e + (1 << k)], W.x, W.y, q, q2);
    }
  }
}

template <int LO, int K0, int K1, bool FOLD>
__device__ __forceinline__ void inv_levels(u64 (&v)[E], const u64* __restrict__ tw, const u64* __restrict__ sc,
                                           int tau_low, u64 q, u64 q2) {
#pragma unroll
  for (int k = K1 - 1; k >= K0; --k) {
#pragma unroll
    for (int e = 0; e < E; ++e) {
      if (e & (1 << k)) continue;
      const int elow = e & ((1 << k) - 1);
      const int idx = (1 << (LO + k)) + (elow << LO) + tau_low;
      const ulonglong2 W = *reinterpret_cast<const ulonglong2*>(tw + 2 * (size_t)idx);
      if (FOLD && LO + k == 0) {
        // last level: both outputs are scaled by mhat^-1 (crt.cpp:573-579); tw[1] holds
        // psi_2^-1 * mhat^-1 and sc holds mhat^-1, both as Shoup pairs
        u64 s = v[e] + v[e + 1];
        u64 d = v[e] - v[e + 1] + q2;
        v[e] = shoup_lazy(s, sc[0], sc[1], q);
        v[e + 1] = shoup_lazy(d, W.x, W.y, q);
      } else {
        bfly_inv(v[e], v[e + (1 << k)], W.x, W.y, q, q2);
      }
    }
  }
}

template <int LO_FROM, int LO_TO>
__device__ __forceinline__ void transpose(u64 (&v)[E], u64* lds, int tau) {
#pragma unroll
  for (int e = 0; e < E; ++e) lds[lpad(xpos(tau, e, LO_FROM))] = v[e];
  __syncthreads();
#pragma unroll
  for (int e = 0; e < E; ++e) v[e] = lds[lpad(xpos(tau, e, LO_TO))];
  __syncthreads();
}

// pass schedule for n = 2^L: full passes at lo = 0, R, 2R, ...; if R does not
// divide L the last pass sits at lo = L-R and only runs the levels still missing.
template <int L> struct Sched {
  static constexpr int NP = (L + R - 1) / R;
  static constexpr int lo(int p) { return (p == NP - 1) ? (L - R) : p * R; }
  static constexpr int k0(int p) { return (p == NP - 1) ? ((NP - 1) * R - (L - R)) : 0; }
};

template <int L, int P>
__device__ __forceinline__ void fwd_from(u64 (&v)[E], u64* lds, const u64* tw, int tau, u64 q, u64 q2) {
  if constexpr (P < Sched<L>::NP) {
    constexpr int LO = Sched<L>::lo(P);
    fwd_levels<LO, Sched<L>::k0(P), R>(v, tw, tau & ((1 << LO) - 1), q, q2);
    if constexpr (P + 1 < Sched<L>::NP) {
      transpose<LO, Sched<L>::lo(P + 1)>(v, lds, tau);
      fwd_from<L, P + 1>(v, lds, tw, tau, q, q2);
    }
  }
}
template <int L, int P>
__device__ __forceinline__ void inv_from(u64 (&v)[E], u64* lds, const u64* tw, const u64* sc, int tau, u64 q, u64 q2) {
  if constexpr (P >= 0) {
    constexpr int LO = Sched<L>::lo(P);
    inv_levels<LO, Sched<L>::k0(P), R, true>(v, tw, sc, tau & ((1 << LO) - 1), q, q2);
    if constexpr (P > 0) {
      transpose<LO, Sched<L>::lo(P - 1)>(v, lds, tau);
      inv_from<L, P - 1>(v, lds, tw, sc, tau, q, q2);
    }
  }
}

// MODE 0: crt in place, 1: crtInv in place, 2: c = crtInv(crt(a) * crt(b))
template <int L, int MODE>
__global__ void __launch_bounds__((1 << (L - R)) * ((1 << (L - R)) >= 256 ? 1 : 256 / (1 << (L - R))))
k_pow2(i64* __restrict__ y, const i64* __restrict__ a_in, const i64* __restrict__ b_in, i64 B, int T,
       const u64* __restrict__ tw_fwd, const u64* __restrict__ tw_inv, const u64* __restrict__ scale,
       const ModCtx* __restrict__ mod, int xcd_map) {
  constexpr int n = 1 << L;
  constexpr int NT = 1 << (L - R);                  // threads per polynomial
  constexpr int PPW = NT >= 256 ? 1 : 256 / NT;     // polynomials per workgroup
  constexpr int LDSW = n + n / 16;                  // padded words per polynomial
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u64* lds = reinterpret_cast<u64*>(smem) + (threadIdx.x / NT) * LDSW;
  const int tau = threadIdx.x % NT;

  // work item -> (b, t); with xcd_map the T components of one polynomial land on
  // workgroups that share an XCD (equal blockIdx % 8) so its cache lines are
  // fetched from HBM once.  Placement only affects speed.
  i64 item = (i64)blockIdx.x * PPW + threadIdx.x / NT;
  i64 b; int t;
  if (xcd_map) { i64 g = item / (8 * (i64)T); int r = (int)(item % (8 * T)); b = g * 8 + (r & 7); t = r >> 3; }
  else { b = item / T; t = (int)(item % T); }
  const bool live = b < B;          // tail workgroup of a packed launch
  if (!live) { b = B - 1; }         // keep every thread in the barriers; stores are masked

  const ModCtx mc = mod[t];
  const u64 q = mc.q, q2 = 2 * mc.q;
  const u64* twf = tw_fwd + (size_t)t * n * 2;
  const u64* twi = tw_inv + (size_t)t * n * 2;
  const u64* sc = scale + (size_t)t * 2;
  const size_t base = (size_t)b * n;

  u64 v[E];
  if constexpr (MODE == 0 || MODE == 2) {
    const i64* src = (MODE == 2) ? a_in : y;
#pragma unroll
    for (int e = 0; e < E; ++e) v[e] = canon_in(src[(base + xpos(tau, e, 0)) * T + t], q);
    fwd_from<L, 0>(v, lds, twf, tau, q, q2);
  }
  if constexpr (MODE == 2) {
    u64 va[E];
#pragma unroll
    for (int e = 0; e < E; ++e) va[e] = csub(csub(v[e], q2), q);
    __syncthreads();
#pragma unroll
    for (int e = 0; e < E; ++e) v[e] = canon_in(b_in[(base + xpos(tau, e, 0)) * T + t], q);
    fwd_from<L, 0>(v, lds, twf, tau, q, q2);
#pragma unroll
    for (int e = 0; e < E; ++e) v[e] = mulmod(va[e], csub(csub(v[e], q2), q), mc);
    __syncthreads();
  }
  constexpr int LOL = Sched<L>::lo(Sched<L>::NP - 1);   // layout after the forward transform
  if constexpr (MODE == 1) {
#pragma unroll
    for (int e = 0; e < E; ++e) v[e] = canon_in(y[(base + xpos(tau, e, LOL)) * T + t], q);
  }
  if constexpr (MODE == 0) {
    if (live) {
#pragma unroll
      for (int e = 0; e < E; ++e) y[(base + xpos(tau, e, LOL)) * T + t] = (i64)csub(csub(v[e], q2), q);
    }
  } else {
    inv_from<L, Sched<L>::NP - 1>(v, lds, twi, sc, tau, q, q2);
    if (live) {
#pragma unroll
      for (int e = 0; e < E; ++e) y[(base + xpos(tau, e, 0)) * T + t] = (i64)csub(v[e], q);
    }
  }
}

template <int L, int MODE>
static hipError_t launch_pow2_L(const Pow2Launch& a) {
  constexpr int n = 1 << L;
  constexpr int NT = 1 << (L - R);
  constexpr int PPW = NT >= 256 ? 1 : 256 / NT;
  constexpr int LDSW = n + n / 16;
  const size_t lds_bytes = (size_t)PPW * LDSW * sizeof(u64);
  const i64 items = a.B * a.T;
  const int xcd_map = (a.T > 1 && PPW == 1 && a.B % 8 == 0) ? 1 : 0;
  const i64 grid = (items + PPW - 1) / PPW;
  if (grid == 0) return hipSuccess;
  static bool attr_set = false;
  if (!attr_set && lds_bytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pow2<L, MODE>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  hipLaunchKernelGGL((k_pow2<L, MODE>), dim3((unsigned)grid), dim3(NT * PPW), lds_bytes, a.stream,
                     a.y, a.a, a.b, a.B, a.T, a.tw_fwd, a.tw_inv, a.scale, a.mod, xcd_map);
  return hipGetLastError();
}

template <int MODE>
static hipError_t launch_pow2_mode(const Pow2Launch& a) {
  switch (a.L) {
    case 4: return launch_pow2_L<4, MODE>(a);
    case 5: return launch_pow2_L<5, MODE>(a);
    case 6: return launch_pow2_L<6, MODE>(a);
    case 7: return launch_pow2_L<7, MODE>(a);
    case 8: return launch_pow2_L<8, MODE>(a);
    case 9: return launch_pow2_L<9, MODE>(a);
    case 10: return launch_pow2_L<10, MODE>(a);
    case 11: return launch_pow2_L<11, MODE>(a);
    case 12: return launch_pow2_L<12, MODE>(a);
    case 13: return launch_pow2_L<13, MODE>(a);
    case 14: return launch_pow2_L<14, MODE>(a);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_pow2(const Pow2Launch& a, int mode) {
  switch (mode) {
    case 0: return launch_pow2_mode<0>(a);
    case 1: return launch_pow2_mode<1>(a);
    case 2: return launch_pow2_mode<2>(a);
    default: return hipErrorInvalidValue;
  }
}

// =============================================================================
// generic path: stage-program interpreter
// =============================================================================

__device__ __forceinline__ u64 dot_reduce(unsigned __int128 acc, const ModCtx& mc) {
  return reduce128((u64)(acc >> 64), (u64)acc, mc);
}

// out-of-place evaluation of one output element of one stage
__device__ __forceinline__ u64 stage_eval(const Stage& st, const u64* __restrict__ in, int x,
                                          const u64* __restrict__ cst, const ModCtx& mc) {
  const u64 q = mc.q;
  u64 out;
  if (st.kind == ST_DIAG) {
    out = in[x];
  } else if (st.kind == ST_SCALE) {
    return mulmod(in[x], cst[st.tw_off], mc);
  } else {
    const int rts = st.rts, d = st.d, p = st.p;
    const int i = (x / rts) % d;
    const u64* vin = in + (x - i * rts);
    switch (st.kind) {
      case ST_DFTP:
      case ST_CRTP:
      case ST_CRTPINV: {
        const u64* wp = cst + st.wp_off;
        unsigned __int128 acc = 0, sh = 0;
        u64 part = 0, spart = 0;
        int cnt = 0;
        for (int c = 0; c < d; ++c) {
          const u64 xc = vin[c * rts];
          int widx = (st.kind == ST_DFTP) ? (c * i) % p : (st.kind == ST_CRTP) ? (c * (i + 1)) % p : (i * (c + 1)) % p;
          acc += (unsigned __int128)xc * wp[widx];
          if (st.kind == ST_CRTPINV) sh += (unsigned __int128)xc * wp[p - c - 1];
          if (++cnt == 8) {   // 8 products of < 2^124 fit in 128 bits
            part = addmod(part, dot_reduce(acc, mc), q); acc = 0;
            if (st.kind == ST_CRTPINV) { spart = addmod(spart, dot_reduce(sh, mc), q); sh = 0; }
            cnt = 0;
          }
        }
        out = addmod(part, dot_reduce(acc, mc), q);
        if (st.kind == ST_CRTPINV) out = submod(out, addmod(spart, dot_reduce(sh, mc), q), q);
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
        out = submod(mulmod((u64)(p - 1 - i) % q, le, mc), mulmod((u64)(i + 1) % q, re, mc), q);
        break;
      }
      case ST_GINVDEC: {
        u64 s = 0, hi = 0;
        for (int c = 0; c < d; ++c) {
          s = addmod(s, mulmod((u64)(c + 1) % q, vin[c * rts], mc), q);
          if (c > i) hi = addmod(hi, vin[c * rts], q);
        }
        out = submod(s, mulmod((u64)p % q, hi, mc), q);
        break;
      }
      default:
        out = in[x];
    }
  }
  if (st.tw_off >= 0) out = mulmod(out, cst[st.tw_off + (x / st.tw_div) % st.tw_mod], mc);
  return out;
}

__global__ void __launch_bounds__(256)
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
        const int pi = x / n, xi = x - pi * n;
        bufB[x] = stage_eval(st, bufA + pi * n, xi, cst, mc);
      }
      __syncthreads();
      u64* tmp = bufA; bufA = bufB; bufB = tmp;
    }
    for (int x = threadIdx.x; x < tot; x += blockDim.x)
      y[((size_t)b0 * n + x) * T + t] = (i64)bufA[x];
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
  if (per_poly <= lds_budget) {
    // pack small polynomials: aim for >= 2048 coefficients per workgroup, <= 32 KiB per buffer
    while ((size_t)(ppw * 2) * a.n <= 2048 && ppw * 2 <= a.B) ppw *= 2;
    lds_bytes = (size_t)ppw * per_poly;
    const i64 ngroups = (a.B + ppw - 1) / ppw;
    grid = ngroups * a.T;
    if (grid > 65536) grid = 65536;
    static bool attr_set = false;
    if (!attr_set) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_generic),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_budget);
      if (e != hipSuccess) return e;
      attr_set = true;
    }
    hipLaunchKernelGGL(k_generic, dim3((unsigned)grid), dim3(256), lds_bytes, a.stream, a.y, a.B, a.T, (int)a.n,
                       a.stages, a.nstages, a.consts, a.cpc, a.mod, ppw, scratch, ngroups);
  } else {
    if (!a.scratch) return hipErrorInvalidValue;
    grid = a.B * a.T;
    const i64 maxg = (i64)(a.scratch_bytes / (2 * (size_t)a.n * sizeof(u64)));
    if (grid > maxg) grid = maxg;
    if (grid < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_generic, dim3((unsigned)grid), dim3(256), 0, a.stream, a.y, a.B, a.T, (int)a.n,
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
