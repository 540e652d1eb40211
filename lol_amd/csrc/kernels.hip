// lol_amd/csrc/kernels.hip — hand-written gfx950 kernels for the Lol Tensor hot path.
//
// One polynomial (one RNS component of it) is owned by one workgroup, staged
// once into registers/LDS, taken through EVERY stage of the transform there and
// written back once: HBM sees 8 bytes in + 8 bytes out per coefficient, where the
// reference makes one full sweep over memory per stage per prime power
// (crt.cpp:459-538, tensor.h:76-95).
//
//   k_pow2<L,MODE>   m = 2^(L+1): negacyclic NTT (CRT), inverse, fused poly-mul
//   k_generic        any m: scalar interpreter of the plan's stage program — the fallback for
//                    primes > 13 or n > 8192 (the vector interpreter and the fused mixed-radix
//                    poly-mul are in mixed.hip)
//   k_pointwise_mul  mulRq (mul.cpp:14-30) and mulGCRT/divGCRT (CPP.hs:230-231)
//   k_gather / k_twace_crt   twace*/embed* (Extension.hs:54-129)
//
// Layout at the boundary is the reference's: y[(b*n + j)*T + t] (tensor.h:69).
#include "pow2_impl.h"

namespace lolhip {

// the arithmetic classes of the m = 2^k path live in pow2_ar{0,1,2,3}.hip
#define LOLHIP_POW2_EXTERN(AR) \
  extern template hipError_t launch_pow2_ar<AR, false>(const Pow2Launch&, int); \
  extern template hipError_t launch_pow2_ar<AR, true>(const Pow2Launch&, int);
LOLHIP_POW2_EXTERN(0) LOLHIP_POW2_EXTERN(1) LOLHIP_POW2_EXTERN(2) LOLHIP_POW2_EXTERN(3) LOLHIP_POW2_EXTERN(4)
#undef LOLHIP_POW2_EXTERN
template <int AR> static hipError_t launch_pow2_class(const Pow2Launch& a, int mode) {
  // large single-modulus 32-bit poly-mul batches: the persistent, DMA-pipelined kernel (pow2_pipe.hip)
  if (mode == 2 && !sw(SW_NO_PIPE) && pow2_pipe_ok(a, sw(SW_FORCE_PIPE))) return launch_pow2_pipe(a);
  // one modulus and 16-byte-aligned slabs: the variant that moves 16 bytes per lane
  const uintptr_t al = (uintptr_t)a.y | (mode == 2 ? ((uintptr_t)a.a | (uintptr_t)a.b) : 0);
  const bool t1 = a.T == 1 && (al & 15) == 0 && !pow2_no_t1();
  return t1 ? launch_pow2_ar<AR, true>(a, mode) : launch_pow2_ar<AR, false>(a, mode);
}
hipError_t launch_pow2(const Pow2Launch& a, int mode) {
  switch (a.arith) {
    case 0: return launch_pow2_class<0>(a, mode);
    case 1: return launch_pow2_class<1>(a, mode);
    case 2: return launch_pow2_class<2>(a, mode);
    case 3: return launch_pow2_class<3>(a, mode);
    case 4: return launch_pow2_class<4>(a, mode);
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
template <int L, int AR>
__global__ void __launch_bounds__(pow2_threads(L), 4)
k_keyswitch(const i64* __restrict__ c2, const i64* __restrict__ hint, const i64* addend, i64* out, i64 B, int T,
            const u32* __restrict__ tw_fwd, const ModCtx* __restrict__ mod, DecompParams dp, u32 magic32, int xcd_map) {
  static_assert(AR == 2 || AR == 4, "32-bit classes with [0,4q) / wide lazy forward ranges");
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
  const QK32 qk(ms, std::bool_constant<(NT >= 64)>{});
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
  tw.sc0 = 0; tw.sc1 = 0; tw.l1w = 0; tw.l1wp = 0;
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
        // hint residues may be any representative in (-q_s, q_s), like every other input
        const u32 h0 = from_i64<AR>((i64)load_u64(rh, off_h, hoff + eo), qk);
        const u32 h1 = from_i64<AR>((i64)load_u64(rh, off_h, hoff + hstride + eo), qk);
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

template <int L, int AR>
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
  if (lds_bytes > 64 * 1024) {
    static KernelDev tab[MAX_DEV];
    hipError_t e = kernel_dev_setup(tab, [&]() -> hipError_t {
      return hipFuncSetAttribute(reinterpret_cast<const void*>(&k_keyswitch<L, AR>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    });
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL((k_keyswitch<L, AR>), dim3((unsigned)grid), dim3(NT * PPW), lds_bytes, a.stream, a.c2, a.hint,
                     a.addend, a.out, a.B, a.T, a.tw_fwd32, a.mod, a.dp, a.magic32, xcd_map);
  return hipGetLastError();
}

template <int AR>
static hipError_t launch_keyswitch_ar(const KeySwitchLaunch& a) {
  switch (a.L) {
    case 4: return launch_keyswitch_L<4, AR>(a);
    case 5: return launch_keyswitch_L<5, AR>(a);
    case 6: return launch_keyswitch_L<6, AR>(a);
    case 7: return launch_keyswitch_L<7, AR>(a);
    case 8: return launch_keyswitch_L<8, AR>(a);
    case 9: return launch_keyswitch_L<9, AR>(a);
    case 10: return launch_keyswitch_L<10, AR>(a);
    case 11: return launch_keyswitch_L<11, AR>(a);
    case 12: return launch_keyswitch_L<12, AR>(a);
    case 13: return launch_keyswitch_L<13, AR>(a);
    case 14: return launch_keyswitch_L<14, AR>(a);
    default: return hipErrorInvalidValue;
  }
}
hipError_t launch_keyswitch_fused(const KeySwitchLaunch& a) {
  return a.arith == 4 ? launch_keyswitch_ar<4>(a) : launch_keyswitch_ar<2>(a);
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
    static KernelDev tab[MAX_DEV];
    hipError_t e = kernel_dev_setup(tab, [&]() -> hipError_t {
      hipError_t r = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_generic<true, false>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_budget);
      if (r == hipSuccess)
        r = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_generic<true, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_budget);
      return r;
    });
    if (e != hipSuccess) return e;
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

// ---- streaming kernels -------------------------------------------------------------------------
// One tile of consecutive elements per workgroup; the few divisions (which polynomial, which
// modulus) are done once per workgroup or in 32 bits — the first version divided 64-bit indices
// three times per element and ran at 0.29-0.39 of peak on config 5's embed/twace.

// a[i] = a[i] * b[i mod bperiod]  (bperiod = total length for mulRq, n*T for g vectors)
constexpr int PW_K = 8;                     // elements per thread
__global__ void __launch_bounds__(256)
k_pointwise_mul(i64* __restrict__ a, const i64* __restrict__ b, i64 total, i64 bperiod, int T,
                const ModCtx* __restrict__ mod) {
  const i64 s0 = (i64)blockIdx.x * (256 * PW_K);          // wave-uniform
  const bool flat = (bperiod >= total);
  const u32 t_s = (u32)((u64)s0 % (u32)T);
  const u32 r_s = flat ? 0u : (u32)((u64)s0 % (u64)bperiod);     // bperiod = n * T < 2^31 whenever it is not the whole array
  const u32 per = (u32)bperiod;
  u64 x[PW_K], z[PW_K];
#pragma unroll
  for (int k = 0; k < PW_K; ++k) {
    const u32 l = (u32)k * 256u + threadIdx.x;
    const i64 i = s0 + l;
    if (i < total) {
      x[k] = (u64)a[i];
      z[k] = (u64)b[flat ? i : (i64)((r_s + l) % per)];
    }
  }
#pragma unroll
  for (int k = 0; k < PW_K; ++k) {
    const u32 l = (u32)k * 256u + threadIdx.x;
    const i64 i = s0 + l;
    if (i < total) {
      const ModCtx mc = mod[T == 1 ? 0u : (t_s + l) % (u32)T];
      a[i] = (i64)mulmod(canon_in((i64)x[k], mc.q), canon_in((i64)z[k], mc.q), mc);
    }
  }
}

typedef u64 u64x2 __attribute__((ext_vector_type(2)));
// the same, two consecutive elements (16 bytes) per lane and access: even total and period, 16-byte-aligned
// slabs (whatever T: the two elements of a pair take their own modulus)
constexpr int PW2_K = 4;                    // pairs per thread
__global__ void __launch_bounds__(256)
k_pointwise_mul2(i64* __restrict__ a, const i64* __restrict__ b, i64 total, i64 bperiod, int T,
                 const ModCtx* __restrict__ mod) {
  const i64 s0 = (i64)blockIdx.x * (512 * PW2_K);         // first element of the tile (wave-uniform)
  const bool flat = (bperiod >= total);
  const u32 t_s = (u32)((u64)s0 % (u32)T);
  const u32 r_s = flat ? 0u : (u32)((u64)s0 % (u64)bperiod);
  const u32 per = (u32)bperiod;
  u64x2 x[PW2_K], z[PW2_K];
#pragma unroll
  for (int k = 0; k < PW2_K; ++k) {
    const u32 l = ((u32)k * 256u + threadIdx.x) * 2u;
    const i64 i = s0 + l;
    if (i < total) {
      x[k] = *reinterpret_cast<const u64x2*>(a + i);
      z[k] = *reinterpret_cast<const u64x2*>(b + (flat ? i : (i64)((r_s + l) % per)));
    }
  }
#pragma unroll
  for (int k = 0; k < PW2_K; ++k) {
    const u32 l = ((u32)k * 256u + threadIdx.x) * 2u;
    const i64 i = s0 + l;
    if (i < total) {
      const ModCtx m0 = mod[T == 1 ? 0u : (t_s + l) % (u32)T];
      const ModCtx m1 = mod[T == 1 ? 0u : (t_s + l + 1u) % (u32)T];
      u64x2 r;
      r.x = mulmod(canon_in((i64)x[k].x, m0.q), canon_in((i64)z[k].x, m0.q), m0);
      r.y = mulmod(canon_in((i64)x[k].y, m1.q), canon_in((i64)z[k].y, m1.q), m1);
      *reinterpret_cast<u64x2*>(a + i) = r;
    }
  }
}

hipError_t launch_pointwise_mul(hipStream_t s, i64* a, const i64* b, i64 total, i64 bperiod, int T, const ModCtx* mod) {
  if (total == 0) return hipSuccess;
  if ((((uintptr_t)a | (uintptr_t)b) & 15) == 0 && total % 2 == 0 && (bperiod >= total || bperiod % 2 == 0)) {
    const i64 blocks = (total + 512 * PW2_K - 1) / (512 * PW2_K);
    if (blocks > 0x7fffffff) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_pointwise_mul2, dim3((unsigned)blocks), dim3(256), 0, s, a, b, total, bperiod, T, mod);
    return hipGetLastError();
  }
  const i64 blocks = (total + 256 * PW_K - 1) / (256 * PW_K);
  if (blocks > 0x7fffffff) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_pointwise_mul, dim3((unsigned)blocks), dim3(256), 0, s, a, b, total, bperiod, T, mod);
  return hipGetLastError();
}

// ---- the yardstick: a read-once / write-once copy of a slab, 16 bytes per lane ----------------
// What a streaming kernel with no arithmetic reaches on this part; bench.py reports every HBM-bound
// leg beside it (MI355X_MICROARCH.md measures 6.29 TB/s of the 8 TB/s datasheet peak this way).
// variant 0: one 16 KiB tile per 256-thread workgroup, all loads issued before the stores;
// variant 1: 2048 persistent workgroups walking the slab with the same tile.
template <bool PERSIST>
__global__ void __launch_bounds__(256)
k_copy16(u32x4* __restrict__ dst, const u32x4* __restrict__ src, u64 nvec) {
  constexpr int K = 4;
  const u64 ntiles = (nvec + 256 * K - 1) / (256 * K);
  for (u64 tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const u64 base = tile * (256 * K) + threadIdx.x;
    u32x4 v[K];
#pragma unroll
    for (int k = 0; k < K; ++k) if (base + (u64)k * 256 < nvec) v[k] = src[base + (u64)k * 256];
#pragma unroll
    for (int k = 0; k < K; ++k) if (base + (u64)k * 256 < nvec) dst[base + (u64)k * 256] = v[k];
    if constexpr (!PERSIST) break;
  }
}
hipError_t launch_copy16(hipStream_t s, void* dst, const void* src, size_t bytes, int variant) {
  if (bytes == 0) return hipSuccess;
  if ((((uintptr_t)dst | (uintptr_t)src) & 15) || bytes % 16) return hipErrorInvalidValue;
  const u64 nvec = bytes / 16, ntiles = (nvec + 1023) / 1024;
  if (variant == 1) hipLaunchKernelGGL((k_copy16<true>), dim3((unsigned)(ntiles < 2048 ? ntiles : 2048)), dim3(256), 0, s, (u32x4*)dst, (const u32x4*)src, nvec);
  else if (ntiles > 0x7fffffff) return hipErrorInvalidValue;
  else hipLaunchKernelGGL((k_copy16<false>), dim3((unsigned)ntiles), dim3(256), 0, s, (u32x4*)dst, (const u32x4*)src, nvec);
  return hipGetLastError();
}

// TW consecutive components of one coefficient (16-byte accesses when T is even and the slabs are aligned)
template <int TW> struct Chunk { u64 v[TW]; };
template <int TW> __device__ __forceinline__ Chunk<TW> load_chunk(const i64* p) {
  Chunk<TW> c;
  if constexpr (TW == 2) { const u64x2 w = *reinterpret_cast<const u64x2*>(p); c.v[0] = w.x; c.v[1] = w.y; }
  else c.v[0] = (u64)*p;
  return c;
}
template <int TW> __device__ __forceinline__ void store_chunk(i64* p, const Chunk<TW>& c) {
  if constexpr (TW == 2) { u64x2 w; w.x = c.v[0]; w.y = c.v[1]; *reinterpret_cast<u64x2*>(p) = w; }
  else *p = (i64)c.v[0];
}
// tile bookkeeping shared by the gather-type kernels: chunk c of polynomial b -> (coefficient i, first component t0)
struct TileArgs { u32 tiles, cpt, cpt_magic; };     // cpt = T / TW chunks per coefficient; magic = floor(2^32/cpt)+1 (unused for cpt = 1)
static TileArgs tile_args(i64 n_out, int T, int TW, int K) {
  TileArgs t;
  t.cpt = (u32)(T / TW);
  t.cpt_magic = t.cpt > 1 ? (u32)((((u64)1 << 32) / t.cpt) + 1) : 0;
  const u64 chunks = (u64)n_out * t.cpt;
  t.tiles = (u32)((chunks + 256 * K - 1) / (256 * K));
  return t;
}
static bool aligned16(const void* a, const void* b) { return (((uintptr_t)a | (uintptr_t)b) & 15) == 0; }

// out[b][i][t] = +-in[b][idx[i]][t] or 0  (embedPow/Dec/CRT, twacePowDec; Extension.hs:54-101)
constexpr int GA_K = 4;
template <int TW>
__global__ void __launch_bounds__(256)
k_gather(i64* __restrict__ out, const i64* __restrict__ in, const int32_t* __restrict__ idx, u32 n_out, u32 n_in, int T,
         const ModCtx* __restrict__ mod, TileArgs ta) {
  const u32 b = blockIdx.x / ta.tiles, tile = blockIdx.x - b * ta.tiles;          // wave-uniform
  const i64* src = in + (size_t)b * n_in * T;
  i64* dst = out + (size_t)b * n_out * T;
  const u32 nchunks = n_out * ta.cpt;
  int32_t e[GA_K];
  u32 t0[GA_K];
  Chunk<TW> v[GA_K];
#pragma unroll
  for (int k = 0; k < GA_K; ++k) {
    const u32 c = (tile * GA_K + k) * 256u + threadIdx.x;
    const u32 i = ta.cpt == 1 ? c : __umulhi(c, ta.cpt_magic);
    t0[k] = (c - i * ta.cpt) * TW;
    e[k] = c < nchunks ? idx[i] : -1;
  }
#pragma unroll
  for (int k = 0; k < GA_K; ++k) {
    if (e[k] >= 0) v[k] = load_chunk<TW>(src + (size_t)(e[k] & (EMBED_NEG_FLAG_DEV - 1)) * T + t0[k]);
    else for (int j = 0; j < TW; ++j) v[k].v[j] = 0;
  }
#pragma unroll
  for (int k = 0; k < GA_K; ++k) {
    const u32 c = (tile * GA_K + k) * 256u + threadIdx.x;
    if (c >= nchunks) continue;
    if (e[k] >= 0) {
#pragma unroll
      for (int j = 0; j < TW; ++j) {
        const u64 q = mod[t0[k] + j].q;
        const u64 x = canon_in((i64)v[k].v[j], q);
        v[k].v[j] = (e[k] & EMBED_NEG_FLAG_DEV) ? (x == 0 ? 0 : q - x) : x;
      }
    }
    store_chunk<TW>(dst + (size_t)c * TW, v[k]);
  }
}

// The same for a REPLICATING gather whose source polynomial fits LDS (embedCRT, n_out >= 2 n_in; embedPow/Dec
// write mostly zeros and are faster without the staging): the
// source is read from HBM once, coalesced, and the scattered reads hit LDS instead of the vector
// L1 (embedCRT 2048 -> 14336 replicates every coefficient 6 times: 0.47 of peak through L1, 0.54 through LDS).
template <int TW>
__global__ void __launch_bounds__(256)
k_gather_lds(i64* __restrict__ out, const i64* __restrict__ in, const int32_t* __restrict__ idx, u32 n_out, u32 n_in, int T,
             const ModCtx* __restrict__ mod, TileArgs ta, u32 split, u32 cps) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  i64* sm = reinterpret_cast<i64*>(smem);
  const u32 b = blockIdx.x / split, part = blockIdx.x - b * split;               // wave-uniform
  const i64* src = in + (size_t)b * n_in * T;
  i64* dst = out + (size_t)b * n_out * T;
  const u32 nsrc = n_in * (u32)T;
  for (u32 x = threadIdx.x * TW; x < nsrc; x += 256u * TW) store_chunk<TW>(sm + x, load_chunk<TW>(src + x));
  __syncthreads();
  const u32 nchunks = n_out * ta.cpt;
  const u32 c_end = min(nchunks, (part + 1) * cps);
  for (u32 c0 = part * cps; c0 < c_end; c0 += 256u * GA_K) {
    int32_t e[GA_K];
    u32 t0[GA_K];
#pragma unroll
    for (int k = 0; k < GA_K; ++k) {
      const u32 c = c0 + (u32)k * 256u + threadIdx.x;
      const u32 i = ta.cpt == 1 ? c : __umulhi(c, ta.cpt_magic);
      t0[k] = (c - i * ta.cpt) * TW;
      e[k] = c < c_end ? idx[i] : -1;
    }
#pragma unroll
    for (int k = 0; k < GA_K; ++k) {
      const u32 c = c0 + (u32)k * 256u + threadIdx.x;
      if (c >= c_end) continue;
      Chunk<TW> v;
      if (e[k] >= 0) {
        v = load_chunk<TW>(sm + (size_t)(e[k] & (EMBED_NEG_FLAG_DEV - 1)) * T + t0[k]);
#pragma unroll
        for (int j = 0; j < TW; ++j) {
          const u64 q = mod[t0[k] + j].q;
          const u64 x = canon_in((i64)v.v[j], q);
          v.v[j] = (e[k] & EMBED_NEG_FLAG_DEV) ? (x == 0 ? 0 : q - x) : x;
        }
      } else {
#pragma unroll
        for (int j = 0; j < TW; ++j) v.v[j] = 0;
      }
      store_chunk<TW>(dst + (size_t)c * TW, v);
    }
  }
}

hipError_t launch_gather(hipStream_t s, i64* out, const i64* in, const int32_t* idx, i64 B, i64 n_out, i64 n_in,
                         int T, const ModCtx* mod, bool replicating) {
  if (B * n_out * T == 0) return hipSuccess;
  const int TW = (T % 2 == 0 && aligned16(out, in)) ? 2 : 1;
  const size_t src_bytes = (size_t)n_in * T * 8;
  if (replicating && src_bytes <= 64 * 1024 && n_out >= 2 * n_in) {
    const TileArgs ta = tile_args(n_out, T, TW, GA_K);
    // one workgroup copies the source once: give it at least ~4 source sizes of output to write
    const u64 nchunks = (u64)n_out * ta.cpt;
    u64 cps = (u64)4 * n_in * ta.cpt;
    cps = (cps + 256 * GA_K - 1) / (256 * GA_K) * (256 * GA_K);
    if (cps > nchunks) cps = (nchunks + 256 * GA_K - 1) / (256 * GA_K) * (256 * GA_K);
    const u32 split = (u32)((nchunks + cps - 1) / cps);
    const i64 blocks = B * split;
    if (blocks > 0x7fffffff) return hipErrorInvalidValue;
    if (TW == 2) hipLaunchKernelGGL(k_gather_lds<2>, dim3((unsigned)blocks), dim3(256), src_bytes, s, out, in, idx, (u32)n_out, (u32)n_in, T, mod, ta, split, (u32)cps);
    else hipLaunchKernelGGL(k_gather_lds<1>, dim3((unsigned)blocks), dim3(256), src_bytes, s, out, in, idx, (u32)n_out, (u32)n_in, T, mod, ta, split, (u32)cps);
    return hipGetLastError();
  }
  const TileArgs ta = tile_args(n_out, T, TW, GA_K);
  const i64 blocks = B * ta.tiles;
  if (blocks > 0x7fffffff) return hipErrorInvalidValue;
  if (TW == 2) hipLaunchKernelGGL(k_gather<2>, dim3((unsigned)blocks), dim3(256), 0, s, out, in, idx, (u32)n_out, (u32)n_in, T, mod, ta);
  else hipLaunchKernelGGL(k_gather<1>, dim3((unsigned)blocks), dim3(256), 0, s, out, in, idx, (u32)n_out, (u32)n_in, T, mod, ta);
  return hipGetLastError();
}

// twaceCRT (Extension.hs:110-129): out[b][i][t] = sum_{r<rel} tweak[e]*in[b][e][t], e = idx[i*rel + r]
constexpr int TC_K = 2;
template <int TW>
__global__ void __launch_bounds__(256)
k_twace_crt(i64* __restrict__ out, const i64* __restrict__ in, const int32_t* __restrict__ idx,
            const i64* __restrict__ tweak, u32 n_out, u32 n_in, int T, const ModCtx* __restrict__ mod, TileArgs ta) {
  const u32 b = blockIdx.x / ta.tiles, tile = blockIdx.x - b * ta.tiles;
  const i64* src = in + (size_t)b * n_in * T;
  i64* dst = out + (size_t)b * n_out * T;
  const u32 nchunks = n_out * ta.cpt;
  const u32 rel = n_in / n_out;
#pragma unroll
  for (int k = 0; k < TC_K; ++k) {
    const u32 c = (tile * TC_K + k) * 256u + threadIdx.x;
    if (c >= nchunks) continue;
    const u32 i = ta.cpt == 1 ? c : __umulhi(c, ta.cpt_magic);
    const u32 t0 = (c - i * ta.cpt) * TW;
    ModCtx mc[TW];
#pragma unroll
    for (int j = 0; j < TW; ++j) mc[j] = mod[t0 + j];
    Chunk<TW> acc;
#pragma unroll
    for (int j = 0; j < TW; ++j) acc.v[j] = 0;
    const int32_t* ip = idx + (size_t)i * rel;
    for (u32 r = 0; r < rel; ++r) {
      const size_t eo = (size_t)ip[r] * T + t0;
      const Chunk<TW> x = load_chunk<TW>(src + eo), w = load_chunk<TW>(tweak + eo);
#pragma unroll
      for (int j = 0; j < TW; ++j) acc.v[j] = addmod(acc.v[j], mulmod(canon_in((i64)x.v[j], mc[j].q), w.v[j], mc[j]), mc[j].q);
    }
    store_chunk<TW>(dst + (size_t)c * TW, acc);
  }
}

hipError_t launch_twace_crt(hipStream_t s, i64* out, const i64* in, const int32_t* idx, const i64* tweak, i64 B,
                            i64 n_out, i64 n_in, int T, const ModCtx* mod) {
  if (B * n_out * T == 0) return hipSuccess;
  const int TW = (T % 2 == 0 && aligned16(out, in) && aligned16(tweak, tweak)) ? 2 : 1;
  const TileArgs ta = tile_args(n_out, T, TW, TC_K);
  const i64 blocks = B * ta.tiles;
  if (blocks > 0x7fffffff) return hipErrorInvalidValue;
  if (TW == 2) hipLaunchKernelGGL(k_twace_crt<2>, dim3((unsigned)blocks), dim3(256), 0, s, out, in, idx, tweak, (u32)n_out, (u32)n_in, T, mod, ta);
  else hipLaunchKernelGGL(k_twace_crt<1>, dim3((unsigned)blocks), dim3(256), 0, s, out, in, idx, tweak, (u32)n_out, (u32)n_in, T, mod, ta);
  return hipGetLastError();
}

}  // namespace lolhip
