// lol_amd/csrc/mixed_impl.h — device templates of the mixed-radix path (instantiated once per
// arithmetic class in mixed_cls{0,1,2,3}.hip so the classes compile in parallel; dispatcher in mixed.hip):
// the mixed-radix (any m with every prime <= 13, n <= 8192) path:
// vector-per-thread interpreter of the plan's stage program, and the FUSED mixed-radix
// poly-mul c = crtInv(crt a * crt b) in one launch.
//
// What the reference does per prime power with one sweep over memory per stage
// (crt.cpp:459-538 through tensor.h:76-95; dense p-point forms crt.cpp:226-245,324-345,433-456;
// L/G forms l.cpp:28-98, g.cpp:16-123) runs here with one polynomial component per workgroup
// resident in LDS: every stage (I (x) A_p (x) I_rts) is applied by threads that each own one
// d-vector, read it from LDS once, apply A_p in registers and write it back in place.
//
// Against the first version of this kernel (k_generic_vec, round 1):
//   * global loads are issued in one batch of up to 16 per thread (the per-element load loop
//     put ONE load per wave in flight: ~11 serial HBM round trips per polynomial of m = 15015);
//   * three arithmetic/storage classes chosen per plan on the host:
//       CLS 0  any q < 2^62:  64-bit residues in LDS, 128-bit dot-product accumulators;
//       CLS 1  every q < 2^32: 32-bit residues in LDS and registers (half the LDS traffic and
//              footprint), 32x32 products accumulated in 128 bits;
//       CLS 3  64-bit residues, every q odd: as CLS 0 with the constants pre-scaled by 2^64 and ONE
//              Montgomery reduction per dot product (~30 instructions against ~70);
//       CLS 2  every q odd with 13 (q-1)^2 < 2^64: the whole dot product accumulates in ONE 64-bit
//              v_mad_u64_u32 chain and is reduced once by a 9-instruction 32-bit Montgomery step
//              (constants pre-scaled by 2^32) — the reference's own domain (q < 2^31.5,
//              types.h:79-84) up to q < 2^30.15;
//   * the poly-mul is ONE launch: a-hat waits in registers (each thread keeps the coefficients
//     it loaded) while b goes through the same LDS buffer; b's global loads are issued before
//     a's stages and land under them; HBM sees a and b once in and c once out (round 1: four
//     launches through a plan-owned temp, 9 slab passes).
#pragma once
#include "pow2_impl.h"     // buffer-descriptor loads/stores, fresh()

namespace lolhip {

#ifndef LOLHIP_MIXED_W2
#define LOLHIP_MIXED_W2 6      // waves/SIMD bound of the 32-bit single-program kernel (A/B with the 32-bit pool: 6 beats 8, which spills to scratch)
#endif
// coefficients a thread loads/stores: ppw * n <= KMAX * blockDim, KMAX in {12, 16} picked by the launcher
// (m = 15015: 5760 / 512 = 11.25 -> 12: a quarter fewer predicated load/store/convert slots than 16)

template <int CLS> using MV = std::conditional_t<CLS == 0 || CLS == 3, u64, u32>;
template <int CLS> constexpr bool wide() { return CLS == 0 || CLS == 3; }
// element type of the constant pool the kernel reads: the 32-bit classes get a 32-bit copy (half the
// scalar-load bytes and SGPRs of a matrix row, half the bytes of a per-lane twiddle load)
#ifndef LOLHIP_MIXED_HOIST0
#define LOLHIP_MIXED_HOIST0 0     // hoisted dispatch in the single-program kernels too (A/B)
#endif
#ifndef LOLHIP_MIXED_POOL32
#define LOLHIP_MIXED_POOL32 1
#endif
template <int CLS> using PT = std::conditional_t<(CLS == 1 || CLS == 2 || CLS == 4) && LOLHIP_MIXED_POOL32, u32, u64>;
// CLS 4 = class 2 for moduli below 2^27 with LAZY dense stages: a dot product of up to 16 terms of values below 2q has
// T / 2^32 < q, so the Montgomery step's result h + ceil(m q / 2^32) is below 2q as it stands — 3 instructions instead
// of the 9 of redc64 (no pre-subtraction, no canonicalising min-subtractions).  Values in [0, 2q) travel between dense
// stages and diagonals; the 2-power tiles, the additive maps (L, G) and the store canonicalise / stay canonical.
template <int CLS> constexpr bool c2() { return CLS == 2 || CLS == 4; }

// x / v for x <= 8192 (every index here is below ppw * n <= 8192) from the plan's 2^40-scaled
// reciprocal M = floor(2^40/v)+1: (M >> 8) + 1 is floor(2^32/v) + 1 or + 2, whose error times x
// stays below 2^27, so ONE v_mul_hi_u32 is exact (the scalar part is wave-uniform SALU work)
__device__ __forceinline__ int mdiv(int x, u64 M) {
  return (M >> 40) ? x : (int)__umulhi((u32)x, (u32)(M >> 8) + 1u);      // M >> 40 is set only for v = 1
}

template <int CLS> __device__ __forceinline__ MV<CLS> m_add(MV<CLS> a, MV<CLS> b, u64 q) {
  if constexpr (wide<CLS>()) return addmod(a, b, q);
  else if constexpr (c2<CLS>()) { const u32 s = a + b; return min(s, s - (u32)q); }   // q < 2^30.2: the sum fits a word
  else { const u64 s = (u64)a + b; return (u32)(s >= q ? s - q : s); }          // q may exceed 2^31: 33-bit sum
}
template <int CLS> __device__ __forceinline__ MV<CLS> m_sub(MV<CLS> a, MV<CLS> b, u64 q) {
  if constexpr (wide<CLS>()) return submod(a, b, q);
  else if constexpr (c2<CLS>()) { const u32 d = a - b; return min(d, d + (u32)q); }   // a < b: d wraps high, d + q is the residue
  else return a >= b ? a - b : (u32)(a + (u32)q - b);
}
// x mod q for a 64-bit x: Barrett with mu = floor(2^64 / q)
__device__ __forceinline__ u32 barrett64(u64 x, const ModCtx& mc) {
  const u64 Q = __umul64hi(x, mc.mu);             // floor(x/q) or one less
  u64 r = x - Q * mc.q;                            // [0, 2q)
  if (r >= mc.q) r -= mc.q;
  return (u32)r;
}
// Montgomery reduction of a 128-bit value T < 13 q^2 with q < 2^61 (so T / 2^64 < 13 q / 8):
//   (T + ((T mod 2^64) * (-q^-1) mod 2^64) * q) / 2^64  =  T * 2^-64 mod q,  below 2.625 q < 2^63;
// two conditional subtractions make it canonical.  ~25 instructions against ~70 for the exact
// two-step 128-by-64 division; the 2^-64 is absorbed by constants pre-scaled on the host.
__device__ __forceinline__ u64 redc128(unsigned __int128 T, const ModCtx& mc) {
  const u64 lo = (u64)T, hi = (u64)(T >> 64);
  const u64 m = lo * mc.nqinv;
  u64 r = hi + __umul64hi(m, mc.q) + (lo != 0);      // lo + lo64(m q) = 0 mod 2^64: it carries iff lo != 0
  r = csub(r, 2 * mc.q);
  return csub(r, mc.q);
}
// 32-bit Montgomery reduction of T < 13 q^2 < 2^64 (class 2; q odd, q < 2^30.15):
//   h = T >> 32 < 3.6 q fits a word; h >= 2q ? h - 2q : h changes T by a multiple of q and leaves h < 2q;
//   m = (T mod 2^32) * (-q^-1) mod 2^32;  T + m q = 0 mod 2^32, and (T mod 2^32) + (m q mod 2^32)
//   carries iff m != 0, so the carry folds into the high word of m q + (2^32 - 1): ONE v_mad_u64_u32;
//   result h + ceil(m q / 2^32) < 3 q + 1 < 2^32 = T 2^-32 mod q, two min-subtractions to canonical.
// 9 instructions against ~25 for a 64-bit Barrett step; the 2^-32 is absorbed by constants
// pre-scaled on the host.
__device__ __forceinline__ u32 redc64(u64 T, const ModCtx& mc) {
  const u32 q = (u32)mc.q;
  u32 h = (u32)(T >> 32);
  h = min(h, h - 2 * q);
  const u32 m = (u32)T * (u32)mc.nqinv;
  u32 r = h + (u32)(((u64)m * q + 0xFFFFFFFFull) >> 32);
  r = min(r, r - 2 * q);
  return min(r, r - q);
}
// The same reductions for ONE product T = a b < q^2 of canonical residues (twiddle diagonals, the 2-power
// tiles): the high word is already small (class 2: T / 2^32 < q / 3.6; class 3: T / 2^64 < q / 8), so the
// result is below 2q without the pre-subtractions and one conditional subtraction makes it canonical.
__device__ __forceinline__ u64 redc128_1(unsigned __int128 T, const ModCtx& mc) {
  const u64 lo = (u64)T, hi = (u64)(T >> 64);
  const u64 m = lo * mc.nqinv;
  return csub(hi + __umul64hi(m, mc.q) + (lo != 0), mc.q);
}
__device__ __forceinline__ u32 redc64_1(u64 T, const ModCtx& mc) {
  const u32 q = (u32)mc.q;
  const u32 m = (u32)T * (u32)mc.nqinv;
  const u32 r = (u32)(T >> 32) + (u32)(((u64)m * q + 0xFFFFFFFFull) >> 32);
  return min(r, r - q);
}
// class 4: the bare Montgomery step.  T < 16 (2q) q with q < 2^27: T / 2^32 < q, result < 2q; same congruence as redc64
__device__ __forceinline__ u32 redc64_lazy(u64 T, const ModCtx& mc) {
  const u32 q = (u32)mc.q;
  const u32 m = (u32)T * (u32)mc.nqinv;
  return (u32)(T >> 32) + (u32)(((u64)m * q + 0xFFFFFFFFull) >> 32);
}
// a * b mod q for canonical a.  KPOOL: b comes from the constant pool (pre-scaled by 2^32 / 2^64 in
// classes 2 / 3); otherwise b is a plain residue or small integer and those classes multiply exactly.
template <int CLS, bool KPOOL = true> __device__ __forceinline__ MV<CLS> m_mul(MV<CLS> a, u64 b, const ModCtx& mc) {
  if constexpr (CLS == 3 && KPOOL) return redc128_1((unsigned __int128)a * b, mc);
  else if constexpr (CLS == 2 && KPOOL) return redc64_1((u64)a * (u32)b, mc);
  else if constexpr (CLS == 4 && KPOOL) return redc64_lazy((u64)a * (u32)b, mc);      // a < 2q, b < q: T / 2^32 < q / 16, result < 2q
  else if constexpr (wide<CLS>()) return mulmod(a, b, mc);
  else return barrett64((u64)a * (u32)b, mc);
}

// one output of a dense stage: sum_c v[c] * row[c] mod q
template <int CLS, int D>
__device__ __forceinline__ MV<CLS> m_dot(const MV<CLS> (&v)[D], const PT<CLS>* __restrict__ row, const ModCtx& mc) {
  static_assert(D <= 16 || c2<CLS>(), "16 products below 2^124 fit in 128 bits");
  if constexpr (c2<CLS>()) {      // D <= 13, or the merged prime powers (D = 18, 20) of plans whose moduli leave the room (plan.cpp)
    u64 acc = 0;
#pragma unroll
    for (int c = 0; c < D; ++c) acc += (u64)v[c] * (u32)row[c];      // one v_mad_u64_u32 per term; D (q-1)^2 < 2^64
    if constexpr (CLS == 4) {
      const u32 r = redc64_lazy(acc, mc);             // D <= 16: < 2q;  D = 18, 20: T / 2^32 < 1.25 q, r < 2.25 q
      if constexpr (D > 16) return min(r, r - 2 * (u32)mc.q); else return r;
    } else
    return redc64(acc, mc);
  } else {
    unsigned __int128 acc = 0;
#pragma unroll
    for (int c = 0; c < D; ++c) {
      if constexpr (CLS == 1) acc += (unsigned __int128)((u64)v[c] * (u32)row[c]);
      else acc += (unsigned __int128)v[c] * row[c];          // classes 0 and 3: 13 products below 2^124
    }
    if constexpr (CLS == 3) return redc128(acc, mc);
    if (CLS == 1 && mc.q > 16) return (MV<CLS>)rem128((u64)(acc >> 64), (u64)acc, mc);   // high word < 16 < q: one division step
    return (MV<CLS>)reduce128((u64)(acc >> 64), (u64)acc, mc);
  }
}

// Class 3 (64-bit residues: one multiply-accumulate is ~8 instructions): DFT_p, CRT_p and the scaled CRT_p^-1 of an odd
// prime in their even/odd form.  With w = omega_p, h = (p-1)/2, e_c = x_c + x_(p-c), o_c = x_c - x_(p-c) (c = 1..h; a
// position the vector does not have counts as 0):
//   sum_c x_c w^(rc)  =  x_0 + S_r + A_r,   sum_c x_c w^(-rc)  =  x_0 + S_r - A_r,
//   S_r = sum_c e_c (w^(rc) + w^(-rc))/2,   A_r = sum_c o_c (w^(rc) - w^(-rc))/2          (r = 1..h)
// so rows r and p-r share their products: 2 h^2 multiply-accumulates instead of (p-1)^2 (72 instead of 144 at p = 13),
// the same number of Montgomery reductions, and ~3 modular additions per output.  Exact arithmetic in Z_q: the same
// residues as the dense form (crt.cpp:226-245, 324-345, 433-456).  Tables: plan.cpp (Stage::pad[0]); e is canonical
// where it is also summed, o = x + q - y stays below 2q: every accumulator is below 2 h q^2 <= 12 q^2 (redc128: < 13 q^2).
template <int D>
__device__ __forceinline__ void apply_eo(const Stage& st, const u64 (&v)[D], u64 (&o)[D], const u64* __restrict__ cst, const ModCtx& mc) {
  constexpr bool DFT = (D & 1) != 0;
  constexpr int P = DFT ? D : D + 1, H = (P - 1) / 2;
  const u64 q = mc.q;
  const u64* Cs = cst + st.pad[0];
  const u64* Sn = Cs + H * H;
  const bool inv = !DFT && st.kind == ST_CRTPINV;
  // x_k for k = 0..P-1 in terms of the vector: DFT x_k = v[k]; CRT_p x_k = v[k], x_(P-1) = 0; CRT_p^-1 x_k = v[k-1], x_0 = 0
  u64 e[H + 1], od[H + 1];
#pragma unroll
  for (int c = 1; c <= H; ++c) {
    u64 a, b;
    bool lone = false;                              // CRT_p: x_(P-1) does not exist
    if (DFT) { a = v[c]; b = v[P - c]; }
    else if (inv) { a = v[c - 1]; b = v[P - c - 1]; }
    else { a = v[c]; lone = (c == 1); b = lone ? 0 : v[P - c]; }
    e[c] = lone ? a : addmod(a, b, q);
    od[c] = lone ? a : a + (q - b);
  }
  // one row pair at a time, its outputs finished before the next pair's products start (no S/A arrays: the fused
  // poly-mul of this class has ~64 VGPRs to spare)
  auto pair = [&](int r, u64& S, u64& A) {
    unsigned __int128 as = 0, aa = 0;
#pragma unroll
    for (int c = 1; c <= H; ++c) {
      as += (unsigned __int128)e[c] * Cs[(r - 1) * H + (c - 1)];
      aa += (unsigned __int128)od[c] * Sn[(r - 1) * H + (c - 1)];
    }
    S = redc128(as, mc);
    A = redc128(aa, mc);
  };
  if (!inv) {
    const u64 x0 = v[0];
    if constexpr (DFT) {
      u64 s0 = x0;
#pragma unroll
      for (int c = 1; c <= H; ++c) s0 = addmod(s0, e[c], q);
      o[0] = s0;
    }
    constexpr int SH = DFT ? 0 : 1;                 // CRT_p: output i is row i + 1
#pragma unroll
    for (int r = 1; r <= H; ++r) {
      u64 S, A;
      pair(r, S, A);
      const u64 t = addmod(S, x0, q);
      o[r - SH] = addmod(t, A, q);
      o[P - r - SH] = submod(t, A, q);
    }
  } else {
    // M[i][c] = w^(i (c+1)) - w^(-(c+1)): row i of the transform minus C = sum_k x_k w^(-k) = S_1 - A_1 (row p-1)
    u64 s0 = 0;
#pragma unroll
    for (int c = 1; c <= H; ++c) s0 = addmod(s0, e[c], q);
    u64 C = 0;
#pragma unroll
    for (int i = 1; i <= H; ++i) {
      u64 S, A;
      pair(i, S, A);
      if (i == 1) { C = submod(S, A, q); o[0] = submod(s0, C, q); }
      const u64 t = submod(S, C, q);
      o[i] = addmod(t, A, q);
      if (i >= 2) o[P - i] = submod(t, A, q);
    }
  }
}

// the linear map of one stage on one d-vector in registers: o = A v
template <int CLS, int D>
__device__ __forceinline__ void apply_kind(const Stage& st, const MV<CLS> (&v)[D], MV<CLS> (&o)[D],
                                           const PT<CLS>* __restrict__ cst, const ModCtx& mc) {
  using V = MV<CLS>;
  const u64 q = mc.q;
  // vectors of odd length p only occur in DFT_p stages; the L and G maps act on p-1 elements
  // (and small multipliers such as p-1-i are below q: the host sends moduli <= 16 elsewhere)
  switch ((D & 1) ? (int)ST_DFTP : st.kind) {
    case ST_DFTP:
    case ST_CRTP:
    case ST_CRTPINV: {
#ifndef LH_NO_EO
      if constexpr (CLS == 3 && D >= 3 && D <= 13) { apply_eo<D>(st, v, o, cst, mc); break; }      // every class-3 plan carries the tables (plan.cpp finish)
#endif
      const PT<CLS>* M = cst + st.mat_off;
#pragma unroll
      for (int i = 0; i < D; ++i) o[i] = m_dot<CLS, D>(v, M + i * D, mc);
      break;
    }
    case ST_L: {                         // prefix sums (l.cpp:28-57)
      V s = 0;
#pragma unroll
      for (int i = 0; i < D; ++i) { s = m_add<CLS>(s, v[i], q); o[i] = s; }
      break;
    }
    case ST_LINV: {                      // adjacent differences (l.cpp:67-98)
      o[0] = v[0];
#pragma unroll
      for (int i = 1; i < D; ++i) o[i] = m_sub<CLS>(v[i], v[i - 1], q);
      break;
    }
    case ST_GPOW: {                      // g.cpp:16-35
      const V last = v[D - 1];
      o[0] = m_add<CLS>(v[0], last, q);
#pragma unroll
      for (int i = 1; i < D; ++i) o[i] = m_sub<CLS>(m_add<CLS>(v[i], last, q), v[i - 1], q);
      break;
    }
    case ST_GDEC: {                      // g.cpp:37-58
      V s = v[0];
#pragma unroll
      for (int c = 0; c < D; ++c) s = m_add<CLS>(s, v[c], q);
      o[0] = s;
#pragma unroll
      for (int i = 1; i < D; ++i) o[i] = m_sub<CLS>(v[i], v[i - 1], q);
      break;
    }
    case ST_GINVPOW: {                   // g.cpp:60-90: (p-1-i) * sum_{c<=i} - (i+1) * sum_{c>i}
      V tot = 0;
#pragma unroll
      for (int c = 0; c < D; ++c) tot = m_add<CLS>(tot, v[c], q);
      V le = 0;
#pragma unroll
      for (int i = 0; i < D; ++i) {
        le = m_add<CLS>(le, v[i], q);
        const V re = m_sub<CLS>(tot, le, q);
        o[i] = m_sub<CLS>(m_mul<CLS, false>(le, (u64)(st.p - 1 - i), mc), m_mul<CLS, false>(re, (u64)(i + 1), mc), q);
      }
      break;
    }
    case ST_GINVDEC: {                   // g.cpp:92-123: sum_c (c+1) v_c - p * sum_{c>i} v_c
      V s = 0, tot = 0;
#pragma unroll
      for (int c = 0; c < D; ++c) {
        s = m_add<CLS>(s, m_mul<CLS, false>(v[c], (u64)(c + 1), mc), q);
        tot = m_add<CLS>(tot, v[c], q);
      }
      const u64 pm = (u64)st.p;
      V le = 0;
#pragma unroll
      for (int i = 0; i < D; ++i) {
        le = m_add<CLS>(le, v[i], q);
        o[i] = m_sub<CLS>(s, m_mul<CLS, false>(m_sub<CLS>(tot, le, q), pm, mc), q);
      }
      break;
    }
    default:
#pragma unroll
      for (int i = 0; i < D; ++i) o[i] = v[i];
  }
}

// one d-vector of one stage, in place in LDS
template <int CLS, int D>
__device__ __forceinline__ void stage_vec(const Stage& st, MV<CLS>* __restrict__ buf, int vec, int n, u64 n_magic,
                                          const PT<CLS>* __restrict__ cst, const ModCtx& mc) {
  using V = MV<CLS>;
  const int rts = st.rts;
  const int blk = mdiv(vec, st.m_rts), r = vec - blk * rts;
  const int x0 = blk * D * rts + r;
  V* base = buf + x0;
  V v[D], o[D];
#pragma unroll
  for (int i = 0; i < D; ++i) v[i] = base[i * rts];
  apply_kind<CLS, D>(st, v, o, cst, mc);
  if (st.tw_off >= 0) {                  // the diagonal folded into this stage (crtTwiddle/dftTwiddle, mhat^-1, oddRad^-1)
    if (st.tw_per > 0) {                 // this stage's own diagonal (forward programs): consecutive entries, one division per vector
      const PT<CLS>* tw = cst + st.tw_off + (blk - mdiv(blk, st.m_twper) * st.tw_per) * D;
#pragma unroll
      for (int i = 0; i < D; ++i) o[i] = m_mul<CLS>(o[i], tw[i], mc);
    } else {
      const int pi = mdiv(x0, n_magic);
      const int xi0 = x0 - pi * n;
#pragma unroll
      for (int i = 0; i < D; ++i) {
        const int xd = mdiv(xi0 + i * rts, st.m_twdiv);
        o[i] = m_mul<CLS>(o[i], cst[st.tw_off + xd - mdiv(xd, st.m_twmod) * st.tw_mod], mc);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < D; ++i) base[i * rts] = o[i];
}

// a merged prime power (plan.cpp merge_stages: D = 18, 20; no diagonal): every output is stored as soon as
// its dot product is reduced — the D inputs are in registers by then — so D + 3 values are live, not 2 D
template <int CLS, int D>
__device__ __forceinline__ void stage_vec_big(const Stage& st, MV<CLS>* __restrict__ buf, int vec,
                                              const PT<CLS>* __restrict__ cst, const ModCtx& mc) {
  using V = MV<CLS>;
  const int rts = st.rts;
  const int blk = mdiv(vec, st.m_rts), r = vec - blk * rts;
  V* base = buf + (blk * D * rts + r);
  V v[D];
#pragma unroll
  for (int i = 0; i < D; ++i) v[i] = base[i * rts];
  // (opaque per vector: hoisted out of the loop over vectors, the D x D matrix would be 400 SGPRs spilled to VGPR lanes)
  int z = 0;
  asm volatile("" : "+v"(z));
  const PT<CLS>* M = cst + __builtin_amdgcn_readfirstlane(st.mat_off + z);
#pragma unroll
  for (int i = 0; i < D; ++i) {
    base[i * rts] = m_dot<CLS, D>(v, M + i * D, mc);
    if (i & 1) __builtin_amdgcn_sched_barrier(0);       // two rows (40 SGPRs) in flight, not the whole matrix
  }
}

// FOUR adjacent d-vectors (positions r .. r+3 of one block) per thread, for stages whose stride is a
// multiple of 4: the four share the index arithmetic, the matrix rows in SGPRs and — because the
// diagonal's index is x / tw_div with tw_div a multiple of 4 (or the diagonal is a single constant) —
// the twiddle of every row; LDS is accessed 16 bytes at a time.  Cuts the per-coefficient overhead of
// the short dense vectors (d <= 7: the CRT_p / DFT_p stages of 3^e, 5^e, 7^e) of prime-power indices, where it
// exceeds the arithmetic.  (The L and G maps stay one vector per thread.)
template <int CLS, int D, int W>
__device__ __forceinline__ void stage_vecw(const Stage& st, MV<CLS>* __restrict__ buf, int vq, int n, u64 n_magic,
                                           const PT<CLS>* __restrict__ cst, const ModCtx& mc) {
  using V = MV<CLS>;
  constexpr int BYTES = W * (int)sizeof(V) < 16 ? W * (int)sizeof(V) : 16;
  constexpr int PER = BYTES / (int)sizeof(V), NCH = W / PER;     // LDS chunks of up to 16 bytes per W coefficients
  typedef V VV __attribute__((ext_vector_type(PER)));
  const int rts = st.rts, vec = vq * W;
  const int blk = mdiv(vec, st.m_rts), r = vec - blk * rts;
  const int x0 = blk * D * rts + r;
  V* base = buf + x0;
  V v[W][D];
#pragma unroll
  for (int i = 0; i < D; ++i) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const VV w = reinterpret_cast<const VV*>(base + i * rts)[c];
#pragma unroll
      for (int k = 0; k < PER; ++k) v[c * PER + k][i] = w[k];
    }
  }
  const PT<CLS>* M = cst + st.mat_off;          // dense kinds only (the dispatcher sends the L and G maps to stage_vec)
  const bool has_tw = st.tw_off >= 0;
  const bool own_tw = st.tw_per > 0;     // see stage_vec
  int xi0 = 0;
  const PT<CLS>* twp = cst + st.tw_off;
  if (has_tw) {
    if (own_tw) twp += (blk - mdiv(blk, st.m_twper) * st.tw_per) * D;
    else { const int pi = mdiv(x0, n_magic); xi0 = x0 - pi * n; }
  }
  // row by row, each stored as soon as it is complete (every input is in registers already): 4 d + 4 live values
#pragma unroll
  for (int i = 0; i < D; ++i) {
    V o[W];
#pragma unroll
    for (int w = 0; w < W; ++w) o[w] = m_dot<CLS, D>(v[w], M + i * D, mc);
    if (has_tw) {
      int ti = i;
      if (!own_tw) { const int xd = mdiv(xi0 + i * rts, st.m_twdiv); ti = xd - mdiv(xd, st.m_twmod) * st.tw_mod; }
      const PT<CLS> tw = twp[ti];
#pragma unroll
      for (int w = 0; w < W; ++w) o[w] = m_mul<CLS>(o[w], tw, mc);
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      VV w;
#pragma unroll
      for (int k = 0; k < PER; ++k) w[k] = o[c * PER + k];
      reinterpret_cast<VV*>(base + i * rts)[c] = w;
    }
  }
}

// ST_POW2F / ST_POW2I: K levels of the 2-power factor on one 2^K-element register tile (plan.h).
// Element j of tile (high, low) sits at (high << (sh + K) | low) + j * rts, rts = 2^sh = 2^(s_lo - 1);
// the butterfly of level s_lo + l on elements j, j | 2^l uses table entry (rts << l) + low + (j mod 2^l) * rts.
// Canonical residues in and out of every butterfly (m_mul reduces fully), so the tiles compose with the
// odd primes' stages in any order.  (Unreduced butterflies for q < 2^27 — 7 instead of 12 instructions — were
// built and measured twice: as a second path in this function they spill 66-99 VGPRs to scratch (crt of 64*9*25
// 0.18 -> 0.31 ms); as separate instantiations they gain 0-5 % (2^11*7 crt 0.284 -> 0.269 ms) for +50 % build time.)
template <int CLS, int K, bool INV>
__device__ __forceinline__ void stage_pow2(const Stage& st, MV<CLS>* __restrict__ buf, int tile,
                                           const PT<CLS>* __restrict__ cst, const ModCtx& mc) {
  using V = MV<CLS>;
  constexpr int NE = 1 << K;
  constexpr int AC = CLS == 4 ? 2 : CLS;       // class 4: the tiles compute on canonical residues with class 2's arithmetic
  const u64 q = mc.q;
  const int rts = st.rts, sh = st.p - 1;
  const int low = tile & (rts - 1), high = tile >> sh;
  V* base = buf + ((high << (sh + K)) | low);
  V v[NE];
  if (rts == 1) {                      // contiguous tile: 16-byte LDS accesses
    constexpr int PER = 16 / (int)sizeof(V);
    typedef V VV __attribute__((ext_vector_type(PER)));
    if constexpr (NE >= PER) {
#pragma unroll
      for (int c = 0; c < NE / PER; ++c) {
        const VV w = reinterpret_cast<const VV*>(base)[c];
#pragma unroll
        for (int k = 0; k < PER; ++k) v[c * PER + k] = w[k];
      }
    } else {
#pragma unroll
      for (int j = 0; j < NE; ++j) v[j] = base[j];
    }
  } else {
#pragma unroll
    for (int j = 0; j < NE; ++j) v[j] = base[j * rts];
  }
  if constexpr (CLS == 4 && INV) {
    if (st.pad[1]) {                     // the tile that follows the lazy dense stages of an inverse program: [0,2q) -> [0,q)
#pragma unroll
      for (int j = 0; j < NE; ++j) v[j] = min(v[j], v[j] - (u32)q);
    }
  }
  const PT<CLS>* tw = cst + st.tw_off + low;
  if constexpr (!INV) {
#pragma unroll
    for (int l = 0; l < K; ++l) {
      const int half = rts << l;
#pragma unroll
      for (int j = 0; j < NE; ++j) {
        if (j & (1 << l)) continue;
        const V t = m_mul<AC>(v[j | (1 << l)], tw[half + (j & ((1 << l) - 1)) * rts], mc);
        const V x = v[j];
        v[j] = m_add<AC>(x, t, q);
        v[j | (1 << l)] = m_sub<AC>(x, t, q);
      }
    }
  } else {
#pragma unroll
    for (int l = K - 1; l >= 0; --l) {
      const int half = rts << l;
#pragma unroll
      for (int j = 0; j < NE; ++j) {
        if (j & (1 << l)) continue;
        const V x = v[j], y = v[j | (1 << l)];
        v[j] = m_add<AC>(x, y, q);
        v[j | (1 << l)] = m_mul<AC>(m_sub<AC>(x, y, q), tw[half + (j & ((1 << l) - 1)) * rts], mc);
      }
    }
    if (st.mat_off >= 0) {             // this tile holds level 1: its X outputs take mhat^-1 here (the Y outputs through the table)
      const u64 mh = cst[st.mat_off];
#pragma unroll
      for (int j = 0; j < NE; j += 2) v[j] = m_mul<AC>(v[j], mh, mc);
    }
  }
  if (rts == 1) {
    constexpr int PER = 16 / (int)sizeof(V);
    typedef V VV __attribute__((ext_vector_type(PER)));
    if constexpr (NE >= PER) {
#pragma unroll
      for (int c = 0; c < NE / PER; ++c) {
        VV w;
#pragma unroll
        for (int k = 0; k < PER; ++k) w[k] = v[c * PER + k];
        reinterpret_cast<VV*>(base)[c] = w;
      }
    } else {
#pragma unroll
      for (int j = 0; j < NE; ++j) base[j] = v[j];
    }
  } else {
#pragma unroll
    for (int j = 0; j < NE; ++j) base[j * rts] = v[j];
  }
}

// every stage of one program over the `tot` packed coefficients in buf; ends with a barrier
// HOIST: the dispatch on the vector length / tile size OUTSIDE the loop over vectors.  The compiler then keeps
// the stage's constants (a whole d x d matrix: 169 SGPRs at p = 13) in registers across the loop: good
// where there are VGPRs to spill SGPRs into (the fused poly-mul at 128: -3..6 %), bad at the 64-80 of the
// single-program kernels (scratch spills: crt of m = 15015 0.042 -> 0.058 ms).
// BIG (class 2): the 18- and 20-element vectors of merged prime powers (stage_vec_big) are instantiated too.  Separate
// kernels — inside the ordinary ones these bodies cost every program registers (spills in the fused poly-mul).
template <int CLS, bool HOIST, bool BIG = false>
__device__ __forceinline__ void run_stages(MV<CLS>* __restrict__ buf, int tot, int n, u64 n_magic,
                                           const Stage* __restrict__ stages, int nstages,
                                           const PT<CLS>* __restrict__ cst, const ModCtx& mc) {
#define LOLHIP_LOOP(COUNT, CALL) for (int it = threadIdx.x; it < (COUNT); it += blockDim.x) { CALL; }
#define LOLHIP_TILES(X) X(2, 1, false) X(3, 1, true) X(4, 2, false) X(5, 2, true) X(6, 3, false) X(7, 3, true) X(8, 4, false) X(9, 4, true)
#define LOLHIP_VECS(X) X(2) X(3) X(4) X(5) X(6) X(7) X(10) X(11) X(12) X(13)
#define LOLHIP_VECS8(X) X(8) X(9)                                // BIG kernels only: 3 (x) 5 as one stage (plan.cpp merge_stages<true>)
#ifdef LH_NO_VL
#define LOLHIP_VECSL(X)
#else
#define LOLHIP_VECSL(X) X(18) X(20)                           // class 2 only: merged prime powers 3^3, 5^2
#endif
  for (int s = 0; s < nstages; ++s) {
#ifdef LH_ABL_SKIP_STAGES        // timing-only ablation: bit s set = stage s of every program is skipped (results garbage)
    if ((LH_ABL_SKIP_STAGES >> s) & 1) continue;
#endif
    const Stage st = stages[s];
    if (st.kind == ST_DIAG || st.kind == ST_SCALE) {
      for (int x = threadIdx.x; x < tot; x += blockDim.x) {
        const int pi = mdiv(x, n_magic);
        const int xd = mdiv(x - pi * n, st.m_twdiv);
        buf[x] = m_mul<CLS>(buf[x], cst[st.tw_off + xd - mdiv(xd, st.m_twmod) * st.tw_mod], mc);
      }
    } else if (st.kind == ST_POW2F || st.kind == ST_POW2I) {
      const int ntile = tot >> st.d;
      const int sel = st.d * 2 + (st.kind == ST_POW2I ? 1 : 0);
      if constexpr (HOIST) {
#define LOLHIP_X(SEL, K, INV) case SEL: LOLHIP_LOOP(ntile, (stage_pow2<CLS, K, INV>(st, buf, it, cst, mc))) break;
        switch (sel) { LOLHIP_TILES(LOLHIP_X) default: break; }
#undef LOLHIP_X
      } else {
#define LOLHIP_X(SEL, K, INV) case SEL: stage_pow2<CLS, K, INV>(st, buf, it, cst, mc); break;
        LOLHIP_LOOP(ntile, switch (sel) { LOLHIP_TILES(LOLHIP_X) default: break; })
#undef LOLHIP_X
      }
    } else {
      const int nvec = mdiv(tot, st.m_d);           // tot / d: exact, tot < 2^20
      // four (d <= 7) or two adjacent vectors per thread where the stride allows it (stage_vecw)
      // (class 2 only: in the classes with 128-bit accumulators the fourfold dot products multiply the code size
      // and the compile time — 15 minutes for one translation unit — for kernels that are multiplier-bound anyway)
      const bool dense = st.kind == ST_DFTP || st.kind == ST_CRTP || st.kind == ST_CRTPINV;
      const int wd = st.d <= 7 ? 4 : 2;            // vectors per thread
      // (two vectors per thread for d = 10..13 was tried: with the hoisted dispatch the 169-entry matrix and two
      // vectors spill 184 VGPRs to scratch in the fused poly-mul)
      const bool wide = c2<CLS>() && dense && st.d <= 7 && (st.rts & (wd - 1)) == 0 && (st.tw_off < 0 || (st.tw_div & (wd - 1)) == 0 || st.tw_mod == 1);
      bool done = false;
      if constexpr (c2<CLS>()) if (wide) {
        done = true;
        const int nqw = st.d <= 7 ? (nvec >> 2) : (nvec >> 1);      // rts | n / d, so nvec is a multiple of the width too
#define LOLHIP_VECSW(X) X(2, 4) X(3, 4) X(4, 4) X(5, 4) X(6, 4) X(7, 4)
        if constexpr (HOIST) {
#define LOLHIP_X(D, W) case D: LOLHIP_LOOP(nqw, (stage_vecw<CLS, D, W>(st, buf, it, n, n_magic, cst, mc))) break;
          switch (st.d) { LOLHIP_VECSW(LOLHIP_X) default: break; }
#undef LOLHIP_X
        } else {
#define LOLHIP_X(D, W) case D: stage_vecw<CLS, D, W>(st, buf, it, n, n_magic, cst, mc); break;
          LOLHIP_LOOP(nqw, switch (st.d) { LOLHIP_VECSW(LOLHIP_X) default: break; })
#undef LOLHIP_X
        }
#undef LOLHIP_VECSW
      }
      if (!done) {
        if constexpr (HOIST) {
#define LOLHIP_X(D) case D: LOLHIP_LOOP(nvec, (stage_vec<CLS, D>(st, buf, it, n, n_magic, cst, mc))) break;
#define LOLHIP_XL(D) case D: LOLHIP_LOOP(nvec, (stage_vec_big<CLS, D>(st, buf, it, cst, mc))) break;
          if constexpr (c2<CLS>() && BIG) { switch (st.d) { LOLHIP_VECS(LOLHIP_X) LOLHIP_VECS8(LOLHIP_X) LOLHIP_VECSL(LOLHIP_XL) default: break; } }
          else { switch (st.d) { LOLHIP_VECS(LOLHIP_X) default: break; } }      // other lengths are excluded on the host (mixed_ok)
#undef LOLHIP_XL
#undef LOLHIP_X
        } else {
#define LOLHIP_X(D) case D: stage_vec<CLS, D>(st, buf, it, n, n_magic, cst, mc); break;
#define LOLHIP_XL(D) case D: stage_vec_big<CLS, D>(st, buf, it, cst, mc); break;
          if constexpr (c2<CLS>() && BIG) { LOLHIP_LOOP(nvec, switch (st.d) { LOLHIP_VECS(LOLHIP_X) LOLHIP_VECS8(LOLHIP_X) LOLHIP_VECSL(LOLHIP_XL) default: break; }) }
          else { LOLHIP_LOOP(nvec, switch (st.d) { LOLHIP_VECS(LOLHIP_X) default: break; }) }
#undef LOLHIP_XL
#undef LOLHIP_X
        }
      }
    }
    __syncthreads();
  }
#undef LOLHIP_LOOP
#undef LOLHIP_TILES
#undef LOLHIP_VECS
#undef LOLHIP_VECS8
#undef LOLHIP_VECSL
}

template <int CLS> __device__ __forceinline__ MV<CLS> from_raw(i64 x, u64 q) {
  if constexpr (wide<CLS>()) return canon_in(x, q);
  else return (u32)x + ((u32)q & (u32)(x >> 63));       // (-q, q) -> [0, q), q < 2^32
}

// MODE 0: y = program(a).  MODE 2: y = crtInv(crt(a) * crt(b)).
// (Tables are separate __restrict__ parameters, not members of a struct: only then can the
// compiler prove them unclobbered by the stores of earlier items and fetch matrix rows with
// scalar loads — as struct members they became per-lane vector loads, 338 VGPRs of them at p = 13.)
// Occupancy: a 46 KiB (23 KiB in the 32-bit classes) polynomial leaves LDS room for 3-6 workgroups
// per CU; the single-program form of the 32-bit classes fits 80 VGPRs (6 waves/SIMD); the fused
// poly-mul keeps a-hat and b's loads live and gets 128 (4 waves/SIMD), as does the 64-bit class
// (at 80 it spills: measured slower).
template <int CLS, int MODE, int KMAX, bool BIG = false>
__global__ void __launch_bounds__(512, (MODE == 0 && (CLS == 2 || CLS == 4)) ? LOLHIP_MIXED_W2 : (MODE == 0 && CLS == 1) ? 6 : 4)
k_mixed(i64* y_out, const i64* a_in, const i64* b_in, i64 B, int T, int n, int ppw, i64 ngroups,
        const Stage* __restrict__ st_a, int n_a, const Stage* __restrict__ st_b, int n_b,
        const u64* __restrict__ consts64, const u32* __restrict__ consts32, int cpc, const ModCtx* __restrict__ mod) {
  using V = MV<CLS>;
  // hoisted dispatch (run_stages): the fused poly-mul of the 32-bit classes only — in the 64-bit classes it measured
  // no gain (m = 15015, 61 bits: 0.324 vs 0.320 ms)
  constexpr bool HOISTED = ((MODE == 2 && !wide<CLS>()) || LOLHIP_MIXED_HOIST0) && !BIG;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  V* buf = reinterpret_cast<V*>(smem);
  const u64 n_magic = (((u64)1 << 40) / (u64)n) + 1;
  const i64 items = ngroups * T;
  const int nthr = blockDim.x, tid = threadIdx.x;
  for (i64 item = blockIdx.x; item < items; item += gridDim.x) {
    const i64 g = item / T;
    const int t = (int)(item % T);
    const i64 b0 = g * ppw;
    const int np = (int)((B - b0) < ppw ? (B - b0) : ppw);
    const int tot = np * n;
    const ModCtx mc = mod[t];
    const PT<CLS>* cst;
    if constexpr (sizeof(PT<CLS>) == 4) cst = consts32 + (size_t)t * cpc; else cst = consts64 + (size_t)t * cpc;
    // Global memory through buffer descriptors over this group's window: one 32-bit lane offset
    // plus a wave-uniform step per batch entry (no 64-bit addresses in VGPRs), and the range
    // check does the x < tot predicate: loads past the window return 0, stores are dropped.
    const size_t gbase = (size_t)b0 * n * T + (size_t)t;       // element x of this group lives at gbase + x * T
    const u32 wbytes = ((u32)(tot - 1) * (u32)T + 1u) * 8u;    // <= 8192 * 64 * 8
    const u32 step = (u32)nthr * (u32)T * 8u;
    const rsrc_t wa = __builtin_amdgcn_make_buffer_rsrc((void*)(a_in + gbase), 0, wbytes, 0x00020000);
    const rsrc_t wy = __builtin_amdgcn_make_buffer_rsrc((void*)(y_out + gbase), 0, wbytes, 0x00020000);
    auto goff = [&]() -> u32 { return (u32)fresh(tid) * (u32)T * 8u; };
    auto load16 = [&](u64 (&r)[KMAX], rsrc_t w) {
      const u32 o = goff();
#pragma unroll
      for (int k = 0; k < KMAX; ++k) r[k] = (k * nthr < tot) ? load_u64(w, o, (u32)k * step) : 0;      // uniform skip of empty batches
      __builtin_amdgcn_sched_barrier(0);
    };
    auto to_lds = [&](const u64 (&r)[KMAX]) {
      const int x0 = fresh(tid);
#pragma unroll
      for (int k = 0; k < KMAX; ++k) { const int x = x0 + k * nthr; if (x < tot) buf[x] = from_raw<CLS>((i64)r[k], mc.q); }
    };
    auto store16 = [&]() {
      const u32 o = goff();
      const int x0 = fresh(tid);
#pragma unroll
      for (int k = 0; k < KMAX; ++k) {
        const int x = x0 + k * nthr;
        if (k * nthr < tot) {
          V r = buf[x < tot ? x : 0];
          if constexpr (CLS == 4) r = min(r, r - (u32)mc.q);        // a program may end on a lazy stage: [0,2q) -> [0,q)
          store_u64(wy, o, (u32)k * step, (u64)r);
        }
      }
    };

    u64 ra[KMAX];
    load16(ra, wa);
    if constexpr (MODE == 0) {
      to_lds(ra);
      __syncthreads();
      run_stages<CLS, HOISTED, BIG>(buf, tot, n, n_magic, st_a, n_a, cst, mc);
      store16();
    } else {
      const bool square = (a_in == b_in);
      to_lds(ra);
      // b's loads go out now and land under a's stages (holding them back to fit a third
      // workgroup per CU measured 0.125 vs 0.122 ms on config 4: not worth it)
      // (BIG: b's 2 KMAX raw registers would spill under the 20-element vectors: loaded after a's stages instead; in the
      // 12-coefficient variant, where they fit, the early load measured 2-3 % SLOWER at m = 15015, q ~ 2^29)
      if constexpr (!BIG) if (!square) load16(ra, __builtin_amdgcn_make_buffer_rsrc((void*)(b_in + gbase), 0, wbytes, 0x00020000));
      __syncthreads();
      run_stages<CLS, HOISTED, BIG>(buf, tot, n, n_magic, st_a, n_a, cst, mc);
      if constexpr (BIG) if (!square) load16(ra, __builtin_amdgcn_make_buffer_rsrc((void*)(b_in + gbase), 0, wbytes, 0x00020000));
      V ah[KMAX];                        // a-hat: every thread keeps the positions it owns
      {
        const int x0 = fresh(tid);
#pragma unroll
        for (int k = 0; k < KMAX; ++k) { const int x = x0 + k * nthr; ah[k] = x < tot ? buf[x] : 0; }
      }
      if (!square) {
        __syncthreads();                 // every a-hat coefficient is in registers before b overwrites the buffer
        to_lds(ra);
        __syncthreads();
        run_stages<CLS, HOISTED, BIG>(buf, tot, n, n_magic, st_a, n_a, cst, mc);
      }
      {
        const int x0 = fresh(tid);
#pragma unroll
        for (int k = 0; k < KMAX; ++k) { const int x = x0 + k * nthr; if (x < tot) buf[x] = m_mul<CLS, false>(ah[k], (u64)buf[x], mc); }
      }
      __syncthreads();
      run_stages<CLS, HOISTED, BIG>(buf, tot, n, n_magic, st_b, n_b, cst, mc);
      store16();
    }
    __syncthreads();                     // the buffer is reused by the next item
  }
}

// one kernel instantiation per (class, mode, coefficients per thread): the 64-bit fused poly-mul takes minutes to
// compile per instantiation, so its two KMAX variants live in separate translation units (mixed_cls{0,3}f{12,16}.hip)
struct MixedGeom { int ppw; i64 ngroups; size_t lds_bytes; int threads; i64 grid; size_t per_thread; };
template <int CLS> static MixedGeom mixed_geom(const MixedLaunch& a) {
  MixedGeom g;
  // pack small polynomials up to ~2048 coefficients per workgroup
  g.ppw = 1;
  while ((size_t)(g.ppw * 2) * a.n <= 2048 && g.ppw * 2 <= a.B) g.ppw *= 2;
  g.ngroups = (a.B + g.ppw - 1) / g.ppw;
  const size_t coeffs = (size_t)g.ppw * a.n;
  g.lds_bytes = coeffs * sizeof(MV<CLS>);
  g.threads = coeffs > 4096 ? 512 : (coeffs > 2048 ? 256 : 128);     // coeffs <= 16 * threads
  g.grid = g.ngroups * a.T;
  if (g.grid > 65536) g.grid = 65536;
  g.per_thread = (coeffs + g.threads - 1) / g.threads;
  return g;
}
template <int CLS, int MODE, int KMAX, bool BIG = false>
hipError_t launch_cls_k(const MixedLaunch& a) {
  const MixedGeom g = mixed_geom<CLS>(a);
  hipLaunchKernelGGL((k_mixed<CLS, MODE, KMAX, BIG>), dim3((unsigned)g.grid), dim3(g.threads), g.lds_bytes, a.stream, a.y, a.a, a.b,
                     a.B, a.T, (int)a.n, g.ppw, g.ngroups, a.st_a, a.n_a, a.st_b, a.n_b, a.consts, a.consts32, a.cpc, a.mod);
  return hipGetLastError();
}
template <int CLS, int MODE>
hipError_t launch_cls(const MixedLaunch& a) {
  const bool k12 = mixed_geom<CLS>(a).per_thread <= 12;
  if constexpr (c2<CLS>()) if (a.big) return k12 ? launch_cls_k<CLS, MODE, 12, true>(a) : launch_cls_k<CLS, MODE, 16, true>(a);
  return k12 ? launch_cls_k<CLS, MODE, 12>(a) : launch_cls_k<CLS, MODE, 16>(a);
}

}  // namespace lolhip
