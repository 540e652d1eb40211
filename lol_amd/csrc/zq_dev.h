// lol_amd/csrc/zq_dev.h — Z_q arithmetic for gfx950 device code (and a host mirror
// of the constant derivations).
//
// Replaces the reference's `Zq` class (lol-cpp/.../CPP/types.h:52-94), which reduces
// with a 64-bit `%` after every operation and keeps a process-global modulus.  Here:
//   * multiplications by plan constants (twiddles, scales) use Shoup's precomputed-
//     quotient form with Harvey-style lazy ranges ([0,2q) / [0,4q)), integer only;
//   * variable*variable products (pointwise mul, dense p-point stages) use an exact
//     128/64 remainder by a precomputed reciprocal (Moeller-Granlund);
//   * every kernel canonicalises to [0,q) exactly once before its final store,
//     which is the reference's post-condition (zq.cpp:57-68).
// Valid for any modulus 2 <= q < 2^62 (even moduli included; the reference's own
// correct range is q < ~2^31.5, types.h:79-84).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define LH_HD __host__ __device__ __forceinline__
#define LH_D __device__ __forceinline__
#else
#define LH_HD inline
#define LH_D inline
#endif

namespace lolhip {

typedef uint64_t u64;
typedef uint32_t u32;
typedef int64_t i64;

// Per-modulus context for exact (a*b) mod q with arbitrary operands.
struct ModCtx {
  u64 q;   // modulus
  u64 d;   // q << s, top bit set
  u64 v;   // floor((2^128-1)/d) - 2^64
  u32 s;   // normalisation shift (>= 2 because q < 2^62)
  u32 r32; // 2^32 mod q when q < 2^32 (else 0): into 32-bit Montgomery form with one Shoup product
  u64 mu;  // floor(2^64 / q) (Barrett constant of the 32-bit path)
  u64 nqinv;  // -q^-1 mod 2^64 for odd q (Montgomery reductions), else 0
  u32 r32p;   // floor(r32 * 2^32 / q)
  u32 pad;
};

// Shoup pair: w and floor(w * 2^64 / q)
struct ShoupW { u64 w, wp; };

// host-side derivations (used by plan.cpp)
inline ModCtx make_modctx(u64 q) {
  ModCtx c;
  c.q = q;
  c.s = (u32)__builtin_clzll(q);
  c.d = q << c.s;
  c.v = (u64)((~(unsigned __int128)0) / c.d - ((unsigned __int128)1 << 64));
  c.pad = 0;
  c.r32 = c.r32p = 0;
  if (q < ((u64)1 << 32)) {
    c.r32 = (u32)((((u64)1) << 32) % q);
    c.r32p = (u32)((((u64)c.r32) << 32) / q);
  }
  c.mu = (u64)((((unsigned __int128)1) << 64) / q);
  c.nqinv = 0;
  if (q & 1) {                       // Newton iteration: x <- x (2 - q x), doubling the correct low bits
    u64 x = q;                       // q * q = 1 mod 8: three correct bits
    for (int i = 0; i < 6; ++i) x *= 2 - q * x;
    c.nqinv = 0 - x;
  }
  return c;
}
inline ShoupW make_shoup(u64 w, u64 q) {
  ShoupW s;
  s.w = w;
  s.wp = (u64)((((unsigned __int128)w) << 64) / q);
  return s;
}

#if defined(__HIPCC__)

LH_D u64 mulhi64(u64 a, u64 b) { return __umul64hi(a, b); }

// x - m if x >= m else x; requires x, m < 2^63
LH_D u64 csub(u64 x, u64 m) {
  u64 t = x - m;
  return ((i64)t < 0) ? x : t;
}

// canonical representative of a reference-style input in (-q, q)
LH_D u64 canon_in(i64 x, u64 q) { return x < 0 ? (u64)(x + (i64)q) : (u64)x; }

// w*y mod q in [0, 2q) for ANY 64-bit y (w < q)
LH_D u64 shoup_lazy(u64 y, u64 w, u64 wp, u64 q) {
  u64 Q = mulhi64(wp, y);
  return w * y - Q * q;
}

// ---- hand-scheduled pieces for the NTT inner loop --------------------------------
// v_mad_u64_u32 (32x32+64 -> 64) and v_lshl_add_u64 are the two 64-bit integer
// workhorses on gfx950; spelling them out keeps hipcc from splitting the chains into
// v_mul_lo_u32 + carry adds (each carry add needs a 2-wait-state hazard nop here).
#ifndef LOLHIP_ASM_MAD
#define LOLHIP_ASM_MAD 3
#endif
LH_D u64 mad64(u32 a, u32 b, u64 c) {
#if LOLHIP_ASM_MAD
  u64 d, carry;
  asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(carry) : "v"(a), "v"(b), "v"(c));
  return d;
#else
  return (u64)a * (u64)b + c;
#endif
}
// 64-bit adds.  Plain C: hipcc selects v_lshl_add_u64 for these by itself — PROVIDED a uniform
// addend is not a compile-time-visible negation (x + (0 - m) becomes a two-instruction
// v_sub_co/v_subb; the per-modulus constants are laundered once in QK, pow2_impl.h).  As
// single-instruction inline asm each of these cost an s_nop: the hazard recognizer pads every
// asm block boundary (1,460 s_nop per wave in the round-1 fused kernel, 630 now).
LH_D u64 add64(u64 a, u64 b) { return a + b; }
LH_D u64 shl1_add64(u64 a, u64 b) { return (a << 1) + b; }
LH_D u64 add64u(u64 a, u64 b_uniform) { return a + b_uniform; }
LH_D u64 shl1_add64u(u64 a, u64 b_uniform) { return (a << 1) + b_uniform; }
LH_D u32 lo32(u64 x) { return (u32)x; }
LH_D u32 hi32(u64 x) { return (u32)(x >> 32); }

// x + negm if that is non-negative else x, i.e. csub(x, m) with negm = -m (mod 2^64);
// requires |x - m| < 2^63
LH_D u64 csubn(u64 x, u64 negm) {
  u64 t = add64u(x, negm);     // negm is a per-modulus constant everywhere this is used
  return ((int)hi32(t) < 0) ? x : t;
}

// init + w*y - Q*q (mod 2^64) with an APPROXIMATE quotient
//   Q = wp.hi*y.hi + hi32(wp.hi*y.lo) + hi32(wp.lo*y.hi)  in {floor(wp*y/2^64) - 2 .. same}
// so (result - init) = w*y mod q + {0..3}*q  in [0, 4q) for any 64-bit y.  7 mads + 2
// mul_hi instead of the 10 multiplies of the exact form; nq = -q (mod 2^64).
// WS: the twiddle pair is wave-uniform (the scalar-loaded twiddles of the bottom levels, the scale
// constants): its halves go into the multiplies as SGPR operands — every instruction below reads at most
// one — instead of being copied to VGPRs first (450 v_mov per wave of the fused kernel).
template <bool WS = false>
LH_D u64 shoup_acc(u64 y, u64 w, u64 wp, u64 nq, u64 init) {
#if LOLHIP_ASM_MAD == 4
  // One block, 9 v_mad_u64_u32.  The quotient's cross terms are summed EXACTLY:
  //   C = wp.hi*y.lo + wp.lo*y.hi  (65 bits: 64 in v[120:121], the carry in vcc)
  //   Q = wp.hi*y.hi + (C >> 32)   in {floor(wp*y/2^64) - 1, same}
  // and C >> 32 is assembled as the pair v[122:123] = {C.hi, carry} (VGPR pairs must be
  // even-aligned, so it cannot simply overlap C).  Halves of a 64-bit asm operand cannot be
  // named, hence the six hard-wired scratch VGPRs (every kernel using this has a 128-VGPR
  // budget).  gfx950 needs 2 wait states between a VALU
  // write of vcc and a VALU read of it; three independent mads sit in between, their junk
  // carries go to another SGPR pair.
  u64 t = init, h, sj;
  asm("v_mad_u64_u32 v[120:121], vcc, %[wph], %[ylo], 0\n\t"
      "v_mad_u64_u32 v[120:121], vcc, %[wpl], %[yhi], v[120:121]\n\t"
      "v_mad_u64_u32 %[t], %[sj], %[wl], %[ylo], %[t]\n\t"
      "v_mad_u64_u32 %[h], %[sj], %[wl], %[yhi], 0\n\t"
      "v_mad_u64_u32 %[h], %[sj], %[wh], %[ylo], %[h]\n\t"
      "v_mov_b32 v122, v121\n\t"
      "v_cndmask_b32_e64 v123, 0, 1, vcc\n\t"
      "v_mad_u64_u32 v[118:119], %[sj], %[wph], %[yhi], v[122:123]\n\t"
      "v_mad_u64_u32 %[t], %[sj], v118, %[nql], %[t]\n\t"
      "v_mad_u64_u32 %[h], %[sj], v118, %[nqh], %[h]\n\t"
      "v_mad_u64_u32 %[h], %[sj], v119, %[nql], %[h]"
      : [t] "+v"(t), [h] "=&v"(h), [sj] "=&s"(sj)
      : [wph] "v"(hi32(wp)), [wpl] "v"(lo32(wp)), [ylo] "v"(lo32(y)), [yhi] "v"(hi32(y)),
        [wl] "v"(lo32(w)), [wh] "v"(hi32(w)), [nql] "s"(lo32(nq)), [nqh] "s"(hi32(nq))
      : "vcc", "v118", "v119", "v120", "v121", "v122", "v123");
  u32 th = hi32(t) + lo32(h);
  asm("" : "+v"(th));          // keeps it ONE v_add_u32 (else: t + (h << 32) as v_mov + v_lshl_add_u64)
  return ((u64)th << 32) | lo32(t);
#elif LOLHIP_ASM_MAD == 3
  // Two asm blocks per product (hipcc pads every separate asm statement that writes an SGPR
  // carry with s_nop; inside a block there is nothing to pad: VGPR RAW is interlocked).
  u64 Q, t, h;
  u32 ah, bh;
#if defined(LH_ABL_PADN) && LH_ABL_PADN == 1      // timing-only ablations: what an s_nop costs in this kernel
#define LH_ABL_PAD "\n\ts_nop 0"
#elif defined(LH_ABL_PADN) && LH_ABL_PADN == 2
#define LH_ABL_PAD "\n\ts_nop 1"
#else
#define LH_ABL_PAD ""
#endif
#define LH_SHOUP_BLOCKS(WC)                                                                          \
  asm("v_mul_hi_u32 %1, %3, %5\n\t"                                                                  \
      "v_mul_hi_u32 %2, %4, %6\n\t"                                                                  \
      "v_mad_u64_u32 %0, vcc, %3, %6, 0\n\t"                                                         \
      "v_mad_u64_u32 %0, vcc, %1, 1, %0\n\t"                                                         \
      "v_mad_u64_u32 %0, vcc, %2, 1, %0"                                                             \
      : "=&v"(Q), "=&v"(ah), "=&v"(bh)                                                               \
      : WC(hi32(wp)), WC(lo32(wp)), "v"(lo32(y)), "v"(hi32(y))                                       \
      : "vcc");                                                                                      \
  asm("v_mad_u64_u32 %0, vcc, %2, %4, %10\n\t"                                                       \
      "v_mad_u64_u32 %1, vcc, %2, %5, 0\n\t"                                                         \
      "v_mad_u64_u32 %0, vcc, %6, %8, %0\n\t"                                                        \
      "v_mad_u64_u32 %1, vcc, %3, %4, %1\n\t"                                                        \
      "v_mad_u64_u32 %1, vcc, %6, %9, %1\n\t"                                                        \
      "v_mad_u64_u32 %1, vcc, %7, %8, %1" LH_ABL_PAD                                                 \
      : "=&v"(t), "=&v"(h)                                                                           \
      : WC(lo32(w)), WC(hi32(w)), "v"(lo32(y)), "v"(hi32(y)), "v"(lo32(Q)), "v"(hi32(Q)),            \
        "s"(lo32(nq)), "s"(hi32(nq)), "v"(init)      /* nq: wave-uniform, one SGPR operand per mad */ \
      : "vcc")
#define LH_WC_V(x) "v"(x)
#define LH_WC_S(x) "s"(x)
  // Two asm blocks per product (hipcc pads every separate asm statement that writes an SGPR
  // carry with s_nop; inside a block there is nothing to pad: VGPR RAW is interlocked).
  if constexpr (WS) { LH_SHOUP_BLOCKS(LH_WC_S); } else { LH_SHOUP_BLOCKS(LH_WC_V); }
#undef LH_SHOUP_BLOCKS
#undef LH_WC_V
#undef LH_WC_S
  u32 th = hi32(t) + lo32(h);
  asm("" : "+v"(th));          // keeps it ONE v_add_u32 (else: t + (h << 32) as v_mov + v_lshl_add_u64)
  return ((u64)th << 32) | lo32(t);
#else
  const u32 ah = __umulhi(hi32(wp), lo32(y));
  const u32 bh = __umulhi(lo32(wp), hi32(y));
  const u64 Q = mad64(bh, 1u, mad64(hi32(wp), hi32(y), (u64)ah));
  u64 t = mad64(lo32(w), lo32(y), init);
  t = mad64(lo32(Q), lo32(nq), t);
  u64 h = mad64(lo32(w), hi32(y), 0);
  h = mad64(hi32(w), lo32(y), h);
  h = mad64(lo32(Q), hi32(nq), h);
  h = mad64(hi32(Q), lo32(nq), h);
  u32 th = hi32(t) + lo32(h);
  asm("" : "+v"(th));          // keeps it ONE v_add_u32 (else: t + (h << 32) as v_mov + v_lshl_add_u64)
  return ((u64)th << 32) | lo32(t);
#endif
}

// remainder of the 128-bit value (u1:u0) by c.q, requires (u1:u0) < q * 2^64
LH_D u64 rem128(u64 x1, u64 x0, const ModCtx& c) {
  // normalise
  u64 u1 = (x1 << c.s) | (x0 >> (64 - c.s));
  u64 u0 = x0 << c.s;
  // Moeller-Granlund 2-by-1 division step
  unsigned __int128 qq = (unsigned __int128)c.v * u1 + (((unsigned __int128)u1 << 64) | u0);
  u64 q1 = (u64)(qq >> 64) + 1, q0 = (u64)qq;
  u64 r = u0 - q1 * c.d;
  if (r > q0) r += c.d;
  if (r >= c.d) r -= c.d;
  return r >> c.s;
}

// exact a*b mod q, a,b < q
LH_D u64 mulmod(u64 a, u64 b, const ModCtx& c) {
  unsigned __int128 z = (unsigned __int128)a * b;
  return rem128((u64)(z >> 64), (u64)z, c);
}

// arbitrary 128-bit value mod q
LH_D u64 reduce128(u64 x1, u64 x0, const ModCtx& c) {
  u64 r1 = rem128(0, x1, c);
  return rem128(r1, x0, c);
}

LH_D u64 addmod(u64 a, u64 b, u64 q) { u64 s = a + b; return s >= q ? s - q : s; }
LH_D u64 submod(u64 a, u64 b, u64 q) { return a >= b ? a - b : a + q - b; }

#endif  // __HIPCC__

}  // namespace lolhip
