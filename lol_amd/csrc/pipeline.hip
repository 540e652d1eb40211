// lol_amd/csrc/pipeline.hip — fused streaming kernels for the ring-level pipelines SymmSHE
// builds out of Tensor ops.  gfx950 only; all four are HBM-bound
// element-wise passes over [.][B][n][T] int64 slabs (component t innermost).
//
//   k_ctmul      mulG <$> (c * d) for two linear ciphertexts, CRT basis   SymmSHE.hs:444-449
//   k_decompose  gadget decomposition of powerful-basis coefficients, every integer digit
//                reduced into all T components                            Cyc.hs:592-604,
//                Gadget.hs:96-101, ZqBasic.hs:227-264, Numeric.hs:202-205,227-234, SymmSHE.hs:314
//   k_knapsack   sum_j x_j *>> hint_j (+ addend), CRT basis               SymmSHE.hs:302-304,365-371
//   k_rescale    RescaleCyc (a,b) -> b: q_a^-1 (b - reduce (lift a))      Cyc.hs:529-542
//
// What the reference does op by op on boxed Haskell vectors (4 ring products, an addition and
// three mulG for a ciphertext product: 21 slab passes) is one pass here (4 reads, 3 writes).
#include <hip/hip_runtime.h>

#include <type_traits>

#include "pipeline.h"
#include "zq_dev.h"

namespace lolhip {

typedef unsigned __int128 u128;

// Every kernel here walks a flat element index g.  A workgroup owns one tile of 256 * EPT
// consecutive elements; whatever has to be divided out of g (position inside the polynomial,
// component) is divided once per workgroup in 64 bits and per element in 32 bits only.
constexpr int EPT = 2;                       // elements per thread
constexpr i64 TILE = 256 * EPT;
static inline bool tiles_for(i64 total, unsigned* blocks) {
  const i64 b = (total + TILE - 1) / TILE;
  if (b > 0x7fffffff) return false;
  *blocks = (unsigned)(b < 1 ? 1 : b);
  return true;
}

// ---------------------------------------------------------------------------------------
// ct x ct
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_ctmul(const i64* c0, const i64* c1, const i64* d0, const i64* d1, i64* e0, i64* e1, i64* e2,
        const i64* __restrict__ gcrt, i64 total, u32 per, int T, const ModCtx* __restrict__ mod) {
  const i64 s0 = (i64)blockIdx.x * TILE;                      // wave-uniform
  const u32 r_s = (u32)((u64)s0 % per);
#pragma unroll
  for (int k = 0; k < EPT; ++k) {
    const u32 l = (u32)k * 256u + threadIdx.x;
    const i64 g = s0 + l;
    if (g >= total) continue;
    u32 r = r_s + l;
    if (r >= per) r %= per;
    const ModCtx mc = mod[r % (u32)T];
    const u64 gv = (u64)gcrt[r];
    const u64 a0 = canon_in(c0[g], mc.q), a1 = canon_in(c1[g], mc.q);
    const u64 b0 = canon_in(d0[g], mc.q), b1 = canon_in(d1[g], mc.q);
    const u64 p0 = mulmod(a0, b0, mc);
    const u64 p2 = mulmod(a1, b1, mc);
    const u128 cross = (u128)a0 * b1 + (u128)a1 * b0;          // < 2 q^2 < q * 2^64
    const u64 p1 = rem128((u64)(cross >> 64), (u64)cross, mc);
    // all four inputs are read: the outputs may alias them
    e0[g] = (i64)mulmod(gv, p0, mc);
    e1[g] = (i64)mulmod(gv, p1, mc);
    e2[g] = (i64)mulmod(gv, p2, mc);
  }
}

hipError_t launch_ctmul(hipStream_t s, const i64* c0, const i64* c1, const i64* d0, const i64* d1, i64* e0, i64* e1,
                        i64* e2, const i64* gcrt, i64 B, i64 n, int T, const ModCtx* mod) {
  const i64 total = B * n * T;
  if (total == 0) return hipSuccess;
  unsigned blocks;
  if (!tiles_for(total, &blocks)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_ctmul, dim3(blocks), dim3(256), 0, s, c0, c1, d0, d1, e0, e1, e2, gcrt, total,
                     (u32)(n * T), T, mod);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// decompose
// ---------------------------------------------------------------------------------------
// floor(x / base) for any signed x, base >= 2, through the invariant-divisor multiply
__device__ __forceinline__ i64 floor_div(i64 x, const DecompParams& p) {
  const u64 n = x >= 0 ? (u64)x : (u64)(-x - 1);
  const u64 t1 = __umul64hi(p.magic, n);
  const u64 q = (t1 + ((n - t1) >> p.sh1)) >> p.sh2;
  return x >= 0 ? (i64)q : -(i64)q - 1;
}

// residue of a signed integer (|d| < 2^63) modulo q
__device__ __forceinline__ u64 reduce_signed(i64 d, const ModCtx& mc) {
  const u64 a = d >= 0 ? (u64)d : (u64)(-d);
  const u64 r = rem128(0, a, mc);
  return (d < 0 && r != 0) ? mc.q - r : r;
}

// One thread per ROW (coefficient): the T components are lifted and their digits extracted once, every
// digit is reduced into the T target components and stored as one contiguous T-word chunk, so a wave
// writes 64 * T consecutive words per digit.  (The first version ran one thread per output column: every
// digit was extracted T times and a store instruction covered only every T-th word.)
// Q32: every modulus below 2^31 and base <= 2^31: digits fit 31 bits, one-word Barrett reduction.
// V2: T = 2 and a 16-byte aligned digit slab: the pair goes out as one 16-byte store.
typedef u64 dec_u64x2 __attribute__((ext_vector_type(2)));
template <bool Q32, bool V2>
__global__ void __launch_bounds__(256)
k_decompose(const i64* __restrict__ c, i64* __restrict__ digits, i64 rows, DecompParams p,
            const ModCtx* __restrict__ mod) {
  const int T = V2 ? 2 : p.T;
  const i64 shift = p.base / 2;
  const i64 s0 = (i64)blockIdx.x * TILE;                      // wave-uniform
  auto red = [&](i64 d, int s) -> u64 {
    if constexpr (Q32) {
      const u32 q = (u32)mod[s].q, mu = (u32)(mod[s].mu >> 32);
      const u32 ad = d >= 0 ? (u32)d : (u32)(-d);
      u32 x = ad - __umulhi(ad, mu) * q;                      // [0, 2q)
      x = min(x, x - q);
      return (u64)((d < 0 && x != 0) ? q - x : x);
    } else {
      return reduce_signed(d, mod[s]);
    }
  };
  auto put = [&](i64 j, i64 r, i64 d) {                       // digit j of row r: reduced into every component
    i64* o = digits + (j * rows + r) * T;
    if constexpr (V2) {
      dec_u64x2 w; w.x = red(d, 0); w.y = red(d, 1);
      *reinterpret_cast<dec_u64x2*>(o) = w;
    } else {
      for (int s = 0; s < T; ++s) o[s] = (i64)red(d, s);
    }
  };
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    const i64 r = s0 + (u32)e * 256u + threadIdx.x;
    if (r >= rows) continue;
    i64 j = 0;
    for (int t = 0; t < T; ++t) {
      const u64 qt = mod[t].q;
      const u64 x = canon_in(c[r * T + t], qt);
      i64 v = (2 * x < qt) ? (i64)x : (i64)x - (i64)qt;          // lift: [-q/2, q/2)
      for (int k = 0; k + 1 < p.k[t]; ++k, ++j) {                 // centred base-b digits
        const i64 a = v + shift;
        const i64 qd = floor_div(a, p);
        const i64 rem = a - qd * p.base - shift;
        v = qd;
        put(j, r, rem);
      }
      put(j, r, v);                                              // last digit: what is left
      ++j;
    }
  }
}

hipError_t launch_decompose(hipStream_t s, const i64* c, i64* digits, i64 B, i64 n, const DecompParams& p,
                            const ModCtx* mod, bool q32) {
  const i64 rows = B * n;
  if (rows == 0) return hipSuccess;
  unsigned blocks;
  if (!tiles_for(rows, &blocks)) return hipErrorInvalidValue;
  const bool fast = q32 && p.base <= ((i64)1 << 31);
  const bool v2 = p.T == 2 && (((uintptr_t)digits) & 15) == 0;
#define LOLHIP_DEC(QQ, VV) hipLaunchKernelGGL((k_decompose<QQ, VV>), dim3(blocks), dim3(256), 0, s, c, digits, rows, p, mod)
  if (fast) { if (v2) LOLHIP_DEC(true, true); else LOLHIP_DEC(true, false); }
  else { if (v2) LOLHIP_DEC(false, true); else LOLHIP_DEC(false, false); }
#undef LOLHIP_DEC
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// knapsack
// ---------------------------------------------------------------------------------------
// Q32: every modulus below 2^29 (the reference's own): products are below 2^58, 32 of them fit one 64-bit
// accumulator — one v_mad_u64_u32 per term and a single-word Barrett step at the end instead of 128-bit sums
// and a two-step division.
template <int K, bool Q32>
__global__ void __launch_bounds__(256)
k_knapsack(const i64* __restrict__ xs, int L, const i64* __restrict__ hint, const i64* addend, i64* out, i64 total,
           u32 per, int T, const ModCtx* __restrict__ mod) {
  const i64 s0 = (i64)blockIdx.x * TILE;                      // wave-uniform
  const u32 r_s = (u32)((u64)s0 % per);
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    const u32 l = (u32)e * 256u + threadIdx.x;
    const i64 g = s0 + l;
    if (g >= total) continue;
    u32 r = r_s + l;
    if (r >= per) r %= per;
    const ModCtx mc = mod[r % (u32)T];
    using Acc = std::conditional_t<Q32, u64, u128>;
    Acc acc[K];
#pragma unroll
    for (int k = 0; k < K; ++k) acc[k] = 0;
    auto fold = [&](Acc a) -> u64 {
      if constexpr (Q32) {
        const u64 rr = (u64)a - __umul64hi((u64)a, mc.mu) * mc.q;       // [0, 2q)
        return rr >= mc.q ? rr - mc.q : rr;
      } else {
        return reduce128((u64)((u128)a >> 64), (u64)a, mc);
      }
    };
    for (int j = 0; j < L; ++j) {
      const u64 x = canon_in(xs[(i64)j * total + g], mc.q);
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const u64 h = canon_in(hint[((i64)j * K + k) * per + r], mc.q);
        if constexpr (Q32) acc[k] += (u64)(u32)x * (u32)h;           // < 2^58
        else acc[k] += (u128)x * h;                                  // each term < q^2 < 2^124
      }
      if ((j & (Q32 ? 31 : 7)) == (Q32 ? 31 : 7)) {                  // keep the sum inside the accumulator
#pragma unroll
        for (int k = 0; k < K; ++k) acc[k] = fold(acc[k]);
      }
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
      u64 v = fold(acc[k]);
      if (addend) v = addmod(v, canon_in(addend[(i64)k * total + g], mc.q), mc.q);
      out[(i64)k * total + g] = (i64)v;
    }
  }
}

hipError_t launch_knapsack(hipStream_t s, const i64* xs, int L, const i64* hint, int K, const i64* addend, i64* out,
                           i64 B, i64 n, int T, const ModCtx* mod, bool q32) {
  const i64 total = B * n * T;
  if (total == 0) return hipSuccess;
  unsigned blocks;
  if (!tiles_for(total, &blocks)) return hipErrorInvalidValue;
  const dim3 grid(blocks), block(256);
  const u32 per = (u32)(n * T);
#define LOLHIP_KS(KK, QQ) hipLaunchKernelGGL((k_knapsack<KK, QQ>), grid, block, 0, s, xs, L, hint, addend, out, total, per, T, mod)
  switch (K * 2 + (q32 ? 1 : 0)) {
    case 2: LOLHIP_KS(1, false); break;
    case 3: LOLHIP_KS(1, true); break;
    case 4: LOLHIP_KS(2, false); break;
    case 5: LOLHIP_KS(2, true); break;
    case 6: LOLHIP_KS(3, false); break;
    case 7: LOLHIP_KS(3, true); break;
    default: return hipErrorInvalidValue;
  }
#undef LOLHIP_KS
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// rescale: drop component 0
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_rescale(const i64* __restrict__ c, i64* __restrict__ out, i64 rows, RescaleParams p, const ModCtx* __restrict__ mod) {
  const int To = p.T - 1;
  const i64 total = rows * To;
  const u64 qa = mod[0].q;
  const i64 s0 = (i64)blockIdx.x * TILE;                      // wave-uniform
  const i64 row_s = s0 / To;
  const u32 t_s = (u32)(s0 - row_s * To);
#pragma unroll
  for (int k = 0; k < EPT; ++k) {
    const u32 l = (u32)k * 256u + threadIdx.x;
    const i64 g = s0 + l;
    if (g >= total) continue;
    const u32 dr = (t_s + l) / (u32)To;
    const i64 r = row_s + dr;
    const int s = (int)(t_s + l - dr * (u32)To) + 1;
    const ModCtx ms = mod[s];
    const u64 a = canon_in(c[r * p.T], qa);
    const i64 z = (2 * a < qa) ? (i64)a : (i64)a - (i64)qa;      // lift a
    const u64 b = canon_in(c[r * p.T + s], ms.q);
    out[g] = (i64)mulmod(submod(b, reduce_signed(z, ms), ms.q), p.qa_inv[s], ms);
  }
}

hipError_t launch_rescale(hipStream_t s, const i64* c, i64* out, i64 B, i64 n, const RescaleParams& p,
                          const ModCtx* mod) {
  const i64 rows = B * n;
  if (rows == 0 || p.T < 2) return hipSuccess;
  unsigned blocks;
  if (!tiles_for(rows * (p.T - 1), &blocks)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_rescale, dim3(blocks), dim3(256), 0, s, c, out, rows, p, mod);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// coeffs: the phi(m')/phi(m) coefficient vectors of an O_m' element over O_m, powerful or
// decoding basis (class Tensor `coeffs`, Tensor.hs:174; CPP/Extension.hs:90-93; table
// extIndicesCoeffs, Tensor.hs:472-477).  A pure permutation: one read, one write.
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_coeffs(i64* __restrict__ out, const i64* __restrict__ in, const int32_t* __restrict__ idx, i64 B, u32 n_lo, u32 n_hi,
         int T, const ModCtx* __restrict__ mod) {
  const i64 slab = B * n_lo * T;                    // one output vector over the whole batch
  const i64 total = B * n_hi * T;
  const u32 per = n_lo * (u32)T;
  const i64 s0 = (i64)blockIdx.x * TILE;                      // wave-uniform
  const i64 i1_s = s0 / slab;
  const i64 w_s = s0 - i1_s * slab;                 // position inside output vector i1: (b, i0, t)
  const i64 b_s = w_s / per;
  const u32 r_s = (u32)(w_s - b_s * per);
#pragma unroll
  for (int k = 0; k < EPT; ++k) {
    const u32 l = (u32)k * 256u + threadIdx.x;
    const i64 g = s0 + l;
    if (g >= total) continue;
    u32 r = r_s + l;
    const u32 db = r / per;
    r -= db * per;
    i64 b = b_s + db, i1 = i1_s;
    while (b >= B) { b -= B; ++i1; }                // a tile may run into the next output vector
    const u32 i0 = r / (u32)T, t = r - i0 * (u32)T;
    const int32_t e = idx[i1 * n_lo + i0];
    out[g] = (i64)canon_in(in[(b * n_hi + e) * T + t], mod[t].q);
  }
}

hipError_t launch_coeffs(hipStream_t s, i64* out, const i64* in, const int32_t* idx, i64 B, i64 n_lo, i64 n_hi, int T,
                         const ModCtx* mod) {
  const i64 total = B * n_hi * T;
  if (total == 0) return hipSuccess;
  unsigned blocks;
  if (!tiles_for(total, &blocks)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_coeffs, dim3(blocks), dim3(256), 0, s, out, in, idx, B, (u32)n_lo, (u32)n_hi, T, mod);
  return hipGetLastError();
}

}  // namespace lolhip
