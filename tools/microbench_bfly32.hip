// tools/microbench_bfly32.hip — the 32-bit lazy butterflies (class 4: q < 2^27) in two forms, timed and checked:
//   P: production (pow2_impl.h bfly_fwd<4> / bfly_inv<4>): 7 / 9 instructions on 32-bit registers;
//   M: every residue in the LOW half of a 64-bit register pair (high half: don't care) so that v_mad_u64_u32 does the
//      multiply AND the add/subtract — x + w y - Q q as two chained mads, and 2x + 2q - X' as a mad by -1:
//      forward 5 instructions (v_mul_hi, 3 mads, v_lshl_add_u64), inverse 7.  Costs twice the data registers,
//      which the persistent pipelined poly-mul (pow2_pipe.hip: 4 waves per SIMD) has.
// Development tool / evidence for DESIGN.md section 8.  Build: see tools/Makefile.
#include "pow2_impl.h"

#include <stdio.h>
#include <algorithm>
#include <random>
#include <vector>
using namespace lolhip;
constexpr int ITER = 4096, CH = 16;

__device__ __forceinline__ u64 pair_of(u32 lo) { typedef u32 u32x2 __attribute__((ext_vector_type(2))); u32x2 t; t.x = lo; return __builtin_bit_cast(u64, t); }

struct KM { u32 q, q2, nq; };
__device__ __forceinline__ void fwd_m(u64& X, u64& Y, u32 w, u32 wp, const KM& k, u64 q2p) {
  u64 Xn, Z, cy; u32 Q;
  asm("v_mul_hi_u32 %[Q], %[wp], %[yl]\n\t"
      "v_mad_u64_u32 %[Xn], %[cy], %[w], %[yl], %[X]\n\t"
      "v_lshl_add_u64 %[Z], %[X], 1, %[q2]\n\t"
      "v_mad_u64_u32 %[Xn], %[cy], %[Q], %[nq], %[Xn]"
      : [Q] "=&v"(Q), [Xn] "=&v"(Xn), [Z] "=&v"(Z), [cy] "=&s"(cy)
      : [wp] "v"(wp), [yl] "v"((u32)Y), [w] "v"(w), [X] "v"(X), [q2] "s"(q2p), [nq] "s"(k.nq));
  u64 Yn;
  asm("v_mad_u64_u32 %[Yn], %[cy], %[xl], -1, %[Z]" : [Yn] "=v"(Yn), [cy] "=s"(cy) : [xl] "v"((u32)Xn), [Z] "v"(Z));
  X = Xn; Y = Yn;
}
// inverse (class 4): s = X + Y; d = X - Y + 2q; X' = min(s, s - 2q); Y' = w d - Q q
__device__ __forceinline__ void inv_m(u64& X, u64& Y, u32 w, u32 wp, const KM& k, u64 q2p) {
  u64 D, cy;
  asm("v_mad_u64_u32 %[D], %[cy], %[yl], -1, %[X]" : [D] "=v"(D), [cy] "=s"(cy) : [yl] "v"((u32)Y), [X] "v"(X));      // X - Y
  const u32 s = (u32)X + (u32)Y;
  const u32 xs = min(s, s - k.q2);
  const u32 d = (u32)D + k.q2;
  const u32 Q = __umulhi(wp, d);
  u64 T, Yn;
  asm("v_mad_u64_u32 %[T], %[cy], %[w], %[d], 0\n\t"
      "v_mad_u64_u32 %[Yn], %[cy], %[Q], %[nq], %[T]"
      : [T] "=&v"(T), [Yn] "=&v"(Yn), [cy] "=&s"(cy) : [w] "v"(w), [d] "v"(d), [Q] "v"(Q), [nq] "s"(k.nq));
  X = pair_of(xs); Y = Yn;
}

template <int V, bool INV>
__global__ void __launch_bounds__(256) k_thr(u32* out, unsigned long long* cyc, const u32* tw, u32 q, ModCtx mc) {
  const QK32 qk(mc, std::true_type{});
  const KM km{q, 2 * q, 0u - q};
  const u64 q2p = (u64)(2 * q);
  u32 y[CH]; u64 yp[CH];
  for (int i = 0; i < CH; i++) { y[i] = (u32)(((threadIdx.x + i + 1) * 0x9E3779B9u) >> 4) % q; yp[i] = pair_of(y[i]); }
  const u32* t = tw + 2 * (threadIdx.x & 63);
  u32 w = t[0], wp = t[1];
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
  for (int it = 0; it < ITER; ++it) {
    asm volatile("" : "+v"(w), "+v"(wp));
#pragma unroll
    for (int i = 0; i < CH; i += 2) {
      if constexpr (V == 0) { if constexpr (INV) bfly_inv<4>(y[i], y[i + 1], w, wp, qk); else { bfly_fwd<4>(y[i], y[i + 1], w, wp, qk); y[i] = min(y[i], y[i] - 8 * q); } }
      else { if constexpr (INV) inv_m(yp[i], yp[i + 1], w, wp, km, q2p); else { fwd_m(yp[i], yp[i + 1], w, wp, km, q2p); yp[i] = pair_of(min((u32)yp[i], (u32)yp[i] - 8 * q)); } }
    }
    if constexpr (V == 0) { const u32 r = y[0]; for (int i = 0; i + 1 < CH; i++) y[i] = y[i + 1]; y[CH - 1] = r; }
    else { const u64 r = yp[0]; for (int i = 0; i + 1 < CH; i++) yp[i] = yp[i + 1]; yp[CH - 1] = r; }
  }
  asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
  u32 acc = 0; for (int i = 0; i < CH; i++) acc += (V == 0) ? y[i] : (u32)yp[i];
  out[blockIdx.x * 256 + threadIdx.x] = acc;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
template <int V, bool INV>
__global__ void k_chk(u32* xy, const u32* tw, u32 q, int n, ModCtx mc) {
  int i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
  const QK32 qk(mc, std::true_type{});
  const KM km{q, 2 * q, 0u - q};
  u32 X = xy[2 * i], Y = xy[2 * i + 1];
  if constexpr (V == 0) { if constexpr (INV) bfly_inv<4>(X, Y, tw[2 * i], tw[2 * i + 1], qk); else bfly_fwd<4>(X, Y, tw[2 * i], tw[2 * i + 1], qk); }
  else { u64 a = pair_of(X), b = pair_of(Y); if constexpr (INV) inv_m(a, b, tw[2 * i], tw[2 * i + 1], km, (u64)(2 * q)); else fwd_m(a, b, tw[2 * i], tw[2 * i + 1], km, (u64)(2 * q)); X = (u32)a; Y = (u32)b; }
  xy[2 * i] = X; xy[2 * i + 1] = Y;
}

int main() {
  hipDeviceProp_t pr; (void)hipGetDeviceProperties(&pr, 0);
  const int cus = pr.multiProcessorCount, N = 1 << 20;
  const u32 q = 67239937u;
  const ModCtx mc = make_modctx(q);
  std::mt19937_64 rng(1);
  std::vector<u32> tw(2 * (size_t)N), in(2 * (size_t)N), xy(2 * (size_t)N);
  for (int i = 0; i < N; i++) { u32 w = (u32)(rng() % q); if (i % 101 == 0) w = q - 1 - (i % 3); tw[2 * i] = w; tw[2 * i + 1] = (u32)(((u64)w << 32) / q); }
  u32 *dtw, *dxy, *dout; unsigned long long* dcyc;
  (void)hipMalloc(&dtw, 8ull * N); (void)hipMalloc(&dxy, 8ull * N); (void)hipMalloc(&dout, (size_t)cus * 8 * 256 * 4); (void)hipMalloc(&dcyc, (size_t)cus * 8 * 4 * 8);
  (void)hipMemcpy(dtw, tw.data(), 8ull * N, hipMemcpyHostToDevice);
  printf("device %s CUs=%d q=%u\n", pr.name, cus, q);
  auto run = [&](auto kchk, auto kthr, const char* name, bool inv, u32 inB) {
    for (int i = 0; i < N; i++) { in[2 * i] = (u32)(rng() % ((u64)inB * q)); in[2 * i + 1] = inv ? (u32)(rng() % ((u64)inB * q)) : (u32)rng(); if (i % 97 == 0) in[2 * i] = inB * q - 1 - (i % 3); }
    (void)hipMemcpy(dxy, in.data(), 8ull * N, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(kchk, dim3(N / 256), dim3(256), 0, 0, dxy, dtw, q, N, mc);
    (void)hipMemcpy(xy.data(), dxy, 8ull * N, hipMemcpyDeviceToHost);
    long bad = 0; u32 maxo = 0;
    for (int i = 0; i < N; i++) {
      const u64 X = in[2 * i] % q, Y = in[2 * i + 1] % q, w = tw[2 * i]; u64 ex, ey;
      if (!inv) { u64 t = Y * w % q; ex = (X + t) % q; ey = (X + q - t) % q; } else { ex = (X + Y) % q; ey = ((X + q - Y) % q) * w % q; }
      if (xy[2 * i] % q != ex || xy[2 * i + 1] % q != ey) bad++;
      maxo = std::max(maxo, std::max(xy[2 * i], xy[2 * i + 1]));
    }
    double res[3], ns[3]; int wi = 0;
    hipFuncAttributes fa; (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(kthr));
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int wps : {2, 4, 8}) {
      const int blocks = cus * wps;
      hipLaunchKernelGGL(kthr, dim3(blocks), dim3(256), 0, 0, dout, dcyc, dtw, q, mc);
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(kthr, dim3(blocks), dim3(256), 0, 0, dout, dcyc, dtw, q, mc);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      std::vector<unsigned long long> h((size_t)blocks * 4);
      (void)hipMemcpy(h.data(), dcyc, h.size() * 8, hipMemcpyDeviceToHost);
      std::sort(h.begin(), h.end());
      res[wi] = (double)h[h.size() / 2] / ((double)ITER * (CH / 2) * wps);
      ns[wi++] = ms * 1e6 / ((double)ITER * (CH / 2) * wps);
    }
    printf("%-40s bad=%ld max/q=%.2f vgpr=%d | per wave-butterfly per SIMD at 2/4/8 waves per SIMD: ticks %5.1f %5.1f %5.1f  ns %5.1f %5.1f %5.1f\n", name, bad, (double)maxo / q, fa.numRegs, res[0], res[1], res[2], ns[0], ns[1], ns[2]);
  };
  run(k_chk<0, false>, k_thr<0, false>, "fwd production (7 instr + test csub)", false, 20);
  run(k_chk<1, false>, k_thr<1, false>, "fwd pair/mad   (5 instr + test csub)", false, 20);
  run(k_chk<0, true>, k_thr<0, true>, "inv production (9 instr)", true, 2);
  run(k_chk<1, true>, k_thr<1, true>, "inv pair/mad   (7 instr)", true, 2);
  return 0;
}
