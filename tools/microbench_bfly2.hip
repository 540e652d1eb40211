// tools/microbench_bfly2.hip — candidate formulations of the 64-bit (AR = 1) lazy butterflies,
// timed (cycles per wave-butterfly per SIMD at 4 and 8 waves/SIMD) and checked for exactness
// (congruence mod q and output range) on random + boundary inputs.  Development tool.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++20 -I../include -I../lol_amd/csrc -o microbench_bfly2 microbench_bfly2.hip
#include "pow2_impl.h"

#include <stdio.h>
#include <algorithm>
#include <random>
#include <vector>
using namespace lolhip;
typedef unsigned __int128 u128;
constexpr int ITER = 4096, CH = 8;

__device__ __forceinline__ u32 opq(u32 x) { asm volatile("" : "+v"(x)); return x; }
__device__ __forceinline__ u64 mk64(u32 lo, u32 hi) { return ((u64)hi << 32) | lo; }

// ---- quotient estimates ---------------------------------------------------------------
// A: drops the low halves of both cross terms: Q in {floor(wp*y/2^64) - 2 .. same}
__device__ __forceinline__ u64 quotA(u64 y, u64 wp) {
  const u32 ah = __umulhi(hi32(wp), lo32(y));
  const u32 bh = __umulhi(lo32(wp), hi32(y));
  return (u64)hi32(wp) * hi32(y) + ah + bh;
}
// B: cross terms summed exactly (65 bits): Q in {floor - 1, floor}
__device__ __forceinline__ u64 quotB(u64 y, u64 wp) {
  const u64 c1 = (u64)hi32(wp) * lo32(y);
  const u64 c2 = c1 + (u64)lo32(wp) * hi32(y);
  const u32 carry = c2 < c1;
  return (u64)hi32(wp) * hi32(y) + mk64(hi32(c2), carry);
}
// init + w*y - Q*q mod 2^64 (nq = -q)
__device__ __forceinline__ u64 wyqq(u64 y, u64 w, u64 Q, u64 nq, u64 init) {
  u64 t = (u64)lo32(w) * lo32(y) + init;
  t += (u64)lo32(Q) * lo32(nq);
  const u32 hh = lo32(w) * hi32(y) + hi32(w) * lo32(y) + lo32(Q) * hi32(nq) + hi32(Q) * lo32(nq);
  return mk64(lo32(t), hi32(t) + hh);
}
// the same with the high-word partial products kept in a v_mad_u64_u32 chain
__device__ __forceinline__ u64 wyqq_mad(u64 y, u64 w, u64 Q, u64 nq, u64 init) {
  u64 t = (u64)lo32(w) * lo32(y) + init;
  t += (u64)lo32(Q) * lo32(nq);
  u64 h = (u64)lo32(w) * hi32(y);
  h += (u64)hi32(w) * lo32(y);
  h += (u64)lo32(Q) * hi32(nq);
  h += (u64)hi32(Q) * lo32(nq);
  return mk64(lo32(t), hi32(t) + lo32(h));
}
// ---- conditional subtraction x >= m ? x - m : x  (negm = -m) ---------------------------
__device__ __forceinline__ u64 cs_sel(u64 x, u64 negm) { const u64 t = x + negm; return (i64)t < 0 ? x : t; }
__device__ __forceinline__ u64 cs_hi(u64 x, u64 negm) {                // 32-bit sign test of the high word
  const u64 t = x + negm;
  const bool neg = (int)opq(hi32(t)) < 0;
  return mk64(neg ? lo32(x) : lo32(t), neg ? hi32(x) : hi32(t));
}
__device__ __forceinline__ u64 cs_mask(u64 x, u64 m, u64 negm) {       // no VCC: sign mask
  const u64 t = x + negm;
  const u32 s = opq((u32)((int)hi32(t) >> 31));
  return t + mk64(lo32(m) & s, hi32(m) & s);
}

struct K { u64 q, nq, q2, nq2, q4, nq4, q4p1, q8, nq8; };

// V: 0 production; 1 C, quotA, select; 2 C, quotA, hi-word test; 3 C, quotA, mask;
//    4 as 1 with Y' = (2x + 4q + 1) + ~X'; 5 as 1 with the mad chain for the high word;
//    6 quotB (t < 3q) with select
template <int V>
__device__ __forceinline__ void fwd(u64& X, u64& Y, u64 w, u64 wp, const K& k, const QK& qk) {
  if constexpr (V == 0) { bfly_fwd<1>(X, Y, w, wp, qk); return; }
  u64 x;
  if constexpr (V == 2) x = cs_hi(X, k.nq4);
  else if constexpr (V == 3) x = cs_mask(X, k.q4, k.nq4);
  else x = cs_sel(X, k.nq4);
  const u64 Q = (V == 6) ? quotB(Y, wp) : quotA(Y, wp);
  const u64 xn = (V == 5) ? wyqq_mad(Y, w, Q, k.nq, x) : wyqq(Y, w, Q, k.nq, x);
  if constexpr (V == 4) {
    const u64 nx = mk64(opq(~lo32(xn)), opq(~hi32(xn)));
    Y = ((x << 1) + k.q4p1) + nx;
  } else {
    Y = ((x << 1) + k.q4) - xn;
  }
  X = xn;
}
// inverse: 0 production; 1 C quotA select; 2 C quotA hi-word test; 3 mask
template <int V>
__device__ __forceinline__ void inv(u64& X, u64& Y, u64 w, u64 wp, const K& k, const QK& qk) {
  if constexpr (V == 0) { bfly_inv<1>(X, Y, w, wp, qk); return; }
  const u64 s = X + Y;
  const u64 d = (X + k.q4) - Y;
  if constexpr (V == 2) X = cs_hi(s, k.nq4);
  else if constexpr (V == 3) X = cs_mask(s, k.q4, k.nq4);
  else X = cs_sel(s, k.nq4);
  Y = wyqq(d, w, quotA(d, wp), k.nq, 0);
}

template <int V, bool INV>
__global__ void __launch_bounds__(256) k_thr(u64* out, unsigned long long* cyc, const u64* tw, const K* kp) {
  const K k = *kp;
  const QK qk(k.q, std::true_type{});
  u64 y[CH];
  for (int i = 0; i < CH; i++) y[i] = (((threadIdx.x + i + 1) * 0x9E3779B97F4A7C15ull) >> 4) % k.q;
  const u64* t = tw + 2 * (threadIdx.x & 63);
  u64 w = t[0], wp = t[1];
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
  for (int it = 0; it < ITER; ++it) {
    asm volatile("" : "+v"(w), "+v"(wp));
#pragma unroll
    for (int i = 0; i < CH; i += 2) {
      if constexpr (INV) inv<V>(y[i], y[i + 1], w, wp, k, qk); else fwd<V>(y[i], y[i + 1], w, wp, k, qk);
    }
    // rotate so the chains mix (X of one feeds Y of the next)
    const u64 r = y[0];
#pragma unroll
    for (int i = 0; i + 1 < CH; i++) y[i] = y[i + 1];
    y[CH - 1] = r;
  }
  asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
  u64 acc = 0; for (int i = 0; i < CH; i++) acc += y[i];
  out[blockIdx.x * 256 + threadIdx.x] = acc;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
template <int V, bool INV>
__global__ void k_chk(u64* xy, const u64* tw, const K* kp, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
  const K k = *kp; const QK qk(k.q, std::true_type{});
  u64 X = xy[2 * i], Y = xy[2 * i + 1];
  if constexpr (INV) inv<V>(X, Y, tw[2 * i], tw[2 * i + 1], k, qk); else fwd<V>(X, Y, tw[2 * i], tw[2 * i + 1], k, qk);
  xy[2 * i] = X; xy[2 * i + 1] = Y;
}

struct Ctx { int cus; u64 q; K* dk; u64 *dtw, *dxy, *dout; unsigned long long* dcyc; std::vector<u64> tw; int N; };

template <int V, bool INV>
void run(Ctx& c, const char* name, u64 inBX, u64 inBY, u64 outB) {
  const u64 q = c.q; const int N = c.N;
  std::mt19937_64 rng(7 + V);
  std::vector<u64> in(2 * (size_t)N), xy(2 * (size_t)N);
  auto draw = [&](u64 boundq, int i) -> u64 {                 // boundq = 0: any 64-bit value
    u64 r = rng();
    if (boundq == 0) return (i % 97 == 0) ? ~0ull - (r % 3) : r;
    return (i % 97 == 0) ? (boundq * q - 1 - (r % 3)) : (i % 89 == 0 ? r % 3 : (u64)(((u128)r * (boundq * q)) >> 64));
  };
  for (int i = 0; i < N; i++) { in[2 * i] = draw(inBX, i); in[2 * i + 1] = draw(inBY, i + 31); }
  (void)hipMemcpy(c.dxy, in.data(), 16ull * N, hipMemcpyHostToDevice);
  hipLaunchKernelGGL((k_chk<V, INV>), dim3(N / 256), dim3(256), 0, 0, c.dxy, c.dtw, c.dk, N);
  (void)hipMemcpy(xy.data(), c.dxy, 16ull * N, hipMemcpyDeviceToHost);
  long bad = 0; u64 maxo = 0;
  for (int i = 0; i < N; i++) {
    const u64 X = in[2 * i] % q, Y = in[2 * i + 1] % q, w = c.tw[2 * i]; u64 ex, ey;
    if (!INV) { u64 t = (u64)((u128)Y * w % q); ex = (X + t) % q; ey = (X + q - t) % q; }
    else { ex = (X + Y) % q; ey = (u64)((u128)((X + q - Y) % q) * w % q); }
    if (xy[2 * i] % q != ex || xy[2 * i + 1] % q != ey || xy[2 * i] >= outB * q || xy[2 * i + 1] >= outB * q) bad++;
    maxo = std::max(maxo, std::max(xy[2 * i], xy[2 * i + 1]));
  }
  double res[3], ns[3];
  int wi = 0;
  hipFuncAttributes fa; (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&k_thr<V, INV>));
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int wps : {2, 4, 8}) {
    const int blocks = c.cus * wps;
    hipLaunchKernelGGL((k_thr<V, INV>), dim3(blocks), dim3(256), 0, 0, c.dout, c.dcyc, c.dtw, c.dk);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k_thr<V, INV>), dim3(blocks), dim3(256), 0, 0, c.dout, c.dcyc, c.dtw, c.dk);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h((size_t)blocks * 4);
    (void)hipMemcpy(h.data(), c.dcyc, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    res[wi] = (double)h[h.size() / 2] / ((double)ITER * (CH / 2) * wps);
    ns[wi++] = ms * 1e6 / ((double)ITER * (CH / 2) * wps);
  }
  printf("%-44s bad=%ld max/q=%.3f vgpr=%d | per wave-bfly per SIMD at 2/4/8 launched waves per SIMD: ticks %5.1f %5.1f %5.1f  ns %5.1f %5.1f %5.1f\n", name, bad, (double)maxo / q, fa.numRegs, res[0], res[1], res[2], ns[0], ns[1], ns[2]);
}

int main() {
  hipDeviceProp_t pr; (void)hipGetDeviceProperties(&pr, 0);
  Ctx c; c.cus = pr.multiProcessorCount; c.q = 1152921504606994433ull; c.N = 1 << 20;
  const u64 q = c.q;
  K hk{q, 0 - q, 2 * q, 0 - 2 * q, 4 * q, 0 - 4 * q, 4 * q + 1, 8 * q, 0 - 8 * q};
  (void)hipMalloc(&c.dk, sizeof(K)); (void)hipMemcpy(c.dk, &hk, sizeof(K), hipMemcpyHostToDevice);
  std::mt19937_64 rng(1);
  c.tw.resize(2 * (size_t)c.N);
  for (int i = 0; i < c.N; i++) { u64 w = rng() % q; if (i % 101 == 0) w = q - 1 - (i % 3); c.tw[2 * i] = w; c.tw[2 * i + 1] = (u64)(((u128)w << 64) / q); }
  (void)hipMalloc(&c.dtw, 16ull * c.N); (void)hipMalloc(&c.dxy, 16ull * c.N);
  (void)hipMemcpy(c.dtw, c.tw.data(), 16ull * c.N, hipMemcpyHostToDevice);
  (void)hipMalloc(&c.dout, (size_t)c.cus * 8 * 256 * 8); (void)hipMalloc(&c.dcyc, (size_t)c.cus * 8 * 4 * 8);
  printf("device %s CUs=%d q=%llu\n", pr.name, c.cus, (unsigned long long)q);
  // forward: X < 8q, Y any 64-bit value; outputs < 8q
  run<0, false>(c, "F0 production (asm blocks)", 8, 0, 8);
  run<1, false>(c, "F1 C, quotA, csub select", 8, 0, 8);
  run<2, false>(c, "F2 C, quotA, csub hi-word test", 8, 0, 8);
  run<3, false>(c, "F3 C, quotA, csub sign mask", 8, 0, 8);
  run<4, false>(c, "F4 C, quotA, select, Y'=z+~X'+1", 8, 0, 8);
  run<5, false>(c, "F5 C, quotA, select, mad chain high word", 8, 0, 8);
  run<6, false>(c, "F6 C, quotB (t<3q), select", 8, 0, 7);
  // inverse: X, Y < 4q; outputs < 4q
  run<0, true>(c, "G0 production (asm blocks)", 4, 4, 4);
  run<1, true>(c, "G1 C, quotA, csub select", 4, 4, 4);
  run<2, true>(c, "G2 C, quotA, csub hi-word test", 4, 4, 4);
  run<3, true>(c, "G3 C, quotA, csub sign mask", 4, 4, 4);
  return 0;
}
