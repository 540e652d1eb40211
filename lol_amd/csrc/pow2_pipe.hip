// lol_amd/csrc/pow2_pipe.hip — k_pow2_pipe<L, AR, SQ>: the fused poly-mul c = crtInv(crt a * crt b) of the
// 32-bit arithmetic classes (every modulus < 2^31: the reference's own valid domain, types.h:79-84) as a
// PERSISTENT, software-pipelined kernel.  Replaces crt.cpp:562-581 + mul.cpp:14-30 over a batch.
//
// Why.  k_pow2<L, 2, AR> runs load a -> transform -> load b -> transform -> product -> inverse -> store
// inside one workgroup, and the four workgroups of a CU move through those phases together: measured
// T = T_mem + T_valu - 0.05 ms, although either alone is below 0.19 ms.  Here a workgroup walks
// a strided list of polynomials, and the operand it will need NEXT is on its way while the current one is
// transformed:
//   * operands arrive by LDS-DMA (buffer_load_dword ... lds): no VGPR destination, so nothing is held in
//     registers across a transform.  Only the LOW dword of every int64 coefficient is fetched into LDS: for
//     q < 2^31 a representative in (-q, q) is determined by it (bit 31 = sign), so the 32-bit staging buffer
//     is half the operand (32 KiB at n = 8192) and two workgroups fit a CU beside their working buffers;
//   * every wave fetches exactly the 1024 coefficients it reads back (a wave's block of the powerful basis),
//     so the hand-off needs no workgroup barrier: the issuing wave's own vmcnt wait orders its DMA before its
//     ds_reads (MI355X_MICROARCH.md, Two waves per SIMD, item 7);
//   * the staged operand is read straight in the transform's FIRST register layout (16 consecutive
//     coefficients per thread, four ds_read_b128; rows of 64 words padded by 4 keep them conflict-free), so
//     the global-layout -> W0 LDS transpose of k_pow2 disappears;
//   * both LDS twiddle copies (forward and inverse entries [16, 512)) stay resident for the whole launch.
// Timeline of one wave for polynomial p (G = grid size):
//   wait(a_p landed) -> read a_p -> issue DMA b_p -> transform a -> wait(b_p) -> barrier -> read b_p ->
//   issue DMA a_{p+G} -> transform b -> product -> inverse -> store c_p (16 B per lane).
// vmcnt retires in issue order, so a twiddle load issued after a DMA waits for that DMA to land (measured:
// 14-18 k cycles per forward transform, profiles/r03_pipe_stamps.txt): the forward transforms' vector-memory
// twiddles are fetched once per polynomial, ahead of DMA b_p, and held in registers through both transforms.
// hipcc (ROCm 7.2) cannot count past an LDS-DMA: with one in flight it waits vmcnt(0) at the next use of any
// ordinary load (cdna_hip_programming.md, Pipelining across barriers).  So the DMA and those twiddle loads are
// inline asm and their waits are counted by hand: per polynomial a wave issues 8 + 14 twiddle loads, then 16 DMA
// pieces; the level-10 twiddles are ready at vmcnt(14 + 16), the rest at vmcnt(16), the DMA at vmcnt(0).  Every
// asm vector-memory statement opens with s_nop 4: its SGPR operands may just have been restored from a spill lane
// by v_readlane, and hipcc pads nothing inside an asm string.
// (LOLHIP_PAIR_BFLY=1: the pair/mad butterflies of pow2_impl.h — built and measured in this kernel: 16 % fewer VALU
// instructions, but at its 113-123 VGPRs the doubled data registers spill 10 of them to scratch, whose vector-memory
// traffic lands in the hand-counted vmcnt windows: 0.253 -> 0.264 ms at q < 2^27, no change at q < 2^30;
// profiles/r03_ab_pair_bfly_negative.txt.  Off.)
#include "pow2_impl.h"
#include <cstdio>
#include <cstdlib>

namespace lolhip {

typedef __attribute__((address_space(3))) void lds_void_t;

// ---- vector memory the compiler does not see (counted by hand, see the header) -----------------------------
__device__ __forceinline__ void dma_dword_asm(rsrc_t r, u32 lds_byte, u32 voff, u32 soff) {
  u32 keep;
  asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dword %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "s"(lds_byte), "v"(voff), "s"(r), "s"(soff));
}
// NO "memory" clobbers in this file: with one in sight hipcc no longer proves the twiddle tables unmodified and
// fetches the wave-uniform twiddles (levels 1-4) with VECTOR loads instead of s_load — which then queue behind
// the DMA.  Ordering comes from asm volatile (volatile statements keep their order) and register operands.
__device__ __forceinline__ u32x2 load_b64_asm(rsrc_t r, u32 voff, u32 soff) {
  u32x2 x;
  asm volatile("s_nop 4\n\tbuffer_load_dwordx2 %0, %1, %2, %3 offen" : "=v"(x) : "v"(voff), "s"(r), "s"(soff));
  return x;
}
// the distinct twiddles of the level on register bit K of layout A, as raw (w, w') pairs still in flight
template <Lay A, int K> struct TwRaw { u32x2 r[tw_distinct(A, K)]; };
template <Lay A, int K>
__device__ __forceinline__ void tw_issue(TwRaw<A, K>& raw, rsrc_t tab, int xt) {
  constexpr int beta = A.reg[K];
  const u32 voff = (u32)(xt & ((1 << beta) - 1)) * 8u;
  int ord = 0, j = 0;
#pragma unroll
  for (int e = 0; e < E; ++e) {
    if (e & (1 << K)) continue;
    if (level_tab<A, K>.slot[e] == ord) raw.r[j++] = load_b64_asm(tab, voff, (u32)level_tab<A, K>.cidx[e] * 8u);
    ++ord;
  }
}
// after the counted wait: no consumer may be scheduled above this point, then rename into the level's slots
template <Lay A, int K>
__device__ __forceinline__ void tw_landed(LevelTwT<u32>& t, TwRaw<A, K>& raw) {
#pragma unroll
  for (int j = 0; j < tw_distinct(A, K); ++j) asm volatile("" : "+v"(raw.r[j]));
  int ord = 0, j = 0;
#pragma unroll
  for (int e = 0; e < E; ++e) {
    if (e & (1 << K)) continue;
    if (level_tab<A, K>.slot[e] == ord) { t.w[ord] = raw.r[j].x; t.wp[ord] = raw.r[j].y; ++j; }
    ++ord;
  }
}
// the forward transform's vector-memory twiddles (fwd_transform's TOP provider): level 10 (second lane swap)
// and the levels after the cross-wave exchange, issued in this order, then NDMA pieces of LDS-DMA behind them
template <int L, int NDMA> struct PipeTop {
  using S = Sched<L, true>;
  static constexpr Lay AL = S::w1b(), AG = S::g();
  static constexpr int NG = (S::G_K0 <= 1 ? tw_distinct(AG, 1) : 0) + (S::G_K0 <= 2 ? tw_distinct(AG, 2) : 0) + tw_distinct(AG, 3);
  TwRaw<AL, 2> r10;
  TwRaw<AG, 1> rg1;
  TwRaw<AG, 2> rg2;
  TwRaw<AG, 3> rg3;
  LevelTwT<u32> l10, g[R];
  __device__ __forceinline__ void issue(rsrc_t tab, int tau) {
    static_assert(S::NSWAP == 2 && S::HAS_G && S::G_K0 >= 1, "n = 4096 or 8192");
    tw_issue(r10, tab, xthr<AL>(tau));
    const int xg = xthr<AG>(tau);
    if constexpr (S::G_K0 <= 1) tw_issue(rg1, tab, xg);
    if constexpr (S::G_K0 <= 2) tw_issue(rg2, tab, xg);
    tw_issue(rg3, tab, xg);
  }
  __device__ __forceinline__ void wait_l10() {
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NG + NDMA));
    tw_landed(l10, r10);
  }
  __device__ __forceinline__ void wait_g() {
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NDMA));
    if constexpr (S::G_K0 <= 1) tw_landed(g[1], rg1);
    if constexpr (S::G_K0 <= 2) tw_landed(g[2], rg2);
    tw_landed(g[3], rg3);
  }
};

constexpr int PIPE_ROW = 68;                                    // LDS words per 64 staged coefficients
constexpr int pipe_lds_words(int n) { return (n + n / 16) + 2 * twl_words(n) + (n / 64) * PIPE_ROW; }

// SQ: squaring (a == b): one forward transform.  A compile-time flag, not a run-time branch: hipcc's counted
// vmcnt waits need the same vector-memory issue sequence on every path, or they fall back to vmcnt(0) and a
// twiddle wait drains the DMA in flight (the same holds for the prefetch at the tail: see `pn`).
template <int L, int AR, bool SQ>
__global__ void __launch_bounds__(1 << (L - R), 4)
k_pow2_pipe(i64* __restrict__ c, const i64* a, const i64* b, i64 B, const u32* __restrict__ tw_fwd, const u32* __restrict__ tw_inv,
            const u32* __restrict__ scale, const ModCtx* __restrict__ mod) {
  static_assert(AR >= 2 && L >= 12, "32-bit classes; one polynomial per workgroup of whole waves");
  using S = Sched<L, true>;
  using V = u32;
  constexpr int n = 1 << L;
  constexpr int NT = 1 << (L - R);
  constexpr int WORK = n + n / 16, TWW = twl_words(n);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  V* lds = reinterpret_cast<V*>(smem);
  V* lds_twf = lds + WORK;
  V* lds_twi = lds_twf + TWW;
  V* stage = lds_twi + TWW;
  const int tau = threadIdx.x;
  const int lane = tau & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tau >> 6);

  QK32 qk(mod[0], std::true_type{});
#ifndef LOLHIP_PIPE_Q2_SGPR
  // 2q is an operand of three additions per butterfly: in a VGPR they are plain VOP2 instructions (1.2 ns per
  // wave-instruction at 4 waves per SIMD against 1.96 with an SGPR operand, profiles/r02_microbench_ops.txt)
  asm volatile("" : "+v"(qk.q2));
#endif
  TwCtxT<V> twf;
  twf.fwd = __builtin_amdgcn_make_buffer_rsrc((void*)tw_fwd, 0, (u32)n * 8u, 0x00020000);
  twf.inv = __builtin_amdgcn_make_buffer_rsrc((void*)tw_inv, 0, (u32)n * 8u, 0x00020000);
  twf.comp = 0;
  twf.pf = tw_fwd;
  twf.pi = tw_inv;
  twf.lds_tw = lds_twf;
  twf.sc0 = scale[0]; twf.sc1 = scale[1]; twf.l1w = scale[2]; twf.l1wp = scale[3];
  TwCtxT<V> twi = twf;
  twi.lds_tw = lds_twi;
  tw_fill_lds<NT>(lds_twf, twf.fwd, 0, n, tau);
  tw_fill_lds<NT>(lds_twi, twf.inv, 0, n, tau);

  // the forward table as the asm loads see it: built from a LAUNDERED copy of the pointer.  Handing hipcc's own
  // tw_fwd-derived descriptor to an asm statement makes it give up on s_load for the wave-uniform twiddles of that
  // table (levels 1-4 came back as global_load_dwordx4, queued behind the DMA)
  const u32* tw_fwd_opaque = tw_fwd;
  asm volatile("" : "+s"(tw_fwd_opaque));
  const rsrc_t tab_asm = __builtin_amdgcn_make_buffer_rsrc((void*)tw_fwd_opaque, 0, (u32)n * 8u, 0x00020000);
  // this wave's staging rows and this lane's 16 consecutive coefficients in them
  V* wst = stage + wv * (16 * PIPE_ROW);
  const V* rd = wst + (lane >> 2) * PIPE_ROW + (lane & 3) * 16;
  const u32 wst_byte = __builtin_amdgcn_readfirstlane((u32)(size_t)wst);     // LDS byte address of this wave's rows
  auto dma = [&](const i64* src, i64 poly) {
    const rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)(src + poly * n), 0, (u32)n * 8u, 0x00020000);
#pragma unroll
    for (int ch = 0; ch < 16; ++ch)
      dma_dword_asm(r, wst_byte + (u32)(ch * PIPE_ROW * 4), (u32)lane * 8u, (u32)(wv * 8192 + ch * 512));
  };
  // staged low dwords -> registers in layout W0; (-q, q) representatives -> (0, 2q) by adding q mod 2^32.  One asm
  // statement: wait for this wave's DMA (everything it has in flight), read its rows, wait for the reads — after
  // it the rows are free for the next DMA
  const u32 rd_byte = (u32)(size_t)rd;
  auto take = [&](V (&v)[E]) {
    u32x4 w0, w1, w2, w3;
    asm volatile("s_waitcnt vmcnt(0)\n\tds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:32\n\t"
                 "ds_read_b128 %3, %4 offset:48\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(w0), "=&v"(w1), "=&v"(w2), "=&v"(w3) : "v"(rd_byte));
    const u32x4 w[4] = {w0, w1, w2, w3};
#pragma unroll
    for (int k = 0; k < 4; ++k) { v[4 * k + 0] = w[k].x; v[4 * k + 1] = w[k].y; v[4 * k + 2] = w[k].z; v[4 * k + 3] = w[k].w; }
#pragma unroll
    for (int e = 0; e < E; ++e) v[e] = v[e] + qk.q;          // (-q, q) -> (0, 2q): inside every 32-bit class's forward range
  };

  constexpr Lay LIO = S::io();
  const u32 off_io = (u32)xthr<LIO>(tau) * 8u;
  const i64 G = gridDim.x;
  i64 p = blockIdx.x;
  if (p < B) dma(a, p);
  __syncthreads();                                             // the twiddle copies are visible to every wave
#ifdef LOLHIP_STAMPS      // in-kernel clock: shader cycles (s_memtime) against the 100 MHz counter (s_memrealtime)
  LH_STAMP(26);
  { unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory");
    if (g_stamp_buf && (threadIdx.x & 63) == 0) g_stamp_buf[((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 32 + 28] = t_; }
  unsigned long long* const stamp_all = g_stamp_buf;   // kernel-scope stamps record the FOURTH polynomial of every workgroup (steady state)
#endif
  for (; p < B; p += G) {
    V v[E], va[E];
#ifdef LOLHIP_STAMPS
    unsigned long long* const g_stamp_buf = (p == (i64)blockIdx.x + 3 * G) ? stamp_all : nullptr;   // shadows the global for LH_STAMP below
#endif
    LH_STAMP(0);
    take(v);                                                   // waits until a_p has landed (this wave's own rows)
    LH_STAMP(1);
    // this polynomial's vector-memory twiddles go out BEFORE the DMA (vmcnt retires in issue order); a and b share them
    PipeTop<L, 16> top;
    top.issue(tab_asm, tau);
    // the operand after this one: the next polynomial's a, or at the tail this polynomial's once more (harmless:
    // nobody reads it), so that every iteration issues the same vector-memory sequence
    const i64 pn = (p + G < B) ? p + G : p;
    if constexpr (SQ) dma(a, pn); else dma(b, p);
    fwd_transform<AR, L, S::w0(), 0, false, true, PipeTop<L, 16>>(v, lds, twf, tau, qk, &top);
#pragma unroll
    for (int e = 0; e < E; ++e) va[e] = park_fwd<AR>(v[e], qk);
    LH_STAMP(9);
    if constexpr (!SQ) {
      LH_STAMP(10);
      __syncthreads();                                         // every wave is past a's cross-wave reads (levels 11..L)
      LH_STAMP(11);
      take(v);                                                 // waits until b_p has landed
      dma(a, pn);
      fwd_transform<AR, L, S::w0(), 10, true, true, PipeTop<L, 16>>(v, lds, twf, tau, qk, &top);
    }
    LH_STAMP(19);
    const ModCtx mc = mod[0];
#pragma unroll
    for (int e = 0; e < E; ++e) v[e] = pmul<AR>(va[e], v[e], mc, qk);     // squaring: v still holds a-hat (lazy)
    LH_STAMP(22);
    inv_transform<AR, L, LIO, true>(v, lds, twi, tau, qk);
    LH_STAMP(24);
    const rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void*)(c + p * n), 0, (u32)n * 8u, 0x00020000);
    store_poly<LIO, true>(rc, off_io, 8u, [&](int e) { return (u64)canon_inv<AR>(v[e], qk); });
    LH_STAMP(25);
  }
  asm volatile("s_waitcnt vmcnt(0)");                         // no DMA may land in LDS after this workgroup has released it
#ifdef LOLHIP_STAMPS
  LH_STAMP(27);
  { unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory");
    if (g_stamp_buf && (threadIdx.x & 63) == 0) g_stamp_buf[((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 32 + 29] = t_; }
#endif
}

template <int L, int AR, bool SQ>
static hipError_t launch_pipe_L(const Pow2Launch& a) {
  constexpr int n = 1 << L;
  constexpr int NT = 1 << (L - R);
  const size_t lds_bytes = (size_t)pipe_lds_words(n) * sizeof(u32);
  static KernelDev tab[MAX_DEV];
  static int cus[MAX_DEV];
  hipError_t e = kernel_dev_setup(tab, [&]() -> hipError_t {
    int dev = 0;
    hipError_t r = hipGetDevice(&dev);
    if (r != hipSuccess) return r;
    r = hipDeviceGetAttribute(&cus[dev], hipDeviceAttributeMultiprocessorCount, dev);
    if (r != hipSuccess) return r;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pow2_pipe<L, AR, SQ>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  });
  if (e != hipSuccess) return e;
  int dev = 0;
  if ((e = hipGetDevice(&dev)) != hipSuccess) return e;
  // resident workgroups: LDS-limited, at most 4 waves per SIMD.  The 160 KiB of a CU are handed out as two halves
  // of 80 KiB and a workgroup's allocation does not straddle them (measured with tools/occ_probe.hip: 48 KiB
  // workgroups run two to a CU, not three; the occupancy query says three)
  int per_cu = 2 * (int)((80 * 1024) / lds_bytes);
  if (per_cu < 1) per_cu = 1;
  if (per_cu > 16 * 64 / NT) per_cu = 16 * 64 / NT;
  i64 grid = (i64)cus[dev] * per_cu;
  if (const char* g = getenv("LOLHIP_PIPE_GRID")) grid = atol(g);            // development: occupancy experiments
  if (getenv("LOLHIP_PIPE_INFO")) {
    int occ = -1;
    hipError_t oe = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, reinterpret_cast<const void*>(&k_pow2_pipe<L, AR, SQ>), NT, lds_bytes);
    fprintf(stderr, "k_pow2_pipe<%d,%d>: lds %zu B, per_cu %d, occupancy query %d (err %d), grid %lld\n", L, AR, lds_bytes, per_cu, occ, (int)oe, (long long)grid);
  }
  if (grid > a.B) grid = a.B;
  if (grid == 0) return hipSuccess;
  hipLaunchKernelGGL((k_pow2_pipe<L, AR, SQ>), dim3((unsigned)grid), dim3(NT), lds_bytes, a.stream, a.y, a.a, a.b, a.B,
                     static_cast<const u32*>(a.tw_fwd), static_cast<const u32*>(a.tw_inv), static_cast<const u32*>(a.scale), a.mod);
  return hipGetLastError();
}

// does the pipelined kernel take this fused poly-mul launch?  (one modulus below 2^31, n = 4096 or 8192, the
// output slab 16-byte aligned, and enough polynomials for every resident workgroup to have several)
bool pow2_pipe_ok(const Pow2Launch& a, bool forced) {
  // n = 4096 is instantiated and tested, but measured slower than k_pow2 there (0.257 against 0.240 ms per
  // 8192 x 4096): its workgroups are four waves, two to a CU — taken only when forced
  if (a.T != 1 || a.arith < 2 || (a.L != 13 && !(forced && a.L == 12))) return false;
  if (((uintptr_t)a.y & 15) || (((uintptr_t)a.a | (uintptr_t)a.b) & 7)) return false;
  return forced || a.B >= 2048;
}

template <bool SQ> static hipError_t launch_pipe_sq(const Pow2Launch& a) {
  switch (a.arith * 100 + a.L) {
    case 212: return launch_pipe_L<12, 2, SQ>(a);
    case 213: return launch_pipe_L<13, 2, SQ>(a);
    case 312: return launch_pipe_L<12, 3, SQ>(a);
    case 313: return launch_pipe_L<13, 3, SQ>(a);
    case 412: return launch_pipe_L<12, 4, SQ>(a);
    case 413: return launch_pipe_L<13, 4, SQ>(a);
    default: return hipErrorInvalidValue;
  }
}
hipError_t launch_pow2_pipe(const Pow2Launch& a) { return a.a == a.b ? launch_pipe_sq<true>(a) : launch_pipe_sq<false>(a); }

}  // namespace lolhip
#if defined(LOLHIP_STAMPS) && LOLHIP_STAMPS == 3      // diagnostic build only: phase stamps of the pipelined kernel
extern "C" __attribute__((visibility("default"))) int lolhip_debug_set_stamps(unsigned long long* dev) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(lolhip::g_stamp_buf), &dev, sizeof(dev));
}
#endif
