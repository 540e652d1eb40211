// lol_amd/csrc/pow2_pipe.hip — k_pow2_pipe<L, AR>: the fused poly-mul c = crtInv(crt a * crt b) of the
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
// vmcnt counts in issue order, so a wait for a twiddle load issued after a DMA also waits for that DMA; by
// then (most of a transform later) it has landed.
#include "pow2_impl.h"

namespace lolhip {

typedef __attribute__((address_space(3))) void lds_void_t;

constexpr int PIPE_ROW = 68;                                    // LDS words per 64 staged coefficients
constexpr int pipe_lds_words(int n) { return (n + n / 16) + 2 * twl_words(n) + (n / 64) * PIPE_ROW; }

template <int L, int AR>
__global__ void __launch_bounds__(1 << (L - R), 4)
k_pow2_pipe(i64* c, const i64* a, const i64* b, i64 B, const u32* __restrict__ tw_fwd, const u32* __restrict__ tw_inv,
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

  const QK32 qk(mod[0], std::true_type{});
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

  // this wave's staging rows and this lane's 16 consecutive coefficients in them
  V* wst = stage + wv * (16 * PIPE_ROW);
  const V* rd = wst + (lane >> 2) * PIPE_ROW + (lane & 3) * 16;
  auto dma = [&](const i64* src, i64 poly) {
    const rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)(src + poly * n), 0, (u32)n * 8u, 0x00020000);
#pragma unroll
    for (int ch = 0; ch < 16; ++ch)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_t*)(wst + ch * PIPE_ROW), 4, (u32)lane * 8u,
                                               (u32)(wv * 8192 + ch * 512), 0, 0);
  };
  // staged low dwords -> registers in layout W0; (-q, q) representatives (bit 31 = sign) -> [0, q)
  auto take = [&](V (&v)[E]) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const u32x4 w = *reinterpret_cast<const u32x4*>(rd + 4 * k);
      v[4 * k + 0] = w.x; v[4 * k + 1] = w.y; v[4 * k + 2] = w.z; v[4 * k + 3] = w.w;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the rows are free for the next DMA
#pragma unroll
    for (int e = 0; e < E; ++e) v[e] = v[e] + (qk.q & (u32)((int)v[e] >> 31));
  };

  constexpr Lay LIO = S::io();
  const u32 off_io = (u32)xthr<LIO>(tau) * 8u;
  const i64 G = gridDim.x;
  const bool square = (a == b);
  i64 p = blockIdx.x;
  if (p < B) dma(a, p);
  __syncthreads();                                             // the twiddle copies are visible to every wave
  for (; p < B; p += G) {
    V v[E], va[E];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // a_p has landed (this wave's own rows)
    take(v);
    if (!square) dma(b, p);
    else if (p + G < B) dma(a, p + G);
    fwd_transform<AR, L, S::w0(), 0, false, true>(v, lds, twf, tau, qk);
#pragma unroll
    for (int e = 0; e < E; ++e) va[e] = park_fwd<AR>(v[e], qk);
    if (!square) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // b_p has landed
      __syncthreads();                                         // every wave is past a's cross-wave reads (levels 11..L)
      take(v);
      if (p + G < B) dma(a, p + G);
      fwd_transform<AR, L, S::w0(), 10, false, true>(v, lds, twf, tau, qk);
    }
    const ModCtx mc = mod[0];
#pragma unroll
    for (int e = 0; e < E; ++e) v[e] = pmul<AR>(va[e], v[e], mc, qk);     // squaring: v still holds a-hat (lazy)
    inv_transform<AR, L, LIO, true>(v, lds, twi, tau, qk);
    const rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void*)(c + p * n), 0, (u32)n * 8u, 0x00020000);
    store_poly<LIO, true>(rc, off_io, 8u, [&](int e) { return (u64)canon_inv<AR>(v[e], qk); });
  }
}

template <int L, int AR>
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
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pow2_pipe<L, AR>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  });
  if (e != hipSuccess) return e;
  int dev = 0;
  if ((e = hipGetDevice(&dev)) != hipSuccess) return e;
  // resident workgroups: LDS-limited (160 KiB per CU), at most 4 waves per SIMD
  int per_cu = (int)((160 * 1024) / lds_bytes);
  if (per_cu > 16 * 64 / NT) per_cu = 16 * 64 / NT;
  i64 grid = (i64)cus[dev] * per_cu;
  if (grid > a.B) grid = a.B;
  if (grid == 0) return hipSuccess;
  hipLaunchKernelGGL((k_pow2_pipe<L, AR>), dim3((unsigned)grid), dim3(NT), lds_bytes, a.stream, a.y, a.a, a.b, a.B,
                     static_cast<const u32*>(a.tw_fwd), static_cast<const u32*>(a.tw_inv), static_cast<const u32*>(a.scale), a.mod);
  return hipGetLastError();
}

// does the pipelined kernel take this fused poly-mul launch?  (one modulus below 2^31, n = 4096 or 8192, the
// output slab 16-byte aligned, and enough polynomials for every resident workgroup to have several)
bool pow2_pipe_ok(const Pow2Launch& a, bool forced) {
  if (a.T != 1 || a.arith < 2 || (a.L != 12 && a.L != 13)) return false;
  if (((uintptr_t)a.y & 15) || (((uintptr_t)a.a | (uintptr_t)a.b) & 7)) return false;
  return forced || a.B >= 2048;
}

hipError_t launch_pow2_pipe(const Pow2Launch& a) {
  switch (a.arith * 100 + a.L) {
    case 212: return launch_pipe_L<12, 2>(a);
    case 213: return launch_pipe_L<13, 2>(a);
    case 312: return launch_pipe_L<12, 3>(a);
    case 313: return launch_pipe_L<13, 3>(a);
    case 412: return launch_pipe_L<12, 4>(a);
    case 413: return launch_pipe_L<13, 4>(a);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace lolhip
