// lol_amd/csrc/floatpath.hip — the floating-point members of the Tensor class that sit beside the
// Z_q hot path, float64, separate tolerance contract (rel 1e-12 against lol-cpp):
//
//   tensorCRTC / tensorCRTInvC   crt.cpp:583-598: ppcrt / ppcrtinv instantiated at Complex — the CRT
//                                over C that UCyc falls back to when a modulus has no CRT basis
//                                (CRTExt, UCyc.hs:422-444).  The plan's Z_q stage lists are
//                                reused with a second constant pool over C (plan.cpp).
//   tensorGaussianDec            random.cpp:19-64: the linear map that turns iid real Gaussians
//                                into a sample in the decoding basis (CPP.hs:376-389): per odd
//                                prime p a real (p-1) x (p-1) matrix on every (p-1)-vector.
//
// One polynomial per workgroup, resident in LDS (n <= 8192: 128 KiB of complex doubles); one
// thread owns one d-vector of a stage (d <= 13), reads it once, applies the dense map in
// registers in the reference's summation order, writes it back in place.
#include <hip/hip_runtime.h>

#include <atomic>

#include "kernels.h"

namespace lolhip {

namespace {

__device__ __forceinline__ int fdiv40(int x, u64 M) { return (int)(((u64)(u32)x * M) >> 40); }   // x / v, M = floor(2^40/v)+1

struct cd { double re, im; };
__device__ __forceinline__ cd cmul(cd a, cd b) { return cd{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__device__ __forceinline__ cd cadd(cd a, cd b) { return cd{a.re + b.re, a.im + b.im}; }

template <int D>
__device__ __forceinline__ void cplx_stage_vec(const Stage& st, cd* __restrict__ buf, int vec, const cd* __restrict__ cst) {
  const int rts = st.rts;
  const int blk = fdiv40(vec, st.m_rts), r = vec - blk * rts;
  const int x0 = blk * D * rts + r;
  cd* base = buf + x0;
  cd v[D], o[D];
#pragma unroll
  for (int i = 0; i < D; ++i) v[i] = base[i * rts];
  const cd* M = cst + st.mat_off;
#pragma unroll
  for (int i = 0; i < D; ++i) {
    cd acc = cmul(v[0], M[i * D]);
#pragma unroll
    for (int c = 1; c < D; ++c) acc = cadd(acc, cmul(v[c], M[i * D + c]));
    o[i] = acc;
  }
  if (st.tw_off >= 0) {
#pragma unroll
    for (int i = 0; i < D; ++i) {
      const int xd = fdiv40(x0 + i * rts, st.m_twdiv);
      o[i] = cmul(o[i], cst[st.tw_off + xd - fdiv40(xd, st.m_twmod) * st.tw_mod]);
    }
  }
#pragma unroll
  for (int i = 0; i < D; ++i) base[i * rts] = o[i];
}

__global__ void __launch_bounds__(256)
k_cplx(cd* __restrict__ y, i64 B, int n, const Stage* __restrict__ stages, int nstages, const cd* __restrict__ cst) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  cd* buf = reinterpret_cast<cd*>(smem);
  for (i64 b = blockIdx.x; b < B; b += gridDim.x) {
    cd* yb = y + (size_t)b * n;
    for (int x = threadIdx.x; x < n; x += blockDim.x) buf[x] = yb[x];
    __syncthreads();
    for (int s = 0; s < nstages; ++s) {
      const Stage st = stages[s];
      if (st.kind == ST_DIAG || st.kind == ST_SCALE) {
        for (int x = threadIdx.x; x < n; x += blockDim.x) {
          const int xd = fdiv40(x, st.m_twdiv);
          buf[x] = cmul(buf[x], cst[st.tw_off + xd - fdiv40(xd, st.m_twmod) * st.tw_mod]);
        }
      } else {
        const int nvec = n / st.d;
        for (int vec = threadIdx.x; vec < nvec; vec += blockDim.x) {
          switch (st.d) {
            case 2: cplx_stage_vec<2>(st, buf, vec, cst); break;
            case 3: cplx_stage_vec<3>(st, buf, vec, cst); break;
            case 4: cplx_stage_vec<4>(st, buf, vec, cst); break;
            case 5: cplx_stage_vec<5>(st, buf, vec, cst); break;
            case 6: cplx_stage_vec<6>(st, buf, vec, cst); break;
            case 7: cplx_stage_vec<7>(st, buf, vec, cst); break;
            case 10: cplx_stage_vec<10>(st, buf, vec, cst); break;
            case 11: cplx_stage_vec<11>(st, buf, vec, cst); break;
            case 12: cplx_stage_vec<12>(st, buf, vec, cst); break;
            case 13: cplx_stage_vec<13>(st, buf, vec, cst); break;
            default: break;   // excluded on the host (Plan::float_ok)
          }
        }
      }
      __syncthreads();
    }
    for (int x = threadIdx.x; x < n; x += blockDim.x) yb[x] = buf[x];
    __syncthreads();
  }
}

template <int D>
__device__ __forceinline__ void gauss_stage_vec(const Stage& st, double* __restrict__ buf, int vec, const double* __restrict__ cst) {
  const int rts = st.rts;
  const int blk = fdiv40(vec, st.m_rts), r = vec - blk * rts;
  double* base = buf + blk * D * rts + r;
  double v[D], o[D];
#pragma unroll
  for (int i = 0; i < D; ++i) v[i] = base[i * rts];
  const double* M = cst + st.mat_off;         // entries 2 c(row * col mod p)
#pragma unroll
  for (int i = 0; i < D; ++i) {
    double acc = 0.0;
#pragma unroll
    for (int c = 0; c < D; ++c) acc += M[i * D + c] * v[c];      // the reference's order of summation (random.cpp:35-41)
    o[i] = acc / 1.4142135623730951;                             // acc / sqrt(2) (random.cpp:42)
  }
#pragma unroll
  for (int i = 0; i < D; ++i) base[i * rts] = o[i];
}

__global__ void __launch_bounds__(256)
k_gauss(double* __restrict__ y, i64 B, int n, const Stage* __restrict__ stages, int nstages, const double* __restrict__ cst) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double* buf = reinterpret_cast<double*>(smem);
  for (i64 b = blockIdx.x; b < B; b += gridDim.x) {
    double* yb = y + (size_t)b * n;
    for (int x = threadIdx.x; x < n; x += blockDim.x) buf[x] = yb[x];
    __syncthreads();
    for (int s = 0; s < nstages; ++s) {
      const Stage st = stages[s];
      const int nvec = n / st.d;
      for (int vec = threadIdx.x; vec < nvec; vec += blockDim.x) {
        switch (st.d) {
          case 2: gauss_stage_vec<2>(st, buf, vec, cst); break;
          case 4: gauss_stage_vec<4>(st, buf, vec, cst); break;
          case 6: gauss_stage_vec<6>(st, buf, vec, cst); break;
          case 10: gauss_stage_vec<10>(st, buf, vec, cst); break;
          case 12: gauss_stage_vec<12>(st, buf, vec, cst); break;
          default: break;
        }
      }
      __syncthreads();
    }
    for (int x = threadIdx.x; x < n; x += blockDim.x) yb[x] = buf[x];
    __syncthreads();
  }
}

}  // namespace

hipError_t launch_cplx(hipStream_t s, double* y, i64 B, i64 n, const Stage* stages, int nstages, const double* cconsts) {
  if (B == 0 || nstages == 0) return hipSuccess;
  const size_t lds = (size_t)n * sizeof(cd);
  if (lds > 64 * 1024) {
    // per device, as for every kernel that needs more than 64 KiB of LDS
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    static std::atomic<unsigned long long> done{0};
    if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!(done.load(std::memory_order_acquire) >> dev & 1)) {
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cplx), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
      if (e != hipSuccess) return e;
      done.fetch_or(1ull << dev, std::memory_order_release);
    }
  }
  i64 grid = B < 4096 ? B : 4096;
  hipLaunchKernelGGL(k_cplx, dim3((unsigned)grid), dim3(256), lds, s, reinterpret_cast<cd*>(y), B, (int)n, stages, nstages,
                     reinterpret_cast<const cd*>(cconsts));
  return hipGetLastError();
}

hipError_t launch_gauss(hipStream_t s, double* y, i64 B, i64 n, const Stage* stages, int nstages, const double* rconsts) {
  if (B == 0 || nstages == 0) return hipSuccess;
  const size_t lds = (size_t)n * sizeof(double);          // <= 64 KiB
  i64 grid = B < 4096 ? B : 4096;
  hipLaunchKernelGGL(k_gauss, dim3((unsigned)grid), dim3(256), lds, s, y, B, (int)n, stages, nstages, rconsts);
  return hipGetLastError();
}

}  // namespace lolhip
