// tools/microbench_dma.hip — what does each way of bringing an int64 operand slab on chip sustain, with the
// launch shape of k_pow2_pipe (512-thread workgroups, 2 per CU, every wave fetching its own 8 KiB per operand)?
//   0: buffer_load_dword ... lds, lanes 8 bytes apart (low dwords only; 16 pieces per wave and operand)   [the pipe kernel's form]
//   1: buffer_load_dwordx4 ... lds (whole rows; 8 pieces per wave and operand)
//   2: buffer_load_dwordx4 to registers (8 per lane and operand)
//   3: as 0, two operands in flight per wave (two row sets)
// Every variant waits for an operand to land before it asks for the next (depth 1, as the pipe kernel does per transform).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned u32;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
__device__ __forceinline__ void dma4(rsrc_t r, u32 lds, u32 voff, u32 soff) {
  u32 keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dword %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "s"(lds), "v"(voff), "s"(r), "s"(soff));
}
__device__ __forceinline__ void dma16(rsrc_t r, u32 lds, u32 voff, u32 soff) {
  u32 keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "s"(lds), "v"(voff), "s"(r), "s"(soff));
}
template <int V> __global__ void __launch_bounds__(512, 4) k(const long* src, long nops, u32* out) {
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const u32 base = (u32)(size_t)smem + wv * (V == 1 ? 8192 : 4352) * (V == 3 ? 2 : 1);
  u32 acc = 0;
  for (long op = blockIdx.x; op < nops; op += gridDim.x) {
    const rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)(src + op * 8192), 0, 65536u, 0x00020000);
    if constexpr (V == 0 || V == 3) {
#pragma unroll
      for (int c = 0; c < 16; ++c) dma4(r, base + c * 272 + (V == 3 ? (op / gridDim.x & 1) * 4352 : 0), lane * 8, wv * 8192 + c * 512);
      if (V == 0) asm volatile("s_waitcnt vmcnt(0)"); else asm volatile("s_waitcnt vmcnt(16)");
    } else if constexpr (V == 1) {
#pragma unroll
      for (int c = 0; c < 8; ++c) dma16(r, base + c * 1024, lane * 16, wv * 8192 + c * 1024);
      asm volatile("s_waitcnt vmcnt(0)");
    } else {
      u32x4 x[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) x[c] = __builtin_amdgcn_raw_buffer_load_b128(r, lane * 16, wv * 8192 + c * 1024, 0);
#pragma unroll
      for (int c = 0; c < 8; ++c) acc += x[c].x ^ x[c].z;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)");
  if (acc == 0x12345u) out[threadIdx.x] = acc;
}
int main() {
  const long nops = 8192;                    // 8192 operands of 64 KiB = 512 MiB (beyond the 256 MiB Infinity Cache)
  long* src; u32* out; hipMalloc(&src, nops * 65536); hipMalloc(&out, 4096); hipMemset(src, 1, nops * 65536);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](int v, int grid) {
    size_t lds = v == 1 ? 65536 : (v == 3 ? 69632 : 34816);
    auto launch = [&]() {
      switch (v) { case 0: k<0><<<grid, 512, lds>>>(src, nops, out); break; case 1: k<1><<<grid, 512, lds>>>(src, nops, out); break;
                   case 2: k<2><<<grid, 512, lds>>>(src, nops, out); break; default: k<3><<<grid, 512, lds>>>(src, nops, out); }
    };
    hipFuncSetAttribute((const void*)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void*)k<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 69632);
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0); for (int i = 0; i < 5; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("variant %d grid %4d: %.3f ms  %.2f TB/s of lines (%.1f us per 64 KiB operand per workgroup)\n", v, grid, ms, nops * 65536.0 / ms / 1e9, ms * 1e3 * grid / nops);
  };
  for (int v = 0; v < 4; ++v) for (int grid : {256, 512, 1024}) run(v, grid);
  return 0;
}
