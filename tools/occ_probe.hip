// tools/occ_probe.hip — how many workgroups of a given LDS size / thread count does a CU really hold?
// Each workgroup spins ~T us of s_memrealtime; kernel time / T = dispatch rounds of a 2 x CUs grid.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
extern __shared__ unsigned char smem[];
template <int VG> __global__ void __launch_bounds__(512) spin(unsigned long long ticks, unsigned* out) {
  volatile unsigned char* s = smem;
  s[threadIdx.x] = 1;
  unsigned long long t0, t;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
  do { asm volatile("s_sleep 32\n\ts_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)); } while (t - t0 < ticks);
  if (threadIdx.x == 0) out[blockIdx.x] = (unsigned)(t - t0);
}
int main(int argc, char** argv) {
  int cus = 256; unsigned* out; hipMalloc(&out, 4096 * 4);
  for (int threads : {256, 512}) for (int kb : {32, 48, 64, 72, 76, 78, 80, 96, 128, 160}) {
    size_t lds = (size_t)kb * 1024;
    hipFuncSetAttribute((const void*)spin<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    int occ = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)spin<0>, threads, lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const unsigned long long ticks = 20000;  // 200 us of the 100 MHz counter
    int grid = cus * 4;
    spin<0><<<grid, threads, lds>>>(ticks, out); hipDeviceSynchronize();
    hipEventRecord(e0); spin<0><<<grid, threads, lds>>>(ticks, out); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("threads %d lds %3d KiB: occupancy query %d; %d workgroups took %.3f ms = %.2f rounds of 0.2 ms -> %.2f resident per CU\n",
           threads, kb, occ, grid, ms, ms / 0.2, 4.0 / (ms / 0.2));
  }
  return 0;
}
