// tools/microbench_ops.hip — per-instruction issue cost table for gfx950 integer code.
// For each candidate instruction: cycles per wave-instruction per SIMD at 1, 4 and 8 waves per
// SIMD (s_memtime around the loop, median over waves), which is what the Z_q butterflies of
// zq_dev.h are priced with (profiles/r02_microbench_ops.txt).  Development tool, not product.
// Build: hipcc --offload-arch=gfx950 -O3 -o microbench_ops microbench_ops.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <algorithm>
#include <vector>
typedef uint64_t u64; typedef uint32_t u32;

constexpr int ITER = 8192, CH = 8;

#define OPS(X) \
  X(0,  "v_add_u32",        "v_add_u32 %0, %0, %2") \
  X(1,  "v_sub_u32",        "v_sub_u32 %0, %0, %2") \
  X(2,  "v_and_b32",        "v_and_b32 %0, %0, %2") \
  X(3,  "v_xor_b32",        "v_xor_b32 %0, %0, %2") \
  X(4,  "v_lshlrev_b32",    "v_lshlrev_b32 %0, 1, %0") \
  X(5,  "v_ashrrev_i32",    "v_ashrrev_i32 %0, 31, %0") \
  X(6,  "v_mov_b32",        "v_mov_b32 %0, %2") \
  X(7,  "v_cndmask_b32 vcc","v_cndmask_b32 %0, %0, %2, vcc") \
  X(8,  "v_cmp_gt_u32 vcc", "v_cmp_gt_u32 vcc, %0, %2") \
  X(9,  "v_cmp_lt_i32 sgpr","v_cmp_lt_i32_e64 s[20:21], %0, %2") \
  X(10, "v_min_u32",        "v_min_u32 %0, %0, %2") \
  X(11, "v_add3_u32",       "v_add3_u32 %0, %0, %2, %2") \
  X(12, "v_lshl_add_u32",   "v_lshl_add_u32 %0, %0, 1, %2") \
  X(13, "v_and_or_b32",     "v_and_or_b32 %0, %0, %2, %2") \
  X(14, "v_bfi_b32",        "v_bfi_b32 %0, %2, %0, %2") \
  X(15, "v_alignbit_b32",   "v_alignbit_b32 %0, %0, %2, 7") \
  X(16, "v_mul_u32_u24",    "v_mul_u32_u24 %0, %0, %2") \
  X(17, "v_mad_u32_u24",    "v_mad_u32_u24 %0, %0, %2, %0") \
  X(18, "v_mul_lo_u32",     "v_mul_lo_u32 %0, %0, %2") \
  X(19, "v_mul_hi_u32",     "v_mul_hi_u32 %0, %0, %2") \
  X(20, "v_mad_u64_u32",    "v_mad_u64_u32 %1, s[20:21], %0, %2, %1") \
  X(21, "v_lshl_add_u64",   "v_lshl_add_u64 %1, %1, 0, %3") \
  X(22, "v_add_co_u32",     "v_add_co_u32 %0, vcc, %0, %2") \
  X(23, "v_addc_co_u32",    "v_addc_co_u32 %0, vcc, %0, %2, vcc") \
  X(24, "v_lshlrev_b64",    "v_lshlrev_b64 %1, 1, %1") \
  X(25, "v_ashrrev_i64",    "v_ashrrev_i64 %1, 63, %1") \
  X(26, "v_cmp_lt_u64 vcc", "v_cmp_lt_u64 vcc, %1, %3") \
  X(27, "v_cmp_gt_i64 vcc", "v_cmp_gt_i64 vcc, 0, %1") \
  X(28, "v_not_b32",        "v_not_b32 %0, %0") \
  X(29, "v_mov_b64",        "v_mov_b64 %1, %3") \
  X(30, "v_add_f64",        "v_add_f64 %1, %1, %3") \
  X(31, "v_fma_f64",        "v_fma_f64 %1, %1, %3, %1") \
  X(32, "v_mul_f64",        "v_mul_f64 %1, %1, %3") \
  X(33, "v_cndmask_b32 sgpr","v_cndmask_b32_e64 %0, %0, %2, s[22:23]") \
  X(34, "v_xad_u32",        "v_xad_u32 %0, %0, %2, %2") \
  X(35, "v_sub_co+v_subb (2)","v_sub_co_u32 %0, vcc, %0, %2\n v_subb_co_u32 %2, vcc, %2, %0, vcc") \
  X(36, "s_nop 0",          "s_nop 0") \
  X(37, "v_max_i32",        "v_max_i32 %0, %0, %2") \
  X(38, "v_med3_i32",       "v_med3_i32 %0, %0, %2, %2") \
  X(39, "v_mul_i32_i24",    "v_mul_i32_i24 %0, %0, %2") \
  X(40, "v_mad_i64_i32",    "v_mad_i64_i32 %1, s[20:21], %0, %2, %1") \
  X(41, "v_perm_b32",       "v_perm_b32 %0, %0, %2, %2") \
  X(42, "v_lshrrev_b32",    "v_lshrrev_b32 %0, 3, %0") \
  X(43, "v_or_b32",         "v_or_b32 %0, %0, %2") \
  X(44, "v_cvt_f64_u32",    "v_cvt_f64_u32 %1, %0") \
  X(45, "v_add_u32 sgpr",   "v_add_u32 %0, s24, %0") \
  X(46, "v_sub+v_min (2)",  "v_sub_u32 %2, %0, %2\n v_min_u32 %0, %0, %2") \
  X(47, "v_cmp_i32+2cndmask(3)", "v_cmp_gt_i32 vcc, 0, %0\n v_cndmask_b32 %0, %0, %2, vcc\n v_cndmask_b32 %2, %2, %0, vcc") \
  X(48, "v_pk_add_u16",     "v_pk_add_u16 %0, %0, %2") \
  X(49, "v_mad_u64_u32 x SGPR", "v_mad_u64_u32 %1, s[20:21], %0, s24, %1") \
  X(50, "v_bfe_u32",        "v_bfe_u32 %0, %0, 3, 7") \
  X(51, "v_lshl_or_b32",    "v_lshl_or_b32 %0, %0, 3, %2") \
  X(52, "v_add_lshl_u32",   "v_add_lshl_u32 %0, %0, %2, 1") \
  X(53, "v_sub_co_u32 sgprdst","v_sub_co_u32_e64 %0, s[20:21], %0, %2") \
  X(54, "v_cmp_class? skip -> v_cmp_eq_u32", "v_cmp_eq_u32 vcc, %0, %2") \
  X(55, "v_mul_lo_u32 x SGPR", "v_mul_lo_u32 %0, %0, s24")

template <int OP>
__global__ void __launch_bounds__(256) k_op(u32* out, unsigned long long* cyc, u32 a, u32 b_in) {
  u32 x[CH], bb[CH]; u64 y[CH], z[CH];
  for (int i = 0; i < CH; i++) {
    x[i] = threadIdx.x * (i + 3) + a; bb[i] = b_in + i * 7 + threadIdx.x;
    y[i] = (u64)x[i] * 0x9E3779B97F4A7C15ull; z[i] = y[i] ^ 0x5555555555555555ull;
    if (OP >= 30 && OP <= 32) { y[i] = __double_as_longlong(1.0 + 1e-9 * x[i]); z[i] = __double_as_longlong(1.0 - 1e-9 * bb[i]); }
  }
  asm volatile("s_mov_b64 vcc, 0x5555\n s_mov_b64 s[22:23], 0x3333\n s_mov_b32 s24, 0x12345" ::: "vcc", "s22", "s23", "s24");
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int i = 0; i < CH; i++) {
#define X(ID, NAME, STR) if (OP == ID) asm volatile(STR : "+v"(x[i]), "+v"(y[i]), "+v"(bb[i]), "+v"(z[i]) :: "vcc", "s20", "s21");
      OPS(X)
#undef X
    }
  }
  asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
  u32 acc = 0; for (int i = 0; i < CH; i++) acc += x[i] + bb[i] + (u32)y[i] + (u32)(y[i] >> 32) + (u32)z[i];
  out[blockIdx.x * 256 + threadIdx.x] = acc;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

struct Res { double cyc_per_instr; double ns; };
template <int OP> Res run(int cus, int wps, u32* out, unsigned long long* cyc) {
  // wps waves per SIMD: blocks of 256 threads (4 waves = one per SIMD), wps blocks per CU
  const int blocks = cus * wps;
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  hipLaunchKernelGGL((k_op<OP>), dim3(blocks), dim3(256), 0, 0, out, cyc, 12345u, 0x9E3779B1u);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(a);
  hipLaunchKernelGGL((k_op<OP>), dim3(blocks), dim3(256), 0, 0, out, cyc, 12345u, 0x9E3779B1u);
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  std::vector<unsigned long long> h((size_t)blocks * 4);
  (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  const double med = (double)h[h.size() / 2];
  return {med / ((double)ITER * CH * wps), ms * 1e6 / ((double)ITER * CH * wps)};
}

int main() {
  hipDeviceProp_t pr; (void)hipGetDeviceProperties(&pr, 0);
  const int cus = pr.multiProcessorCount;
  printf("device %s CUs=%d; s_memtime ticks and wall-clock ns per wave-instruction per SIMD at 1 / 4 / 8 waves per SIMD\n", pr.name, cus);
  u32* out; unsigned long long* cyc;
  (void)hipMalloc(&out, (size_t)cus * 8 * 256 * 4); (void)hipMalloc(&cyc, (size_t)cus * 8 * 4 * 8);
#define X(ID, NAME, STR) { Res r1 = run<ID>(cus, 1, out, cyc), r4 = run<ID>(cus, 4, out, cyc), r8 = run<ID>(cus, 8, out, cyc); \
    printf("%-28s ticks %6.2f %6.2f %6.2f   ns %6.3f %6.3f %6.3f\n", NAME, r1.cyc_per_instr, r4.cyc_per_instr, r8.cyc_per_instr, r1.ns, r4.ns, r8.ns); }
  OPS(X)
#undef X
  return 0;
}
