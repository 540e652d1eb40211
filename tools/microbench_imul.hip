// tools/microbench_imul.hip — integer-ALU issue-rate probe for gfx950.
// Measures what bounds the 61-bit NTT: per-CU throughput of the 32x32 multiply
// family vs plain adds, and of whole Shoup mulmods / Harvey butterflies.
// Build: hipcc --offload-arch=gfx950 -O3 -o microbench_imul microbench_imul.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
typedef uint64_t u64; typedef uint32_t u32;

#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); return 1;}}while(0)

constexpr int ITER = 2048, CH = 8;

template<int OP> __global__ void __launch_bounds__(256) k_op(u32* out, u32 a, u32 b){
  u32 x[CH]; u64 y[CH];
  for(int i=0;i<CH;i++){ x[i]=threadIdx.x*(i+3)+a; y[i]=(u64)x[i]*0x9E3779B97F4A7C15ull; }
  for(int it=0; it<ITER; ++it){
    #pragma unroll
    for(int i=0;i<CH;i++){
      if(OP==0) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x[i]) : "v"(b));
      if(OP==1) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x[i]) : "v"(b));
      if(OP==2) asm volatile("v_mad_u64_u32 %0, s[10:11], %1, %2, %0" : "+v"(y[i]) : "v"(x[i]), "v"(b) : "s10","s11");
      if(OP==3) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[i]) : "v"(b));
      if(OP==4) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(y[i]) : "v"(y[(i+1)%CH]));
      if(OP==5) asm volatile("v_add_co_u32 %0, vcc, %0, %2\n v_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(x[i]), "+v"(x[(i+4)%CH]) : "v"(a), "v"(b) : "vcc");
      if(OP==6) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(x[i]) : "v"(b));
      if(OP==7) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(b));
      if(OP==8) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(b) : "vcc");
      if(OP==9) asm volatile("v_min_u32 %0, %0, %1" : "+v"(x[i]) : "v"(b));
      if(OP==10){ double d; asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(y[i])); }
    }
  }
  u32 acc=0; for(int i=0;i<CH;i++) acc += x[i] + (u32)y[i] + (u32)(y[i]>>32);
  out[blockIdx.x*256+threadIdx.x]=acc;
}

__device__ __forceinline__ u64 shoup(u64 y, u64 w, u64 wp, u64 q){ u64 Q=__umul64hi(wp,y); return w*y-Q*q; }

// 64-bit Shoup mulmod chain
__global__ void __launch_bounds__(256) k_shoup64(u64* out, u64 w, u64 wp, u64 q){
  u64 y[CH]; for(int i=0;i<CH;i++) y[i]=(threadIdx.x+i)*0x9E3779B97F4A7C15ull;
  for(int it=0; it<ITER; ++it){
    #pragma unroll
    for(int i=0;i<CH;i++) y[i]=shoup(y[i],w+i,wp+it,q);
  }
  u64 acc=0; for(int i=0;i<CH;i++) acc+=y[i]; out[blockIdx.x*256+threadIdx.x]=acc;
}
// 64-bit Harvey butterfly chain (X,Y) pairs
__global__ void __launch_bounds__(256) k_bfly64(u64* out, u64 w, u64 wp, u64 q){
  u64 y[CH]; for(int i=0;i<CH;i++) y[i]=((threadIdx.x+i)*0x9E3779B97F4A7C15ull)>>4;
  const u64 q2=2*q;
  for(int it=0; it<ITER; ++it){
    #pragma unroll
    for(int i=0;i<CH;i+=2){
      u64 X=y[i], Y=y[i+1];
      u64 t=X-q2; X=(int64_t)t<0?X:t;
      u64 T=shoup(Y,w+i,wp+it,q);
      y[i]=X+T; y[i+1]=X-T+q2;
    }
  }
  u64 acc=0; for(int i=0;i<CH;i++) acc+=y[i]; out[blockIdx.x*256+threadIdx.x]=acc;
}
__device__ __forceinline__ u32 shoup32(u32 y,u32 w,u32 wp,u32 q){ u32 Q=__umulhi(wp,y); return w*y-Q*q; }
__global__ void __launch_bounds__(256) k_bfly32(u32* out, u32 w, u32 wp, u32 q){
  u32 y[CH]; for(int i=0;i<CH;i++) y[i]=((threadIdx.x+i)*0x9E3779B9u)>>3;
  const u32 q2=2*q;
  for(int it=0; it<ITER; ++it){
    #pragma unroll
    for(int i=0;i<CH;i+=2){
      u32 X=y[i], Y=y[i+1];
      X=min(X,X-q2);
      u32 T=shoup32(Y,w+i,wp+it,q);
      y[i]=X+T; y[i+1]=X-T+q2;
    }
  }
  u32 acc=0; for(int i=0;i<CH;i++) acc+=y[i]; out[blockIdx.x*256+threadIdx.x]=acc;
}

template<typename F> double timeit(F f){
  hipEvent_t a,b; hipEventCreate(&a); hipEventCreate(&b);
  f(); hipDeviceSynchronize();
  hipEventRecord(a); for(int i=0;i<5;i++) f(); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms,a,b); return ms/5.0;
}

int main(){
  hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr,0));
  int cus=pr.multiProcessorCount; double ghz=pr.clockRate/1e6;
  printf("device %s CUs=%d clock=%.2f GHz\n", pr.name, cus, ghz);
  int blocks=cus*8; void* out; CK(hipMalloc(&out, (size_t)blocks*256*8));
  const char* names[]={"v_mul_lo_u32","v_mul_hi_u32","v_mad_u64_u32","v_add_u32","v_lshl_add_u64","add_co+addc(2 instr)","v_mad_u32_u24","v_add3_u32","v_cndmask_b32","v_min_u32","v_fma_f64"};
  double ops=(double)blocks*256*ITER*CH;
  #define RUN(OP) { double ms=timeit([&]{ hipLaunchKernelGGL((k_op<OP>), dim3(blocks), dim3(256),0,0,(u32*)out,12345u,0x9E3779B1u);}); \
    printf("%-24s %8.3f ms  %7.2f Gop/s  %6.2f lane-ops/clk/CU (at %.2f GHz)\n", names[OP], ms, ops/ms/1e6, ops/(ms*1e-3)/cus/(ghz*1e9), ghz);}
  RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10)
  u64 q=(1ull<<60)+33*16384+1;
  { double ms=timeit([&]{ hipLaunchKernelGGL(k_shoup64, dim3(blocks), dim3(256),0,0,(u64*)out,q/3,q/5,q);});
    printf("%-24s %8.3f ms  %7.2f Gmulmod/s  %6.3f mulmod/clk/CU\n","shoup64 mulmod",ms,ops/ms/1e6, ops/(ms*1e-3)/cus/(ghz*1e9)); }
  { double ms=timeit([&]{ hipLaunchKernelGGL(k_bfly64, dim3(blocks), dim3(256),0,0,(u64*)out,q/3,q/5,q);});
    printf("%-24s %8.3f ms  %7.2f Gbfly/s  %6.3f bfly/clk/CU\n","harvey bfly64",ms,ops/2/ms/1e6, ops/2/(ms*1e-3)/cus/(ghz*1e9)); }
  { double ms=timeit([&]{ hipLaunchKernelGGL(k_bfly32, dim3(blocks), dim3(256),0,0,(u32*)out,12345u,6789u,1073872897u);});
    printf("%-24s %8.3f ms  %7.2f Gbfly/s  %6.3f bfly/clk/CU\n","harvey bfly32",ms,ops/2/ms/1e6, ops/2/(ms*1e-3)/cus/(ghz*1e9)); }
  return 0;
}
