// tools/microbench_bfly.hip — throughput + exactness of candidate 64-bit NTT butterflies on gfx950.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#include <random>
typedef uint64_t u64; typedef uint32_t u32; typedef int64_t i64; typedef unsigned __int128 u128;
constexpr int ITER = 1024, CH = 8;

__device__ __forceinline__ u64 mad64(u32 a, u32 b, u64 c){ u64 d, s; asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(s) : "v"(a), "v"(b), "v"(c)); return d; }
__device__ __forceinline__ u64 add64(u64 a, u64 b){ u64 d; asm("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ u64 shl1add64(u64 a, u64 b){ u64 d; asm("v_lshl_add_u64 %0, %1, 1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ u32 lo32(u64 x){return (u32)x;} 
__device__ __forceinline__ u32 hi32(u64 x){return (u32)(x>>32);} 

struct K { u64 q, nq, q2, nq2, q4, nq4; };

// ---- F0: exact quotient, plain C (what the first kernel used) ----
__device__ __forceinline__ void f0(u64& X, u64& Y, u64 w, u64 wp, const K& k){
  u64 t0 = X - k.q2; u64 x = (i64)t0 < 0 ? X : t0;
  u64 Q = __umul64hi(wp, Y); u64 t = w*Y - Q*k.q;
  X = x + t; Y = x - t + k.q2;
}
// ---- F1: approximate quotient (err<=2), [0,8q) invariant, mad chains ----
__device__ __forceinline__ u64 shoup_approx(u64 y, u64 w, u64 wp, u64 nq, u64 init){
  u32 ah = __umulhi(hi32(wp), lo32(y));
  u32 bh = __umulhi(lo32(wp), hi32(y));
  u64 Q = add64(mad64(hi32(wp), hi32(y), (u64)ah), (u64)bh);
  u64 t = mad64(lo32(w), lo32(y), init);
  t = mad64(lo32(Q), lo32(nq), t);
  u64 h = mad64(lo32(w), hi32(y), 0);
  h = mad64(hi32(w), lo32(y), h);
  h = mad64(lo32(Q), hi32(nq), h);
  h = mad64(hi32(Q), lo32(nq), h);
  return ((u64)(hi32(t) + lo32(h)) << 32) | lo32(t);
}
__device__ __forceinline__ u64 csubn(u64 x, u64 negm){ u64 t = add64(x, negm); return ((int)hi32(t) < 0) ? x : t; }
__device__ __forceinline__ void f1(u64& X, u64& Y, u64 w, u64 wp, const K& k){
  u64 x = csubn(X, k.nq4);
  u64 Xn = shoup_approx(Y, w, wp, k.nq, x);
  u64 z = shl1add64(x, k.q4);
  X = Xn; Y = z - Xn;
}
// ---- F2: like F1 but T chain with mul_lo/add (compiler's choice) ----
__device__ __forceinline__ void f2(u64& X, u64& Y, u64 w, u64 wp, const K& k){
  u64 x = csubn(X, k.nq4);
  u32 ah = __umulhi(hi32(wp), lo32(Y));
  u32 bh = __umulhi(lo32(wp), hi32(Y));
  u64 Q = (u64)hi32(wp)*hi32(Y) + ah + bh;
  u64 t = w*Y + Q*k.nq;
  u64 Xn = x + t;
  X = Xn; Y = ((x<<1) + k.q4) - Xn;
}
// ---- G1: inverse (GS) butterfly, approx quotient, [0,4q) invariant ----
__device__ __forceinline__ void g1(u64& X, u64& Y, u64 w, u64 wp, const K& k){
  u64 s = add64(X, Y);
  u64 d = add64(X, k.q4) - Y;
  X = csubn(s, k.nq4);
  Y = shoup_approx(d, w, wp, k.nq, 0);
}
__device__ __forceinline__ void g0(u64& X, u64& Y, u64 w, u64 wp, const K& k){
  u64 s = X + Y; u64 d = X - Y + k.q2;
  u64 t0 = s - k.q2; X = (i64)t0 < 0 ? s : t0;
  u64 Q = __umul64hi(wp, d); Y = w*d - Q*k.q;
}

template<int V> __global__ void __launch_bounds__(256) k_thr(u64* out, const u64* tw, const K* kp){
  const K k = *kp;
  u64 y[CH]; for(int i=0;i<CH;i++) y[i] = (((threadIdx.x+i+1)*0x9E3779B97F4A7C15ull)>>4) % k.q;
  const u64* t = tw + 2*(threadIdx.x & 63);
  for(int it=0; it<ITER; ++it){
    u64 w = t[0] , wp = t[1]; t += 0; 
    asm volatile("" : "+v"(w), "+v"(wp));
    #pragma unroll
    for(int i=0;i<CH;i+=2){
      if(V==0) f0(y[i],y[i+1],w,wp,k);
      if(V==1) f1(y[i],y[i+1],w,wp,k);
      if(V==2) f2(y[i],y[i+1],w,wp,k);
      if(V==3) g0(y[i],y[i+1],w,wp,k);
      if(V==4) g1(y[i],y[i+1],w,wp,k);
    }
  }
  u64 acc=0; for(int i=0;i<CH;i++) acc+=y[i]; out[blockIdx.x*256+threadIdx.x]=acc;
}
// exactness: one butterfly per thread on random in-range inputs; host checks congruence + range
template<int V> __global__ void k_chk(u64* xy, const u64* tw, const K* kp, int n){
  int i = blockIdx.x*blockDim.x+threadIdx.x; if(i>=n) return;
  const K k=*kp; u64 X=xy[2*i], Y=xy[2*i+1];
  if(V==0) f0(X,Y,tw[2*i],tw[2*i+1],k);
  if(V==1) f1(X,Y,tw[2*i],tw[2*i+1],k);
  if(V==2) f2(X,Y,tw[2*i],tw[2*i+1],k);
  if(V==3) g0(X,Y,tw[2*i],tw[2*i+1],k);
  if(V==4) g1(X,Y,tw[2*i],tw[2*i+1],k);
  xy[2*i]=X; xy[2*i+1]=Y;
}
template<typename F> double timeit(F f){ hipEvent_t a,b; (void)hipEventCreate(&a); (void)hipEventCreate(&b); f(); (void)hipDeviceSynchronize();
  (void)hipEventRecord(a); for(int i=0;i<10;i++) f(); (void)hipEventRecord(b); (void)hipEventSynchronize(b); float ms; (void)hipEventElapsedTime(&ms,a,b); return ms/10.0; }

int main(){
  hipDeviceProp_t pr; (void)hipGetDeviceProperties(&pr,0); int cus=pr.multiProcessorCount; double ghz=pr.clockRate/1e6;
  u64 q = 1152921504606994433ull;
  K hk{q, 0-q, 2*q, 0-2*q, 4*q, 0-4*q}; K* dk; (void)hipMalloc(&dk,sizeof(K)); (void)hipMemcpy(dk,&hk,sizeof(K),hipMemcpyHostToDevice);
  const int N = 1<<20; std::mt19937_64 rng(1);
  std::vector<u64> tw(2*N), xy(2*N), in(2*N);
  for(int i=0;i<N;i++){ u64 w = rng()%q; tw[2*i]=w; tw[2*i+1]=(u64)(((u128)w<<64)/q); }
  u64 *dtw,*dxy,*dout; (void)hipMalloc(&dtw,16ull*N); (void)hipMalloc(&dxy,16ull*N); (void)hipMemcpy(dtw,tw.data(),16ull*N,hipMemcpyHostToDevice);
  int blocks=cus*8; (void)hipMalloc(&dout,(size_t)blocks*256*8);
  const char* names[]={"F0 fwd exact C","F1 fwd approx mad-chain","F2 fwd approx C","G0 inv exact C","G1 inv approx mad-chain"};
  const u64 inB[]={4,8,8,2,4};   // input range bound (multiples of q)
  const u64 outB[]={4,8,8,2,4};
  for(int v=0; v<5; ++v){
    for(int i=0;i<2*N;i++){ u64 r = rng(); in[i] = (i%97==0) ? (inB[v]*q-1-(r%3)) : (i%89==0 ? r%3 : (u64)(((u128)r*(inB[v]*q))>>64)); }
    (void)hipMemcpy(dxy,in.data(),16ull*N,hipMemcpyHostToDevice);
    switch(v){ case 0: hipLaunchKernelGGL(k_chk<0>,dim3(N/256),dim3(256),0,0,dxy,dtw,dk,N); break; case 1: hipLaunchKernelGGL(k_chk<1>,dim3(N/256),dim3(256),0,0,dxy,dtw,dk,N); break;
      case 2: hipLaunchKernelGGL(k_chk<2>,dim3(N/256),dim3(256),0,0,dxy,dtw,dk,N); break; case 3: hipLaunchKernelGGL(k_chk<3>,dim3(N/256),dim3(256),0,0,dxy,dtw,dk,N); break;
      case 4: hipLaunchKernelGGL(k_chk<4>,dim3(N/256),dim3(256),0,0,dxy,dtw,dk,N); break; }
    (void)hipMemcpy(xy.data(),dxy,16ull*N,hipMemcpyDeviceToHost);
    long bad=0; u64 maxo=0;
    for(int i=0;i<N;i++){ u64 X=in[2*i]%q, Y=in[2*i+1]%q, w=tw[2*i]; u64 ex, ey;
      if(v<3){ u64 t=(u64)((u128)Y*w%q); ex=(X+t)%q; ey=(X+q-t)%q; } else { ex=(X+Y)%q; ey=(u64)((u128)((X+q-Y)%q)*w%q); }
      if(xy[2*i]%q!=ex || xy[2*i+1]%q!=ey || xy[2*i]>=outB[v]*q || xy[2*i+1]>=outB[v]*q) bad++;
      if(xy[2*i]>maxo) maxo=xy[2*i]; if(xy[2*i+1]>maxo) maxo=xy[2*i+1]; }
    double ops=(double)blocks*256*ITER*CH/2; double ms=0;
    switch(v){ case 0: ms=timeit([&]{hipLaunchKernelGGL(k_thr<0>,dim3(blocks),dim3(256),0,0,dout,dtw,dk);}); break; case 1: ms=timeit([&]{hipLaunchKernelGGL(k_thr<1>,dim3(blocks),dim3(256),0,0,dout,dtw,dk);}); break;
      case 2: ms=timeit([&]{hipLaunchKernelGGL(k_thr<2>,dim3(blocks),dim3(256),0,0,dout,dtw,dk);}); break; case 3: ms=timeit([&]{hipLaunchKernelGGL(k_thr<3>,dim3(blocks),dim3(256),0,0,dout,dtw,dk);}); break;
      case 4: ms=timeit([&]{hipLaunchKernelGGL(k_thr<4>,dim3(blocks),dim3(256),0,0,dout,dtw,dk);}); break; }
    double bpc = ops/(ms*1e-3)/cus/(ghz*1e9);
    printf("%-28s bad=%ld max/q=%.3f  %7.3f ms  %6.3f bfly/clk/CU  (%.1f slots/bfly @128 lanes/clk)\n", names[v], bad, (double)maxo/q, ms, bpc, 128.0/bpc);
  }
  return 0;
}
