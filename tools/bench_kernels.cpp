// tools/bench_kernels.cpp — torch-free timing driver over the C ABI (development aid for
// rocprofv3 runs).  usage: bench_kernels <log2 m | mNNN (any index, e.g. m15015)> <T> <B> <op: crt|crtinv|polymul|roundtrip|l|mulgpow|divgdec> <iters> [qbits]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <random>
#include "../include/lolhip.h"
#include <dlfcn.h>
#define CK(x) do{ if((x)!=hipSuccess){ printf("hip error line %d\n", __LINE__); return 1; } }while(0)
int main(int argc, char** argv){
  int lm = (argc>1 && argv[1][0] != 'm') ? atoi(argv[1]) : 14; long marb = (argc>1 && argv[1][0] == 'm') ? atol(argv[1] + 1) : 0;
  int T = argc>2? atoi(argv[2]) : 1; long B = argc>3? atol(argv[3]) : 4096;
  const char* op = argc>4? argv[4] : "polymul"; int iters = argc>5? atoi(argv[5]) : 10; int qbits = argc>6? atoi(argv[6]) : 60;
  std::vector<lolhip_pp> pps; long mval = marb ? marb : ((long)1 << lm);
  if (!marb) pps.push_back(lolhip_pp{2,(int16_t)lm});
  else { long r = marb; for (long p = 2; r > 1; ++p) { int e = 0; while (r % p == 0) { r /= p; ++e; } if (e) pps.push_back(lolhip_pp{(int16_t)p,(int16_t)e}); } }
  std::vector<int64_t> qs; int64_t lower = (int64_t)1<<qbits;
  for(int t=0;t<T;t++){ int64_t q = lolhip_good_q(mval, lower); qs.push_back(q); lower = q; }
  lolhip_plan* P; int rc = lolhip_plan_create(pps.data(),(int)pps.size(),qs.data(),T,0,&P); if(rc){ printf("plan rc=%d\n",rc); return 1; }
  long n = lolhip_plan_n(P); size_t cnt = (size_t)B*n*T;
  std::vector<int64_t> h(cnt); std::mt19937_64 rng(5); for(size_t i=0;i<cnt;i++) h[i] = (int64_t)(rng() % (uint64_t)qs[i%T]);
  int64_t *a,*b,*c; CK(hipMalloc(&a,cnt*8)); CK(hipMalloc(&b,cnt*8)); CK(hipMalloc(&c,cnt*8));
  CK(hipMemcpy(a,h.data(),cnt*8,hipMemcpyHostToDevice)); CK(hipMemcpy(b,h.data(),cnt*8,hipMemcpyHostToDevice));
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run=[&]()->int{ if(!strcmp(op,"crt")) return lolhip_crt_batch(P,0,a,B); if(!strcmp(op,"crtinv")) return lolhip_crtinv_batch(P,0,a,B);
    if(!strcmp(op,"l")) return lolhip_l_batch(P,0,a,B); if(!strcmp(op,"mulgpow")) return lolhip_mulgpow_batch(P,0,a,B); if(!strcmp(op,"divgdec")) return lolhip_divgdec_batch(P,0,a,B);
    if(!strcmp(op,"copy0")) return lolhip_copy_slab(0,c,a,(int64_t)cnt*8,0); if(!strcmp(op,"copy1")) return lolhip_copy_slab(0,c,a,(int64_t)cnt*8,1);
    if(!strcmp(op,"mul")) return lolhip_mul_batch(P,0,a,b,B);
    if(!strcmp(op,"roundtrip")){ int r=lolhip_crt_batch(P,0,a,B); return r? r: lolhip_crtinv_batch(P,0,a,B);} return lolhip_polymul_batch(P,0,c,a,b,B); };
  unsigned long long* dst = nullptr; size_t nw = (size_t)B*T*64; 
  typedef int (*setfn)(unsigned long long*); setfn sf = (setfn)dlsym(RTLD_DEFAULT, "lolhip_debug_set_stamps");
  if(sf){ CK(hipMalloc(&dst, nw*32*8)); CK(hipMemset(dst,0,nw*32*8)); sf(dst); }
  for(int i=0;i<2;i++) if((rc=run())){ printf("run rc=%d\n",rc); return 1; }
  CK(hipDeviceSynchronize()); CK(hipEventRecord(e0,0)); for(int i=0;i<iters;i++) run(); CK(hipEventRecord(e1,0)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms,e0,e1)); ms/=iters;
  double bytes = ((!strcmp(op,"polymul")||!strcmp(op,"mul"))?3.0:(!strcmp(op,"roundtrip")?4.0:2.0))*cnt*8;
  if(sf){ run(); CK(hipDeviceSynchronize()); int wpb = (n/16>=256? n/16:256)/64; size_t nwv=(size_t)((B*T*(n/16>=256?1:1)))*wpb; std::vector<unsigned long long> hs(nwv*32);
    CK(hipMemcpy(hs.data(),dst,nwv*32*8,hipMemcpyDeviceToHost)); double sum[32]={0}; long cnt[32]={0}; int last=-1; 
    for(size_t w=0;w<nwv;w++){ int prev=-1; for(int i=0;i<32;i++){ if(hs[w*32+i]==0) continue; if(prev>=0){ sum[i]+= (double)(hs[w*32+i]-hs[w*32+prev]); cnt[i]++; } prev=i; } }
    { double clk=0; long nc=0; for(size_t w=0;w<nwv;w++){ unsigned long long t0=hs[w*32+26],t1=hs[w*32+27],r0=hs[w*32+28],r1=hs[w*32+29]; if(t0&&t1&&r1>r0){ clk += (double)(t1-t0)/(double)(r1-r0)*100.0; nc++; } }
      double mt=0, mr=0; for(size_t w=0;w<nwv;w++){ unsigned long long t0=hs[w*32+26],t1=hs[w*32+27],r0=hs[w*32+28],r1=hs[w*32+29]; if(t0&&t1&&r1>r0){ mt += (double)(t1-t0); mr += (double)(r1-r0); } }
      if(nc) printf("  in-kernel clock: %.0f MHz (s_memtime / s_memrealtime x 100 MHz, mean over %ld waves); loop lifetime per wave: %.0f ticks = %.1f us\n", clk/nc, nc, mt/nc, mr/nc/100.0); }
    if(getenv("LOLHIP_STAMP_DUMP")){ FILE* f=fopen(getenv("LOLHIP_STAMP_DUMP"),"wb"); fwrite(hs.data(),8,hs.size(),f); fclose(f); }
    for(int i=0;i<32;i++) if(cnt[i]) printf("  stamp %2d: +%8.0f cycles (avg over %ld waves)\n", i, sum[i]/cnt[i], cnt[i]); (void)last; }
  printf("%s m=%ld n=%ld T=%d B=%ld q~2^%d: %.4f ms/iter  %.3f M items/s  %.1f GB/s algorithmic (%.1f%% of 8 TB/s)\n", op, mval, n, T, B, qbits, ms, B/ms/1e3, bytes/ms/1e6, bytes/ms/1e6/80.0);
  return 0;
}
