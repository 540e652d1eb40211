/* tests/native/modea_driver.c — the link test of mode A (haskell/lol-hip/modeA/Makefile).
 * Linked as:  cc modea_driver.c -L<modeA build> -llolcpp_rest -L<repo>/lol_amd -llolhip -lstdc++ -lm -ldl
 * Checks, with the signatures Backend.hs:304-337 declares:
 *   - every one of the 29 symbols resolves (the link itself), none twice (no duplicate-definition error);
 *   - the nine Z_q symbols resolve to liblolhip.so, the other twenty to lol-cpp's code in this executable (dladdr);
 *   - tensorLR really is lol-cpp's (prefix sums along the (p-1)-axis of m = 3: l.cpp:28-57);
 *   - tensorCRTRq really is liblolhip's: without a GPU it reports LOLHIP_ERR_NO_DEVICE through lolhip_last_status
 *     and leaves the operand untouched; with one it returns the CRT of m = 4, q = 5.
 * Prints one line per check; exit status 0 iff all hold. */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

typedef struct { int16_t prime, exponent; } PrimeExponent;
/* the 29 symbols (types.h / tensor.h; totm is hDim_t = int32 in lol-cpp, Int64 from Haskell and in liblolhip) */
#define DECL(name) extern void name();
DECL(tensorLR) DECL(tensorLInvR) DECL(tensorLRq) DECL(tensorLInvRq) DECL(tensorLDouble) DECL(tensorLInvDouble) DECL(tensorLC)
DECL(tensorLInvC) DECL(tensorNormSqR) DECL(tensorNormSqD) DECL(tensorGPowR) DECL(tensorGPowRq) DECL(tensorGPowC) DECL(tensorGDecR)
DECL(tensorGDecRq) DECL(tensorGDecC) DECL(tensorGInvPowR) DECL(tensorGInvPowRq) DECL(tensorGInvPowC) DECL(tensorGInvDecR)
DECL(tensorGInvDecRq) DECL(tensorGInvDecC) DECL(tensorCRTRq) DECL(tensorCRTC) DECL(tensorCRTInvRq) DECL(tensorCRTInvC)
DECL(tensorGaussianDec) DECL(mulRq) DECL(mulC)
extern int lolhip_last_status(void);
extern int lolhip_device_count(void);

struct sym { const char* name; void (*fn)(); int zq; };
#define S(n, z) { #n, n, z }
static struct sym syms[] = {
  S(tensorLR, 0), S(tensorLInvR, 0), S(tensorLRq, 1), S(tensorLInvRq, 1), S(tensorLDouble, 0), S(tensorLInvDouble, 0), S(tensorLC, 0),
  S(tensorLInvC, 0), S(tensorNormSqR, 0), S(tensorNormSqD, 0), S(tensorGPowR, 0), S(tensorGPowRq, 1), S(tensorGPowC, 0), S(tensorGDecR, 0),
  S(tensorGDecRq, 1), S(tensorGDecC, 0), S(tensorGInvPowR, 0), S(tensorGInvPowRq, 1), S(tensorGInvPowC, 0), S(tensorGInvDecR, 0),
  S(tensorGInvDecRq, 1), S(tensorGInvDecC, 0), S(tensorCRTRq, 1), S(tensorCRTC, 0), S(tensorCRTInvRq, 1), S(tensorCRTInvC, 0),
  S(tensorGaussianDec, 0), S(mulRq, 1), S(mulC, 0),
};

int main(void) {
  int bad = 0, nzq = 0;
  for (unsigned i = 0; i < sizeof syms / sizeof syms[0]; ++i) {
    Dl_info info;
    memset(&info, 0, sizeof info);
    const int ok = dladdr((void*)syms[i].fn, &info) && info.dli_fname;
    const int in_hip = ok && strstr(info.dli_fname, "liblolhip") != NULL;
    nzq += syms[i].zq;
    if (!ok || in_hip != syms[i].zq) { ++bad; printf("WRONG  %-18s -> %s\n", syms[i].name, ok ? info.dli_fname : "?"); }
  }
  printf("%u symbols, %d of them Z_q -> liblolhip.so, %d misplaced\n", (unsigned)(sizeof syms / sizeof syms[0]), nzq, bad);

  /* lol-cpp's own tensorLR: m = 3 (one prime 3, exponent 1), two coefficients: y1 += y0 */
  PrimeExponent pe3 = {3, 1};
  int64_t y[2] = {5, 7};
  ((void (*)(int16_t, int64_t*, int64_t, PrimeExponent*, int16_t))tensorLR)(1, y, 2, &pe3, 1);
  const int l_ok = (y[0] == 5 && y[1] == 12);
  printf("tensorLR (lol-cpp): [5,7] -> [%lld,%lld] %s\n", (long long)y[0], (long long)y[1], l_ok ? "ok" : "WRONG");

  /* liblolhip's tensorCRTRq: m = 4 (n = 2), q = 5, omega_4 = 2 (2^2 = -1 mod 5): ru = [1, 2, 4, 3] */
  PrimeExponent pe4 = {2, 2};
  int64_t z[2] = {1, 1}, q = 5, ru0[4] = {1, 2, 4, 3};
  int64_t* ru[1] = {ru0};
  ((void (*)(int16_t, int64_t*, int64_t, PrimeExponent*, int16_t, int64_t**, int64_t*))tensorCRTRq)(1, z, 2, &pe4, 1, ru, &q);
  const int st = lolhip_last_status();
  int c_ok;
  if (lolhip_device_count() == 0) c_ok = (st == -5 /* LOLHIP_ERR_NO_DEVICE */ && z[0] == 1 && z[1] == 1);
  else c_ok = (st == 0 && z[0] == 3 && z[1] == 4);   /* 1 + w^1, 1 + w^3 with w = 2: 3, 1 + 8 = 9 = 4 mod 5 */
  printf("tensorCRTRq (liblolhip): status %d, [1,1] -> [%lld,%lld] %s\n", st, (long long)z[0], (long long)z[1], c_ok ? "ok" : "WRONG");
  return (bad == 0 && l_ok && c_ok) ? 0 : 1;
}
