/*
 * oracle/cpu_ref.h — CPU restatement of the lol-cpp hot path.  TEST INFRASTRUCTURE.
 *
 * This is the checker, never the product: only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  liblolhip never links it.
 *
 * Every function restates one reference function (file:line cited in
 * cpu_ref.c) with two deliberate differences:
 *   - arithmetic is exact for any modulus q < 2^62 (unsigned __int128 products),
 *     whereas the reference's Zq multiplies int64*int64 and overflows for
 *     q >= ~2^31.5 (lol-cpp/.../CPP/types.h:79-84);
 *   - no process-global modulus (types.h:59): everything is re-entrant and
 *     takes a leading batch dimension B.
 *
 * Layout (identical to the reference's, tensor.h:69 / mul.cpp:21):
 *   element j of RNS component t of polynomial b lives at y[(b*n + j)*T + t].
 *
 * Parity pin: tests/test_oracle.py checks this file bit-for-bit against
 * oracle/_ref/libctensor.so (the reference's own C++, compiled by
 * oracle/Makefile) for q < 2^31, and against the closed-form definition
 *  for q up to 2^61.
 */
#ifndef LOL_ORACLE_CPU_REF_H
#define LOL_ORACLE_CPU_REF_H
#include <stdint.h>

typedef struct { int16_t prime; int16_t exponent; } ref_pp;  /* == PrimeExponent, types.h:27-31 */

#ifdef __cplusplus
extern "C" {
#endif

/* ru[k] has pp_k*T entries, ru[k][i*T+t] = omega_{pp_k,t}^i (CPP.hs:422-432, tensor.h:91). */
void ref_crt   (int T, int64_t *y, int64_t B, int64_t n, const ref_pp *pps, int npp,
                const int64_t *const *ru, const int64_t *qs);
void ref_crtinv(int T, int64_t *y, int64_t B, int64_t n, const ref_pp *pps, int npp,
                const int64_t *const *ruinv, const int64_t *mhatinv, const int64_t *qs);
void ref_mul   (int T, int64_t *a, const int64_t *b, int64_t B, int64_t n, const int64_t *qs);
void ref_l     (int T, int64_t *y, int64_t B, int64_t n, const ref_pp *pps, int npp, const int64_t *qs);
void ref_linv  (int T, int64_t *y, int64_t B, int64_t n, const ref_pp *pps, int npp, const int64_t *qs);
void ref_gpow  (int T, int64_t *y, int64_t B, int64_t n, const ref_pp *pps, int npp, const int64_t *qs);
void ref_gdec  (int T, int64_t *y, int64_t B, int64_t n, const ref_pp *pps, int npp, const int64_t *qs);
/* return 1 on success, 0 iff oddRad(m) is not invertible mod some q_t (g.cpp:194-199) */
int  ref_ginvpow(int T, int64_t *y, int64_t B, int64_t n, const ref_pp *pps, int npp, const int64_t *qs);
int  ref_ginvdec(int T, int64_t *y, int64_t B, int64_t n, const ref_pp *pps, int npp, const int64_t *qs);

/* Closed-form O(n^2) definition of crt, single RNS
 * component, for pinning the staged algorithm on small n. omega[k] = omega_{pp_k}. */
void ref_crt_naive(int64_t *out, const int64_t *in, int64_t n, const ref_pp *pps, int npp,
                   const int64_t *omega, int64_t q);

/* fused poly-mul: c = crtinv(crt(a) * crt(b)), the unit BASELINE.json's metric counts */
void ref_polymul(int T, int64_t *c, const int64_t *a, const int64_t *b, int64_t B, int64_t n,
                 const ref_pp *pps, int npp, const int64_t *const *ru,
                 const int64_t *const *ruinv, const int64_t *mhatinv, const int64_t *qs);

#ifdef __cplusplus
}
#endif
#endif
