/*
 * oracle/cpu_ref.c — CPU restatement of the lol-cpp Tensor hot path over Z_q.
 *
 * TEST INFRASTRUCTURE ONLY (see cpu_ref.h).  Written from the reference's
 * algorithm, not from its loop nests: every prime-index operator is expressed
 * once, generically in p, as "apply a small operator to each length-d vector
 * with stride rts" (the reference hand-unrolls p = 2,3,5,7 and mallocs for
 * p >= 11).  All paths in Reference column are relative to
 * /root/reference/lol-cpp/Crypto/Lol/Cyclotomic/Tensor/CPP/.
 *
 *   here               reference
 *   ----               ---------
 *   fuse_crt           tensorFuserCRT        tensor.h:76-95
 *   fuse_prime         tensorFuserPrime      tensor.h:39-74
 *   op_crtp            crtp                  crt.cpp:248-346  (generic arm :324-345)
 *   op_crtpinv         crtpinv               crt.cpp:349-457  (generic arm :433-456)
 *   op_dftp            dftp                  crt.cpp:131-246  (generic arm :226-245)
 *   crt_twiddle        crtTwiddle            crt.cpp:35-81
 *   dft_twiddle        dftTwiddle            crt.cpp:84-126
 *   digitrev           bitrev                crt.cpp:21-33
 *   pp_dft/pp_dftinv   ppDFT / ppDFTInv      crt.cpp:459-486 / 488-516
 *   pp_crt/pp_crtinv   ppcrt / ppcrtinv      crt.cpp:518-538 / 540-560
 *   ref_crt/ref_crtinv tensorCRTRq/InvRq     crt.cpp:562-566 / 569-581
 *   ref_mul            mulRq/zipWithStar     mul.cpp:14-30
 *   op_l/op_linv       lp / lpInv            l.cpp:28-57 / 67-98
 *   op_gpow/op_gdec    gPow / gDec           g.cpp:16-35 / 37-58
 *   op_ginvpow/ginvdec gInvPow / gInvDec     g.cpp:60-90 / 92-123
 *   ref_ginv*          tensorGInv{Pow,Dec}Rq g.cpp:186-207 / 239-260, oddRad :157-167
 *   inv_mod            reciprocal            zq.cpp:20-54
 *
 * Values are kept canonical in [0,q) at all times (the reference keeps them in
 * (-q,q) and fixes the sign once at the end, zq.cpp:57-68; the residues are the
 * same).  Inputs outside [0,q) are reduced first.
 */
#include "cpu_ref.h"
#include <stdlib.h>
#include <string.h>

typedef uint64_t u64;
typedef unsigned __int128 u128;

static inline u64 mulm(u64 a, u64 b, u64 q) { return (u64)(((u128)a * b) % q); }
static inline u64 addm(u64 a, u64 b, u64 q) { u64 s = a + b; return s >= q ? s - q : s; }
static inline u64 subm(u64 a, u64 b, u64 q) { return a >= b ? a - b : a + q - b; }
static inline u64 canon(int64_t x, u64 q) {
  int64_t r = x % (int64_t)q;
  return (u64)(r < 0 ? r + (int64_t)q : r);
}

static int64_t ipow64(int64_t b, int e) { int64_t r = 1; while (e-- > 0) r *= b; return r; }

/* base-p digit reversal over `digits` digits (crt.cpp:21-33) */
static int64_t digitrev(int p, int digits, int64_t j) {
  int64_t acc = 0;
  for (int e = digits - 1; e >= 0; --e) { acc += (j % p) * ipow64(p, e); j /= p; }
  return acc;
}

/* b^-1 mod a, or 0 if not invertible (zq.cpp:20-54), canonical */
static u64 inv_mod(u64 a, u64 b) {
  int64_t r0 = (int64_t)a, r1 = (int64_t)(b % a), t0 = 0, t1 = 1;
  while (r1 != 0) {
    int64_t quo = r0 / r1, tmp = r0 % r1;
    r0 = r1; r1 = tmp;
    tmp = t1; t1 = t0 - quo * t1; t0 = tmp;
  }
  if (r0 != 1) return 0;
  return canon(t0, a);
}

/* ---- one planar RNS component of one polynomial: v[0..n), modulus q ------- */

typedef struct {
  u64 *v;        /* n residues, planar */
  u64 *tmp;      /* scratch, >= max prime */
  u64 *x;        /* scratch, >= max prime */
  u64 q;
} comp_t;

/* ---- prime-index operators on each (blk, r): vector v[(blk*d+i)*rts + r] --- */

typedef void (*vecop_t)(const comp_t *c, u64 *x, u64 *out, int p, const u64 *w, int64_t rustride);

static void apply_vecop(const comp_t *c, int64_t lts, int64_t rts, int d, int p,
                        vecop_t f, const u64 *w, int64_t rustride) {
  for (int64_t blk = 0; blk < lts; ++blk)
    for (int64_t r = 0; r < rts; ++r) {
      u64 *base = c->v + blk * d * rts + r;
      for (int i = 0; i < d; ++i) c->x[i] = base[(int64_t)i * rts];
      f(c, c->x, c->tmp, p, w, rustride);
      for (int i = 0; i < d; ++i) base[(int64_t)i * rts] = c->tmp[i];
    }
}

/* crt.cpp:324-345: out[row-1] = sum_{col<p-1} x[col]*w[(col*row mod p)*rustride], row=1..p-1 */
static void op_crtp(const comp_t *c, u64 *x, u64 *out, int p, const u64 *w, int64_t rs) {
  for (int row = 1; row < p; ++row) {
    u64 acc = 0;
    for (int col = 0; col < p - 1; ++col)
      acc = addm(acc, mulm(x[col], w[((int64_t)col * row % p) * rs], c->q), c->q);
    out[row - 1] = acc;
  }
}
/* crt.cpp:433-456 */
static void op_crtpinv(const comp_t *c, u64 *x, u64 *out, int p, const u64 *w, int64_t rs) {
  u64 shift = 0;
  for (int row = 0; row < p - 1; ++row)
    shift = addm(shift, mulm(x[row], w[(int64_t)(p - row - 1) * rs], c->q), c->q);
  for (int row = 0; row < p - 1; ++row) {
    u64 acc = 0;
    for (int col = 0; col < p - 1; ++col)
      acc = addm(acc, mulm(x[col], w[((int64_t)row * (col + 1) % p) * rs], c->q), c->q);
    out[row] = subm(acc, shift, c->q);
  }
}
/* crt.cpp:226-245: full p-point DFT */
static void op_dftp(const comp_t *c, u64 *x, u64 *out, int p, const u64 *w, int64_t rs) {
  for (int row = 0; row < p; ++row) {
    u64 acc = 0;
    for (int col = 0; col < p; ++col)
      acc = addm(acc, mulm(x[col], w[((int64_t)col * row % p) * rs], c->q), c->q);
    out[row] = acc;
  }
}
/* l.cpp:28-57: y_i += y_{i-1}, ascending (prefix sums) */
static void op_l(const comp_t *c, u64 *x, u64 *out, int p, const u64 *w, int64_t rs) {
  (void)w; (void)rs;
  out[0] = x[0];
  for (int i = 1; i < p - 1; ++i) out[i] = addm(out[i - 1], x[i], c->q);
}
/* l.cpp:67-98: adjacent differences */
static void op_linv(const comp_t *c, u64 *x, u64 *out, int p, const u64 *w, int64_t rs) {
  (void)w; (void)rs;
  out[0] = x[0];
  for (int i = 1; i < p - 1; ++i) out[i] = subm(x[i], x[i - 1], c->q);
}
/* g.cpp:16-35 */
static void op_gpow(const comp_t *c, u64 *x, u64 *out, int p, const u64 *w, int64_t rs) {
  (void)w; (void)rs;
  u64 last = x[p - 2];
  for (int i = p - 2; i != 0; --i) out[i] = addm(x[i], subm(last, x[i - 1], c->q), c->q);
  out[0] = addm(x[0], last, c->q);
}
/* g.cpp:37-58 */
static void op_gdec(const comp_t *c, u64 *x, u64 *out, int p, const u64 *w, int64_t rs) {
  (void)w; (void)rs;
  u64 acc = x[0];
  for (int i = p - 2; i != 0; --i) { acc = addm(acc, x[i], c->q); out[i] = subm(x[i], x[i - 1], c->q); }
  out[0] = addm(x[0], acc, c->q);
}
/* g.cpp:60-90 */
static void op_ginvpow(const comp_t *c, u64 *x, u64 *out, int p, const u64 *w, int64_t rs) {
  (void)w; (void)rs;
  u64 lelts = 0, relts = 0;
  for (int i = 0; i < p - 1; ++i) lelts = addm(lelts, x[i], c->q);
  for (int i = p - 2; i >= 0; --i) {
    u64 z = x[i];
    u64 lmul = (u64)(p - 1 - i) % c->q, rmul = (u64)(i + 1) % c->q;
    out[i] = subm(mulm(lmul, lelts, c->q), mulm(rmul, relts, c->q), c->q);
    lelts = subm(lelts, z, c->q);
    relts = addm(relts, z, c->q);
  }
}
/* g.cpp:92-123 */
static void op_ginvdec(const comp_t *c, u64 *x, u64 *out, int p, const u64 *w, int64_t rs) {
  (void)w; (void)rs;
  u64 lastOut = 0;
  for (int i = 1; i < p; ++i) lastOut = addm(lastOut, mulm((u64)i % c->q, x[i - 1], c->q), c->q);
  u64 rp = (u64)p % c->q, acc = lastOut;
  for (int i = p - 2; i > 0; --i) {
    u64 t = acc;
    acc = subm(acc, mulm(x[i], rp, c->q), c->q);
    out[i] = t;
  }
  out[0] = acc;
}

/* ---- diagonal twiddles --------------------------------------------------- */

/* crt.cpp:35-81 (both arms: for p == 2, (p-1) == 1 and i1 == 0) */
static void crt_twiddle(const comp_t *c, int64_t lts, int64_t rts, int p, int e, const u64 *w) {
  int64_t mprime = ipow64(p, e - 1), blockDim = rts * (p - 1) * mprime;
  for (int64_t i0 = 1; i0 < mprime; ++i0) {
    int64_t rev = digitrev(p, e - 1, i0);
    for (int i1 = 0; i1 < p - 1; ++i1) {
      u64 twid = w[rev * (i1 + 1)];
      for (int64_t blk = 0; blk < lts; ++blk)
        for (int64_t r = 0; r < rts; ++r) {
          u64 *y = c->v + blk * blockDim + (i0 * (p - 1) + i1) * rts + r;
          *y = mulm(*y, twid, c->q);
        }
    }
  }
}
/* crt.cpp:84-126; `e` is the exponent handed to dftTwiddle (digit reversal uses e-1 digits) */
static void dft_twiddle(const comp_t *c, int64_t lts, int64_t rts, int p, int e, int64_t dim,
                        int64_t rustride, const u64 *w) {
  int64_t mprime = dim / p, temp1 = rts * dim;
  for (int64_t i0 = 1; i0 < mprime; ++i0) {
    int64_t rev = digitrev(p, e - 1, i0);
    for (int i1 = 1; i1 < p; ++i1) {
      u64 twid = w[rev * i1 * rustride];
      for (int64_t blk = 0; blk < lts; ++blk)
        for (int64_t r = 0; r < rts; ++r) {
          u64 *y = c->v + blk * temp1 + rts * (i0 * p + i1) + r;
          *y = mulm(*y, twid, c->q);
        }
    }
  }
}

/* ---- prime-power transforms ---------------------------------------------- */

/* crt.cpp:459-486 */
static void pp_dft(const comp_t *c, int64_t lts, int64_t rts, int p, int e, int64_t rustride, const u64 *w) {
  if (e == 0) return;
  int64_t primeRuStride = rustride * ipow64(p, e - 1);
  int64_t ltsScale = ipow64(p, e - 1), rtsScale = 1, twidRuStride = rustride;
  int ecur = e;
  for (int i = 0; i < e; ++i) {
    int64_t rtsDim = rts * rtsScale;
    apply_vecop(c, lts * ltsScale, rtsDim, p, p, op_dftp, w, primeRuStride);
    dft_twiddle(c, lts, rtsDim, p, ecur, ltsScale * p, twidRuStride, w);
    ltsScale /= p; rtsScale *= p; twidRuStride *= p; ecur -= 1;
  }
}
/* crt.cpp:488-516 */
static void pp_dftinv(const comp_t *c, int64_t lts, int64_t rts, int p, int e, int64_t rustride, const u64 *w) {
  if (e == 0) return;
  int64_t primeRuStride = rustride * ipow64(p, e - 1);
  int64_t ltsScale = 1, rtsScale = ipow64(p, e - 1), twidRuStride = primeRuStride;
  int ecur = 1;
  for (int i = 0; i < e; ++i) {
    int64_t rtsDim = rts * rtsScale, ltsScaleP = ltsScale * p;
    dft_twiddle(c, lts, rtsDim, p, ecur, ltsScaleP, twidRuStride, w);
    apply_vecop(c, lts * ltsScale, rtsDim, p, p, op_dftp, w, primeRuStride);
    ltsScale = ltsScaleP; rtsScale /= p; twidRuStride /= p; ecur += 1;
  }
}
/* crt.cpp:518-538 */
static void pp_crt(const comp_t *c, int64_t lts, int64_t rts, int p, int e, const u64 *w) {
  int64_t mprime = ipow64(p, e - 1);
  if (p != 2) apply_vecop(c, lts * mprime, rts, p - 1, p, op_crtp, w, mprime);
  crt_twiddle(c, lts, rts, p, e, w);
  pp_dft(c, lts, rts * (p - 1), p, e - 1, p, w);
}
/* crt.cpp:540-560 */
static void pp_crtinv(const comp_t *c, int64_t lts, int64_t rts, int p, int e, const u64 *w) {
  int64_t mprime = ipow64(p, e - 1);
  pp_dftinv(c, lts, rts * (p - 1), p, e - 1, p, w);
  crt_twiddle(c, lts, rts, p, e, w);
  if (p != 2) apply_vecop(c, lts * mprime, rts, p - 1, p, op_crtpinv, w, mprime);
}

/* ---- gather / scatter one component; scratch management ------------------- */

static int max_prime(const ref_pp *pps, int npp) {
  int m = 2;
  for (int i = 0; i < npp; ++i) if (pps[i].prime > m) m = pps[i].prime;
  return m;
}
static int comp_init(comp_t *c, int64_t n, int maxp) {
  c->v = (u64 *)malloc(sizeof(u64) * (size_t)(n > 0 ? n : 1));
  c->tmp = (u64 *)malloc(sizeof(u64) * (size_t)maxp);
  c->x = (u64 *)malloc(sizeof(u64) * (size_t)maxp);
  return c->v && c->tmp && c->x;
}
static void comp_free(comp_t *c) { free(c->v); free(c->tmp); free(c->x); }
static void comp_load(comp_t *c, const int64_t *y, int64_t n, int T, int t, u64 q) {
  c->q = q;
  for (int64_t j = 0; j < n; ++j) c->v[j] = canon(y[j * T + t], q);
}
static void comp_store(const comp_t *c, int64_t *y, int64_t n, int T, int t) {
  for (int64_t j = 0; j < n; ++j) y[j * T + t] = (int64_t)c->v[j];
}
/* planar copy of one component of a twiddle table: w[i] = ru_k[i*T+t] */
static u64 *ru_component(const int64_t *ruk, int64_t pp, int T, int t, u64 q) {
  u64 *w = (u64 *)malloc(sizeof(u64) * (size_t)pp);
  for (int64_t i = 0; i < pp; ++i) w[i] = canon(ruk[i * T + t], q);
  return w;
}

/* tensor.h:76-95 */
static void fuse_crt(const comp_t *c, int64_t n, const ref_pp *pps, int npp, u64 *const *w, int inverse) {
  int64_t lts = n, rts = 1;
  for (int k = 0; k < npp; ++k) {
    int p = pps[k].prime, e = pps[k].exponent;
    int64_t dim = (p - 1) * ipow64(p, e - 1);
    lts /= dim;
    if (inverse) pp_crtinv(c, lts, rts, p, e, w[k]); else pp_crt(c, lts, rts, p, e, w[k]);
    rts *= dim;
  }
}
/* tensor.h:39-74: A_{p^e} = I_{p^(e-1)} (x) A_p */
static void fuse_prime(const comp_t *c, int64_t n, const ref_pp *pps, int npp, vecop_t f) {
  int64_t lts = n, rts = 1;
  for (int k = 0; k < npp; ++k) {
    int p = pps[k].prime, e = pps[k].exponent;
    int64_t ipow_pe = ipow64(p, e - 1), dim = (p - 1) * ipow_pe;
    lts /= dim;
    if (p != 2) apply_vecop(c, lts * ipow_pe, rts, p - 1, p, f, 0, 0);
    rts *= dim;
  }
}

/* ---- exported ------------------------------------------------------------ */

static void crt_any(int T, int64_t *y, int64_t B, int64_t n, const ref_pp *pps, int npp,
                    const int64_t *const *ru, const int64_t *scale, const int64_t *qs, int inverse) {
  comp_t c;
  if (!comp_init(&c, n, max_prime(pps, npp))) abort();
  u64 **w = (u64 **)malloc(sizeof(u64 *) * (size_t)(npp > 0 ? npp : 1));
  for (int t = 0; t < T; ++t) {
    u64 q = (u64)qs[t];
    for (int k = 0; k < npp; ++k)
      w[k] = ru_component(ru[k], ipow64(pps[k].prime, pps[k].exponent), T, t, q);
    u64 s = scale ? canon(scale[t], q) : 1;
    for (int64_t b = 0; b < B; ++b) {
      int64_t *yb = y + b * n * T;
      comp_load(&c, yb, n, T, t, q);
      fuse_crt(&c, n, pps, npp, w, inverse);
      if (scale) for (int64_t j = 0; j < n; ++j) c.v[j] = mulm(c.v[j], s, q);   /* crt.cpp:573-579 */
      comp_store(&c, yb, n, T, t);
    }
    for (int k = 0; k < npp; ++k) free(w[k]);
  }
  free(w);
  comp_free(&c);
}

void ref_crt(int T, int64_t *y, int64_t B, int64_t n, const ref_pp *pps, int npp,
             const int64_t *const *ru, const int64_t *qs) {
  crt_any(T, y, B, n, pps, npp, ru, 0, qs, 0);
}
void ref_crtinv(int T, int64_t *y, int64_t B, int64_t n, const ref_pp *pps, int npp,
                const int64_t *const *ruinv, const int64_t *mhatinv, const int64_t *qs) {
  crt_any(T, y, B, n, pps, npp, ruinv, mhatinv, qs, 1);
}

/* mul.cpp:14-30 */
void ref_mul(int T, int64_t *a, const int64_t *b, int64_t B, int64_t n, const int64_t *qs) {
  for (int64_t i = 0; i < B * n; ++i)
    for (int t = 0; t < T; ++t) {
      u64 q = (u64)qs[t];
      a[i * T + t] = (int64_t)mulm(canon(a[i * T + t], q), canon(b[i * T + t], q), q);
    }
}

static void prime_any(int T, int64_t *y, int64_t B, int64_t n, const ref_pp *pps, int npp,
                      const int64_t *qs, vecop_t f, const u64 *scale) {
  comp_t c;
  if (!comp_init(&c, n, max_prime(pps, npp))) abort();
  for (int t = 0; t < T; ++t) {
    u64 q = (u64)qs[t];
    for (int64_t b = 0; b < B; ++b) {
      int64_t *yb = y + b * n * T;
      comp_load(&c, yb, n, T, t, q);
      fuse_prime(&c, n, pps, npp, f);
      if (scale) for (int64_t j = 0; j < n; ++j) c.v[j] = mulm(c.v[j], scale[t], q);
      comp_store(&c, yb, n, T, t);
    }
  }
  comp_free(&c);
}

void ref_l   (int T, int64_t *y, int64_t B, int64_t n, const ref_pp *pps, int npp, const int64_t *qs) { prime_any(T, y, B, n, pps, npp, qs, op_l, 0); }
void ref_linv(int T, int64_t *y, int64_t B, int64_t n, const ref_pp *pps, int npp, const int64_t *qs) { prime_any(T, y, B, n, pps, npp, qs, op_linv, 0); }
void ref_gpow(int T, int64_t *y, int64_t B, int64_t n, const ref_pp *pps, int npp, const int64_t *qs) { prime_any(T, y, B, n, pps, npp, qs, op_gpow, 0); }
void ref_gdec(int T, int64_t *y, int64_t B, int64_t n, const ref_pp *pps, int npp, const int64_t *qs) { prime_any(T, y, B, n, pps, npp, qs, op_gdec, 0); }

/* g.cpp:157-167 and :186-207 / :239-260.  Unlike the reference (which has already
 * transformed y when it discovers a non-invertible oddRad and returns 0 half-way),
 * this checks invertibility first and leaves y untouched on failure; the Haskell
 * caller discards y in that case (CPP.hs:321-323). */
static int ginv_any(int T, int64_t *y, int64_t B, int64_t n, const ref_pp *pps, int npp,
                    const int64_t *qs, vecop_t f) {
  u64 oddrad = 1;
  for (int k = 0; k < npp; ++k) if (pps[k].prime != 2) oddrad *= (u64)pps[k].prime;
  u64 *inv = (u64 *)malloc(sizeof(u64) * (size_t)T);
  for (int t = 0; t < T; ++t) {
    inv[t] = inv_mod((u64)qs[t], oddrad % (u64)qs[t]);
    if (inv[t] == 0) { free(inv); return 0; }
  }
  prime_any(T, y, B, n, pps, npp, qs, f, inv);
  free(inv);
  return 1;
}
int ref_ginvpow(int T, int64_t *y, int64_t B, int64_t n, const ref_pp *pps, int npp, const int64_t *qs) { return ginv_any(T, y, B, n, pps, npp, qs, op_ginvpow); }
int ref_ginvdec(int T, int64_t *y, int64_t B, int64_t n, const ref_pp *pps, int npp, const int64_t *qs) { return ginv_any(T, y, B, n, pps, npp, qs, op_ginvdec); }

/* SURVEY.md Appendix A (from Tensor.hs:359-368, tensor.h:46-73):
 *   y[i] = sum_j a[j] * prod_k omega_{pp_k}^( pow_k(j_k) * zms_k(i_k) )
 *   zms(i) = p*floor(i/(p-1)) + i mod (p-1) + 1
 *   pow(j) = p^(e-1) * (j mod (p-1)) + digitrev_{p,e-1}(floor(j/(p-1)))          */
static u64 powm(u64 b, u64 e, u64 q) { u64 r = 1 % q; b %= q; while (e) { if (e & 1) r = mulm(r, b, q); b = mulm(b, b, q); e >>= 1; } return r; }
void ref_crt_naive(int64_t *out, const int64_t *in, int64_t n, const ref_pp *pps, int npp,
                   const int64_t *omega, int64_t q_) {
  u64 q = (u64)q_;
  for (int64_t i = 0; i < n; ++i) {
    u64 acc = 0;
    for (int64_t j = 0; j < n; ++j) {
      int64_t ii = i, jj = j;
      u64 tw = 1 % q;
      for (int k = 0; k < npp; ++k) {
        int p = pps[k].prime, e = pps[k].exponent;
        int64_t phi = (p - 1) * ipow64(p, e - 1), pp = (int64_t)p * ipow64(p, e - 1);
        int64_t ik = ii % phi, jk = jj % phi; ii /= phi; jj /= phi;
        int64_t zms = p * (ik / (p - 1)) + ik % (p - 1) + 1;
        int64_t pw = ipow64(p, e - 1) * (jk % (p - 1)) + digitrev(p, e - 1, jk / (p - 1));
        tw = mulm(tw, powm(canon(omega[k], q), (u64)((pw * zms) % pp), q), q);
      }
      acc = addm(acc, mulm(canon(in[j], q), tw, q), q);
    }
    out[i] = (int64_t)acc;
  }
}

void ref_polymul(int T, int64_t *c, const int64_t *a, const int64_t *b, int64_t B, int64_t n,
                 const ref_pp *pps, int npp, const int64_t *const *ru,
                 const int64_t *const *ruinv, const int64_t *mhatinv, const int64_t *qs) {
  size_t bytes = sizeof(int64_t) * (size_t)(B * n * T);
  int64_t *bb = (int64_t *)malloc(bytes ? bytes : 1);
  if (c != a) memcpy(c, a, bytes);
  memcpy(bb, b, bytes);
  ref_crt(T, c, B, n, pps, npp, ru, qs);
  ref_crt(T, bb, B, n, pps, npp, ru, qs);
  ref_mul(T, c, bb, B, n, qs);
  ref_crtinv(T, c, B, n, pps, npp, ruinv, mhatinv, qs);
  free(bb);
}
