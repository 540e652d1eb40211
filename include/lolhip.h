/*
 * include/lolhip.h — C ABI of liblolhip.so, the MI355X-native backend for the
 * Z_q hot path of Lol's `Tensor` class.
 *
 * Two groups of entry points:
 *
 *  (A) DROP-IN SYMBOLS: the nine `extern "C"` functions lol-cpp exports for Z_q and
 *      that lol-cpp's Haskell shim binds with `foreign import ccall unsafe`
 *      (lol-cpp/Crypto/Lol/Cyclotomic/Tensor/CPP/Backend.hs:304-337).  Same names,
 *      same argument order and meaning, host pointers, in place, one polynomial
 *      per call — so the existing `CT` shim links against liblolhip unchanged.
 *      `totm` is declared int64_t because that is what Haskell passes
 *      (Backend.hs:157; the C++ declares int32 hDim_t, types.h:22).
 *
 *  (B) BATCHED PLAN API: what a `lol-hip` backend binds (INTEGRATION.md shows the
 *      Haskell stubs).  A plan is built once per (prime powers, moduli); data
 *      stays in HBM; every call takes a leading batch dimension B.
 *
 * Data layout everywhere (reference: tensor.h:69, mul.cpp:21, Backend.hs:134-149):
 *   coefficient j of RNS component t of polynomial b is  y[(b*n + j)*T + t],
 *   int64, n = totient(m), T = tupSize.  Inputs may be in (-q_t, q_t); outputs are
 *   always canonical in [0, q_t) (zq.cpp:57-68).
 *
 * All functions are re-entrant (no global modulus, cf. types.h:59): any number of host threads may
 * call into one plan concurrently, on the same or on different streams; per-call workspaces are
 * stream-ordered allocations or caller-provided.  A plan belongs to the HIP device that was
 * current when it was created; calling it with another device current returns LOLHIP_ERR_DEVICE.
 * Nothing calls exit() (cf. ASSERT, types.h:36-41): errors come back as status codes.
 * There is NO CPU fallback: without a usable GPU every compute entry point
 * returns LOLHIP_ERR_NO_DEVICE.
 */
#ifndef LOLHIP_H
#define LOLHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LOLHIP_API __attribute__((visibility("default")))

/* == PrimeExponent (types.h:27-31) == Haskell CPP = (Int16, Int16) (Backend.hs:78) */
typedef struct { int16_t prime; int16_t exponent; } lolhip_pp;

enum {
  LOLHIP_OK = 0,
  LOLHIP_ERR_INVALID = -1,      /* malformed prime-power list / sizes                      */
  LOLHIP_ERR_MODULUS = -2,      /* modulus outside [2, 2^62) or required inverse missing  */
  LOLHIP_ERR_NO_CRT = -3,       /* no CRT basis mod some q_t (q not prime or m !| q-1)    */
  LOLHIP_ERR_ROOT = -4,         /* caller-supplied root of unity has the wrong order      */
  LOLHIP_ERR_NO_DEVICE = -5,    /* no HIP device: compute entry points refuse to run      */
  LOLHIP_ERR_HIP = -6,          /* HIP runtime error                                      */
  LOLHIP_ERR_NOT_DIVISIBLE = -7,/* divG: oddRad(m) not invertible mod some q_t            */
  LOLHIP_ERR_DEVICE = -8        /* the calling thread's current HIP device is not the one the plan's tables live on */
};

/* ------------------------------------------------------------------------- */
/* (A) drop-in symbols                                                         */
/* ------------------------------------------------------------------------- */

/* replaces crt.cpp:562-566.  ru[k] = pp_k*tupSize residues, ru[k][i*tupSize+t] =
 * omega_{pp_k,t}^i (CPP.hs:422-432); the roots the caller chose are honoured. */
LOLHIP_API void tensorCRTRq(int16_t tupSize, int64_t *y, int64_t totm, lolhip_pp *peArr,
                            int16_t sizeOfPE, int64_t **ru, int64_t *qs);
/* replaces crt.cpp:569-581 (takes inverse roots and mhat^-1 per component) */
LOLHIP_API void tensorCRTInvRq(int16_t tupSize, int64_t *y, int64_t totm, lolhip_pp *peArr,
                               int16_t sizeOfPE, int64_t **ruinv, int64_t *mhatInv, int64_t *qs);
/* replaces mul.cpp:27-30: a = zipWith (*) a b */
LOLHIP_API void mulRq(int16_t tupSize, int64_t *a, int64_t *b, int64_t totm, int64_t *qs);
/* replace l.cpp:109-115 / 150-156 */
LOLHIP_API void tensorLRq(int16_t tupSize, int64_t *y, int64_t totm, lolhip_pp *peArr, int16_t sizeOfPE, int64_t *qs);
LOLHIP_API void tensorLInvRq(int16_t tupSize, int64_t *y, int64_t totm, lolhip_pp *peArr, int16_t sizeOfPE, int64_t *qs);
/* replace g.cpp:130-134 / 146-150 */
LOLHIP_API void tensorGPowRq(int16_t tupSize, int64_t *y, int64_t totm, lolhip_pp *peArr, int16_t sizeOfPE, int64_t *qs);
LOLHIP_API void tensorGDecRq(int16_t tupSize, int64_t *y, int64_t totm, lolhip_pp *peArr, int16_t sizeOfPE, int64_t *qs);
/* replace g.cpp:186-207 / 239-260: return 1 on success, 0 on failure (CPP.hs:309-323).
 * Unlike the reference, y is left untouched when 0 is returned. */
LOLHIP_API int16_t tensorGInvPowRq(int16_t tupSize, int64_t *y, int64_t totm, lolhip_pp *peArr, int16_t sizeOfPE, int64_t *qs);
LOLHIP_API int16_t tensorGInvDecRq(int16_t tupSize, int64_t *y, int64_t totm, lolhip_pp *peArr, int16_t sizeOfPE, int64_t *qs);
/* status of the last drop-in call on this thread (the reference's signatures are void) */
LOLHIP_API int lolhip_last_status(void);

/* ------------------------------------------------------------------------- */
/* (B) batched plan API                                                        */
/* ------------------------------------------------------------------------- */

typedef struct lolhip_plan lolhip_plan;
typedef struct lolhip_ext lolhip_ext;

/* Build the plan for index m = prod pps (ascending primes, as ppsFact does,
 * FactoredDefs.hs:360-361) and moduli qs[0..T).  Roots follow Lol's rule: omega_m =
 * g0^((q-1)/m), g0 the smallest generator of Z_q^* (ZqBasic.hs:144-165).  A plan
 * without a CRT basis is still valid for L/G/mul/twacePowDec/embedPow/embedDec.
 * `host_only != 0` skips the device upload (table inspection on a machine
 * without a GPU; compute calls then fail with LOLHIP_ERR_NO_DEVICE). */
LOLHIP_API int lolhip_plan_create(const lolhip_pp *pps, int npps, const int64_t *qs, int T,
                                  int host_only, lolhip_plan **out);
/* Same, with caller-chosen roots: omega_pp[k*T+t] = primitive pp_k-th root mod q_t,
 * and mhatinv[T] (either may be NULL for the default rule). */
LOLHIP_API int lolhip_plan_create_roots(const lolhip_pp *pps, int npps, const int64_t *qs, int T,
                                        const int64_t *omega_pp, const int64_t *mhatinv,
                                        int host_only, lolhip_plan **out);
LOLHIP_API void lolhip_plan_destroy(lolhip_plan *p);

LOLHIP_API int64_t lolhip_plan_n(const lolhip_plan *p);      /* totient(m) */
LOLHIP_API int64_t lolhip_plan_m(const lolhip_plan *p);
LOLHIP_API int lolhip_plan_T(const lolhip_plan *p);
LOLHIP_API int lolhip_plan_has_crt(const lolhip_plan *p);
/* Host tables exactly as lol-cpp's shim would marshal them.  Each copies min(len,
 * available) int64 values and returns the available count.
 *   which: 0 ru[k] (CPP.hs:422-432)   1 ruInv[k] (CPP.hs:434-442)   2 mhatInv
 *          3 gCRT  4 gInvCRT (CPP.hs:444-454; [n*T] AoS)            5 qs
 *          10 / 11 (inspection, tests): the stage program a lone crt / crtInv of an index that is not a power
 *          of two launches, four values per stage: kind, prime (first level for the 2-power tiles), vector
 *          length (levels for the tiles), stride                                  */
LOLHIP_API int64_t lolhip_plan_table(const lolhip_plan *p, int which, int k, int64_t *out, int64_t len);

/* smallest prime > lower congruent to 1 mod m (head of goodQs, ZqBasic.hs:71-73) */
LOLHIP_API int64_t lolhip_good_q(int64_t m, int64_t lower);

/* --- device-pointer batch operations (y, a, b, c live in HBM; stream = hipStream_t
 *     or NULL).  In place unless noted.  Return LOLHIP_OK or an error. ------------- */
LOLHIP_API int lolhip_crt_batch   (const lolhip_plan *p, void *stream, int64_t *y, int64_t B);
LOLHIP_API int lolhip_crtinv_batch(const lolhip_plan *p, void *stream, int64_t *y, int64_t B);
/* a *= b pointwise (mulRq / zipWithT (*), CPP.hs:257-260) */
LOLHIP_API int lolhip_mul_batch   (const lolhip_plan *p, void *stream, int64_t *a, const int64_t *b, int64_t B);
/* c = crtInv(crt(a) * crt(b)): one cyclotomic ring product per batch item, operands and
 * result in the powerful basis (Cyc (*), Cyc.hs:262-297).  c may alias a or b. */
LOLHIP_API int lolhip_polymul_batch(const lolhip_plan *p, void *stream, int64_t *c, const int64_t *a,
                                    const int64_t *b, int64_t B);
LOLHIP_API int lolhip_l_batch      (const lolhip_plan *p, void *stream, int64_t *y, int64_t B);
LOLHIP_API int lolhip_linv_batch   (const lolhip_plan *p, void *stream, int64_t *y, int64_t B);
LOLHIP_API int lolhip_mulgpow_batch(const lolhip_plan *p, void *stream, int64_t *y, int64_t B);
LOLHIP_API int lolhip_mulgdec_batch(const lolhip_plan *p, void *stream, int64_t *y, int64_t B);
/* LOLHIP_ERR_NOT_DIVISIBLE (y untouched) iff the reference would return 0 */
LOLHIP_API int lolhip_divgpow_batch(const lolhip_plan *p, void *stream, int64_t *y, int64_t B);
LOLHIP_API int lolhip_divgdec_batch(const lolhip_plan *p, void *stream, int64_t *y, int64_t B);
/* mulGCRT / divGCRT: pointwise by gCRT / gInvCRT (CPP.hs:230-231) */
LOLHIP_API int lolhip_mulgcrt_batch(const lolhip_plan *p, void *stream, int64_t *y, int64_t B);
LOLHIP_API int lolhip_divgcrt_batch(const lolhip_plan *p, void *stream, int64_t *y, int64_t B);

/* --- floating-point members of the class, float64, tolerance contract:
 *     relative 1e-12 against lol-cpp on the same inputs (integer paths are bit-exact) --------
 * crtc / crtinvc: replace tensorCRTC / tensorCRTInvC (crt.cpp:583-598), the CRT over C with
 *   omega_m = exp(2 pi i / m) that UCyc uses when a modulus has no CRT basis (CRTExt,
 *   UCyc.hs:422-444).  y [B][n] complex doubles, (re, im) interleaved, in place; crtinvc includes
 *   the mhat^-1 scaling.  Works for any plan (the moduli play no role).
 * gaussian_dec: replaces tensorGaussianDec (random.cpp:19-64; caller CPP.hs:376-389): y [B][n]
 *   doubles, on entry iid real Gaussians, on exit the sample in the decoding basis (the caller
 *   draws and scales the Gaussians and rounds the result, as cDispatchGaussian does).
 * Both need n <= 8192 and every prime of m <= 13, else LOLHIP_ERR_INVALID. */
LOLHIP_API int lolhip_crtc_batch        (const lolhip_plan *p, void *stream, double *y, int64_t B);
LOLHIP_API int lolhip_crtinvc_batch     (const lolhip_plan *p, void *stream, double *y, int64_t B);
LOLHIP_API int lolhip_gaussian_dec_batch(const lolhip_plan *p, void *stream, double *y, int64_t B);

/* --- ring extension m | m' (twace/embed, Extension.hs:54-129) ------------------- */
LOLHIP_API int lolhip_ext_create(const lolhip_plan *p_m, const lolhip_plan *p_mprime, lolhip_ext **out);
LOLHIP_API void lolhip_ext_destroy(lolhip_ext *x);
/* out-of-place gathers; `lo` arrays are [B][n][T], `hi` arrays [B][n'][T] */
LOLHIP_API int lolhip_twace_powdec_batch(const lolhip_ext *x, void *stream, int64_t *lo_out, const int64_t *hi_in, int64_t B);
LOLHIP_API int lolhip_twace_crt_batch   (const lolhip_ext *x, void *stream, int64_t *lo_out, const int64_t *hi_in, int64_t B);
LOLHIP_API int lolhip_embed_pow_batch   (const lolhip_ext *x, void *stream, int64_t *hi_out, const int64_t *lo_in, int64_t B);
LOLHIP_API int lolhip_embed_dec_batch   (const lolhip_ext *x, void *stream, int64_t *hi_out, const int64_t *lo_in, int64_t B);
LOLHIP_API int lolhip_embed_crt_batch   (const lolhip_ext *x, void *stream, int64_t *hi_out, const int64_t *lo_in, int64_t B);
/* coeffs (class Tensor, Tensor.hs:174; CPP/Extension.hs:90-93): the phi(m')/phi(m) coefficient
 * vectors of each O_m' element with respect to the relative powerful (or decoding) basis,
 * lo_out [phi(m')/phi(m)][B][n][T] <- hi_in [B][n'][T].  Vector i1 pairs with the relative basis
 * element whose powerful-basis representation is the unit vector at index table5[i1*n]
 * (powBasisPow, Tensor.hs:177: sum_i1 embed(coeffs_i1 x) * b_i1 = x, CycTests.hs:71-76). */
LOLHIP_API int lolhip_coeffs_batch      (const lolhip_ext *x, void *stream, int64_t *lo_out, const int64_t *hi_in, int64_t B);
/* evalLin (lol Linear.hs:75-79): apply the E-linear function R -> S given by its values
 * ys_i in S on the relative decoding basis of R/E:
 *     out = sum_i ys_i * embed (coeffsDec_i r)
 * x_er: extension E in R, x_es: extension E in S (same plan for E in both).  r_dec [B][n_R][T]
 * in the decoding basis of R; ys_crt [n_R/n_E][n_S][T] in the CRT basis of S (what linearDec
 * stores, Linear.hs:68-72); out [B][n_S][T] in the CRT basis of S.  work: device scratch of
 * (n_R/n_E) * B * (n_E + n_S) * T int64.  A composition of the kernels above: coeffs gather,
 * embedDec, l, crt over all (n_R/n_E)*B polynomials at once, knapsack. */
LOLHIP_API int lolhip_evallin_batch(const lolhip_ext *x_er, const lolhip_ext *x_es, void *stream,
                                    const int64_t *r_dec, const int64_t *ys_crt, int64_t *out,
                                    int64_t *work, int64_t B);
/* tunnel (lol-apps SymmSHE.hs:549-570), the body after `toMSD . absorbGFactors`: for a linear
 * ciphertext [c0, c1] over R' (c0_dec in the decoding basis, c1_pow in the powerful basis, both
 * [B][n_R][T]),
 *     c0' = evalLin f c0;  c1s = coeffsPow c1 :: [E'];  c1' = sum_i switch hints_i (embed c1s_i)
 *     out = const c0' + c1'      [2][B][n_S][T], CRT basis of S'
 * ys_crt as for lolhip_evallin_batch; hints [n_R/n_E][L][2][n_S][T] (one KSHint per relative
 * powerful-basis element, CRT basis, L = lolhip_decompose_len of the S' plan); base as for
 * lolhip_keyswitch_batch.  work: lolhip_tunnel_work_len(...) int64 of device scratch. */
LOLHIP_API int64_t lolhip_tunnel_work_len(const lolhip_ext *x_er, const lolhip_ext *x_es, int64_t base, int64_t B);
LOLHIP_API int lolhip_tunnel_batch(const lolhip_ext *x_er, const lolhip_ext *x_es, void *stream,
                                   const int64_t *c0_dec, const int64_t *c1_pow, const int64_t *ys_crt,
                                   const int64_t *hints, int64_t base, int64_t *out, int64_t *work, int64_t B);
/* host index tables: which 0 extIndicesPowDec[n] 1 extIndicesCRT[n'] 2 embedPow[n'] (-1 = zero)
 * 3 embedDec[n'] (-1 zero, bit 30 = negate) 4 baseIndicesCRT[n'] (Tensor.hs:426-468)
 * 5 extIndicesCoeffs[n'/n][n] flattened (Tensor.hs:472-477) */
LOLHIP_API int64_t lolhip_ext_table(const lolhip_ext *x, int which, int32_t *out, int64_t len);

/* --- ring-level pipelines of SymmSHE, device pointers -----------
 * In the reference these are Haskell compositions of the Tensor methods above on one
 * ring element at a time; here each is one or two passes over a batch slab.  All slabs
 * are [.][B][n][T] int64, component t innermost.  Up to 16 RNS components.              */

/* Coefficients of  mulG <$> (c * d)  for two linear ciphertexts c = c0 + c1 s,
 * d = d0 + d1 s, every operand in the CRT basis (lol-apps SymmSHE.hs:444-449; mulG in the
 * CRT basis is the pointwise product with gCRT, CPP.hs:230):
 *   e0 = g c0 d0,  e1 = g (c0 d1 + c1 d0),  e2 = g c1 d1.
 * One pass: 4 reads, 3 writes.  Outputs may alias inputs. */
LOLHIP_API int lolhip_ctmul_crt_batch(const lolhip_plan *p, void *stream, const int64_t *c0, const int64_t *c1,
                                      const int64_t *d0, const int64_t *d1, int64_t *e0, int64_t *e1,
                                      int64_t *e2, int64_t B);

/* Gadget decomposition (Decompose gad (Cyc t m zq), Cyc.hs:592-604): base 0 = TrivGad
 * (ZqBasic.hs:227-232: one digit per component, its centred lift), base b >= 2 = BaseBGad b
 * (ZqBasic.hs:258-264: gadlen(b, q_t) centred base-b digits, Numeric.hs:202-205,227-234);
 * product rings concatenate, first component first (Gadget.hs:96-101).
 * lolhip_decompose_len: number of digits L (negative status on error).
 * lolhip_gadget: the gadget vector as [L][T] residues (b^k in its own component, zero
 *   elsewhere; Gadget.hs:92-94), host array of at least L*T entries; returns L.
 * lolhip_decompose_batch: c in the powerful basis [B][n][T] -> digits [L][B][n][T], every
 *   integer digit polynomial already reduced into all T components (`fmap reduce`,
 *   SymmSHE.hs:314). */
LOLHIP_API int lolhip_decompose_len(const lolhip_plan *p, int64_t base);
LOLHIP_API int lolhip_gadget(const lolhip_plan *p, int64_t base, int64_t *out, int64_t cap);
LOLHIP_API int lolhip_decompose_batch(const lolhip_plan *p, void *stream, const int64_t *c_pow, int64_t base,
                                      int64_t *digits, int64_t B);

/* knapsack (SymmSHE.hs:302-304): out_k = addend_k + sum_j xs_j * hint_jk, CRT basis.
 * xs [L][B][n][T]; hint [L][K][n][T], shared by the whole batch (K = 1..3 coefficients
 * of the hint polynomials); addend [K][B][n][T] or NULL; out [K][B][n][T] (may alias addend). */
LOLHIP_API int lolhip_knapsack_batch(const lolhip_plan *p, void *stream, const int64_t *xs_crt, int L,
                                     const int64_t *hint, int K, const int64_t *addend, int64_t *out, int64_t B);

/* `switch` (SymmSHE.hs:312-314) and with it keySwitchQuadCirc (:361-371, addend = the CRT
 * forms of c0, c1):  out = addend + knapsack hint (crt (reduce <$> decompose c2)).
 * c2 in the powerful basis [B][n][T]; work: caller-provided device scratch of
 * lolhip_decompose_len * B * n * T int64 (it holds the digit polynomials). */
LOLHIP_API int lolhip_keyswitch_batch(const lolhip_plan *p, void *stream, const int64_t *c2_pow, int64_t base,
                                      const int64_t *hint, int K, const int64_t *addend, int64_t *out,
                                      int64_t *work, int64_t B);

/* RescaleCyc (a,b) -> b (Cyc.hs:529-542): drop the first modulus of the tuple.  With
 * z = lift a coefficient-wise (powerful or decoding basis),
 *   out_s = q_0^-1 (c_s - z)  mod q_s   for s = 1..T-1.
 * c [B][n][T] -> out [B][n][T-1].  LOLHIP_ERR_MODULUS if q_0 is not invertible mod some q_s. */
LOLHIP_API int lolhip_rescale_drop_batch(const lolhip_plan *p, void *stream, const int64_t *c, int64_t *out,
                                         int64_t B);

/* --- host-pointer convenience (H2D, run, D2H on an internal stream) --------------
 * op: see LOLHIP_OP_*.  y (and b for MUL/POLYMUL) are host arrays of B polynomials. */
enum {
  LOLHIP_OP_CRT = 0, LOLHIP_OP_CRTINV = 1, LOLHIP_OP_MUL = 2, LOLHIP_OP_POLYMUL = 3,
  LOLHIP_OP_L = 4, LOLHIP_OP_LINV = 5, LOLHIP_OP_MULGPOW = 6, LOLHIP_OP_MULGDEC = 7,
  LOLHIP_OP_DIVGPOW = 8, LOLHIP_OP_DIVGDEC = 9, LOLHIP_OP_MULGCRT = 10, LOLHIP_OP_DIVGCRT = 11
};
LOLHIP_API int lolhip_op_host(const lolhip_plan *p, int op, int64_t *y, const int64_t *b, int64_t B);
enum {
  LOLHIP_EXT_TWACE_POWDEC = 0, LOLHIP_EXT_TWACE_CRT = 1, LOLHIP_EXT_EMBED_POW = 2,
  LOLHIP_EXT_EMBED_DEC = 3, LOLHIP_EXT_EMBED_CRT = 4, LOLHIP_EXT_COEFFS = 5
};
LOLHIP_API int lolhip_ext_host(const lolhip_ext *x, int op, int64_t *out, const int64_t *in, int64_t B);
/* The host-pointer calls (and the drop-in symbols, which go through them) check a staging set — one stream, a
 * pinned staging area and two device buffers, grown on demand — out of a process-wide pool for the duration of
 * the call: the steady state of a call is memcpy, H2D, kernels, D2H and a wait on that set's stream — no
 * allocation, no device-wide synchronisation.  The pool holds as many sets as calls have run concurrently; sets
 * belong to no thread, so exiting worker threads leave nothing behind.  This frees every idle set (optional;
 * e.g. to hand the memory back between phases).  The name dates from round 2, when the sets were per thread. */
LOLHIP_API void lolhip_thread_release(void);

/* --- wire format: Lol's protobuf ring elements, host side -------
 * message Rq { uint32 m = 1; uint64 q = 2; repeated sint64 xs = 3; }   (lol/Lol.proto)
 * message RqProduct { repeated Rq rqlist = 1; }
 * One Rq per RNS modulus, first component first; xs are decoding-basis coefficients written
 * as centred lifts (IZipVector.hs:127-205).  Ingest = read -> [n][T] slab -> l -> crt.
 * read : returns n (coefficients per modulus); fills m, T, qs[T] and, when xs != NULL, the
 *        reduced residues xs[j*T + t] in [0, q_t).  Pass xs = NULL to query sizes.
 * write: returns the number of bytes (needed when out == NULL, written otherwise); xs as
 *        above (any representative in (-q, q)).  Unpacked sint64 encoding (proto2 default);
 *        the reader also accepts the packed form.  Negative return = LOLHIP_ERR_*. */
LOLHIP_API int64_t lolhip_rqproduct_read(const uint8_t *buf, int64_t len, uint32_t *m, int64_t *qs, int cap_T,
                                         int *T, int64_t *xs, int64_t cap_xs);
LOLHIP_API int64_t lolhip_rqproduct_write(uint32_t m, const int64_t *qs, int T, const int64_t *xs, int64_t n,
                                          uint8_t *out, int64_t cap);

/* KSHint (lol-apps/SHE.proto: repeated RqPolynomial hint = 1; TypeRep gad = 2; RqPolynomial =
 * repeated RqProduct coeffs, constant coefficient first): the L hint polynomials of K
 * coefficients each -> xs [L][K][n][T], decoding basis, canonical residues; l and crt over the
 * L*K polynomials then give the hint slab of lolhip_keyswitch_batch.  Returns n; xs = NULL
 * queries L, K, T, m, qs.  The gadget fingerprint is skipped. */
LOLHIP_API int64_t lolhip_kshint_read(const uint8_t *buf, int64_t len, uint32_t *m, int64_t *qs, int cap_T, int *T,
                                      int *L, int *K, int64_t *xs, int64_t cap_xs);

/* The remaining messages of lol/Lol.proto and lol-apps/SHE.proto, same conventions:
 *  r_read          message R { m = 1; repeated sint64 xs = 2 }: integer coefficients, decoding basis.  Returns n.
 *  secretkey_read  message SecretKey { R sk = 1; double v = 2 } (SHE.proto:9).  Returns n.
 *  kqproduct_read  message KqProduct / Kq (`repeated double xs`): xs [n][T] doubles.  Returns n.
 *  linearrq_read   message LinearRq { e = 1; r = 2; repeated RqProduct coeffs = 3 } (Lol.proto:11): the
 *                  values of an E-linear function on the relative decoding basis, xs [C][n][T]; after l
 *                  and crt, the ys of lolhip_evallin_batch.  Returns n; xs = NULL queries the sizes.
 *  kshint_write    the inverse of kshint_read; gad_a/gad_b = the two words of the TypeRep fingerprint.
 *  tunnelhint_read message TunnelHint (SHE.proto:26): e, r, s, p and the byte ranges (offset, length
 *                  into buf) of the embedded LinearRq and KSHints, for the readers above.  Returns the
 *                  number of KSHints. */
LOLHIP_API int64_t lolhip_r_read(const uint8_t *buf, int64_t len, uint32_t *m, int64_t *xs, int64_t cap_xs);
LOLHIP_API int64_t lolhip_secretkey_read(const uint8_t *buf, int64_t len, uint32_t *m, double *v, int64_t *xs, int64_t cap_xs);
LOLHIP_API int64_t lolhip_kqproduct_read(const uint8_t *buf, int64_t len, uint32_t *m, int64_t *qs, int cap_T, int *T,
                                         double *xs, int64_t cap_xs);
LOLHIP_API int64_t lolhip_linearrq_read(const uint8_t *buf, int64_t len, uint32_t *e, uint32_t *r, int *C, uint32_t *m,
                                        int64_t *qs, int cap_T, int *T, int64_t *xs, int64_t cap_xs);
LOLHIP_API int64_t lolhip_kshint_write(uint32_t m, const int64_t *qs, int T, int L, int K, const int64_t *xs, int64_t n,
                                       uint64_t gad_a, uint64_t gad_b, uint8_t *out, int64_t cap);
LOLHIP_API int64_t lolhip_tunnelhint_read(const uint8_t *buf, int64_t len, uint32_t *e, uint32_t *r, uint32_t *s, uint64_t *p,
                                          int64_t *func_off, int64_t *func_len, int64_t *hint_off, int64_t *hint_len,
                                          int cap_hints);

/* Round 3: the remaining writers (same conventions as the readers above; out = NULL queries the size) and the chain
 * messages of lol-apps/HomomPRF.proto:18-26 (LinearFuncChain, TunnelHintChain, RoundHintChain: `repeated X = 1`).
 *  r_write / secretkey_write   message R / SecretKey from integer decoding-basis coefficients xs[n] (and the variance v)
 *  linearrq_write              message LinearRq from xs [C][n][T] (decoding-basis residues of the output ring m)
 *  tunnelhint_write            message TunnelHint from an encoded LinearRq, nh encoded KSHints and e, r, s, p
 *  chain_read                  number of elements + byte ranges (offset, length into buf) of up to cap of them
 *  chain_write                 the chain message of `count` encoded elements */
LOLHIP_API int64_t lolhip_r_write(uint32_t m, const int64_t *xs, int64_t n, uint8_t *out, int64_t cap);
LOLHIP_API int64_t lolhip_secretkey_write(uint32_t m, double v, const int64_t *xs, int64_t n, uint8_t *out, int64_t cap);
LOLHIP_API int64_t lolhip_linearrq_write(uint32_t e, uint32_t r, uint32_t m, const int64_t *qs, int T, int C, const int64_t *xs,
                                         int64_t n, uint8_t *out, int64_t cap);
LOLHIP_API int64_t lolhip_tunnelhint_write(const uint8_t *func, int64_t func_len, const uint8_t *const *hints, const int64_t *hint_len,
                                           int nh, uint32_t e, uint32_t r, uint32_t s, uint64_t p, uint8_t *out, int64_t cap);
LOLHIP_API int64_t lolhip_chain_read(const uint8_t *buf, int64_t len, int64_t *off, int64_t *elem_len, int cap);
LOLHIP_API int64_t lolhip_chain_write(const uint8_t *const *elems, const int64_t *elem_len, int count, uint8_t *out, int64_t cap);

/* Measurement aid: device-to-device copy of `bytes` (a multiple of 16; both pointers 16-byte aligned) with
 * 16 bytes per lane — the read-once/write-once ceiling bench.py quotes beside every HBM-bound leg.
 * variant 0: one tile per workgroup; 1: persistent workgroups. */
LOLHIP_API int lolhip_copy_slab(void *stream, void *dst, const void *src, int64_t bytes, int variant);

/* Test and A/B aid, not part of the drop-in surface: force a launch path.  `name` is one of
 * GENERIC_SCALAR, NO_FUSED2, NO_POW2_PART, POLYMUL_UNFUSED, KEYSWITCH_UNFUSED, NO_T1, NO_PIPE, FORCE_PIPE, NO_OWN_DIAG, NO_MERGE, NO_KRON and NO_LAZY (these four read when a plan is built) (each is
 * also read ONCE at first use from the environment variable LOLHIP_<name>); value 0 restores the
 * default path.  Every path computes the same residues.  Returns LOLHIP_OK or LOLHIP_ERR_INVALID. */
LOLHIP_API int lolhip_debug_set(const char *name, int value);

/* number of HIP devices visible (0 without a GPU); never initialises a context */
LOLHIP_API int lolhip_device_count(void);
LOLHIP_API const char *lolhip_version(void);

#ifdef __cplusplus
}
#endif
#endif /* LOLHIP_H */
