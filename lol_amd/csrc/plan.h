// lol_amd/csrc/plan.h — the run-time "plan": everything the reference recomputes or
// re-marshals per call (twiddle vectors CPP.hs:422-442, g vectors CPP.hs:444-454,
// moduli Backend.hs:195-199, prime-power list CPP.hs:325-337), built once per
// (prime powers, moduli, roots) and kept resident in HBM.
#pragma once
#include <cstdint>
#include <vector>

#include "hostmath.h"
#include "zq_dev.h"

namespace lolhip {

// ---- generic (any m) stage program ----------------------------------------
// One stage = (I_lts (x) A (x) I_rts) on a length-n planar component, written
// out-of-place:  out[x] = post[x] * sum_c M(row,c) * (pre * in)[base + c*rts]
// (tensor.h:39-95 gives the lts/rts bookkeeping this flattens).
enum StageKind : int32_t {
  ST_DIAG = 0,     // out[x] = tw[x'] * in[x]                    crtTwiddle (crt.cpp:35-81)
  ST_DFTP = 1,     // p-point DFT, optional post/pre twiddle     dftp+dftTwiddle (crt.cpp:131-246, 84-126)
  ST_CRTP = 2,     // (p-1)-point CRT_p                          crtp (crt.cpp:248-346)
  ST_CRTPINV = 3,  // scaled inverse CRT_p                       crtpinv (crt.cpp:349-457)
  ST_L = 4,        // prefix sums                                lp (l.cpp:28-57)
  ST_LINV = 5,     // adjacent differences                       lpInv (l.cpp:67-98)
  ST_GPOW = 6,     //                                            gPow (g.cpp:16-35)
  ST_GDEC = 7,     //                                            gDec (g.cpp:37-58)
  ST_GINVPOW = 8,  //                                            gInvPow (g.cpp:60-90)
  ST_GINVDEC = 9,  //                                            gInvDec (g.cpp:92-123)
  ST_SCALE = 10,   // out[x] = s * in[x]  (mhatInv crt.cpp:573-579, oddRad^-1 g.cpp:194-204)
  ST_GAUSS = 11,   // real (p-1) x (p-1) map of tensorGaussianDec           primeD (random.cpp:19-50)
  // Levels p .. p+d-1 (1-based, d <= 4) of the negacyclic 2-power transform on every contiguous block of
  // tw_mod = 2^(e-1) coefficients (the innermost tensor factor of m = 2^e * odd, tensor.h:46-73), one
  // 2^d-element register tile per thread: level s pairs x and x + 2^(s-1) inside blocks of 2^s with the
  // twiddle table entry 2^(s-1) + (x mod 2^(s-1)) (the same merged-twist network as the m = 2^k kernels;
  // CRT_{2^e} of crt.cpp:459-538 in a different but exact factorisation).  rts = 2^(p-1).  Z_q programs of
  // the vector interpreter only (mixed_impl.h).
  ST_POW2F = 12,   // forward: X' = X + w Y, Y' = X - w Y, levels ascending
  ST_POW2I = 13    // inverse: X' = X + Y, Y' = w^-1 (X - Y), levels descending; mat_off >= 0: level 1 also carries mhat^-1
};

struct Stage {
  int32_t kind;
  int32_t p;        // prime
  int32_t d;        // vector length (p or p-1; 1 for DIAG/SCALE)
  int32_t rts;      // stride between vector elements
  int32_t wp_off;   // offset (u64 units, per component) of omega_p^j table, j < p
  int32_t tw_off;   // offset of diagonal table or -1
  int32_t tw_mod;   // diagonal index = (x / tw_div) % tw_mod
  int32_t tw_div;
  int32_t mat_off;  // offset of the d x d coefficient matrix (DFTP/CRTP/CRTPINV), row-major
  // tw_per > 0: the diagonal belongs to THIS stage's own vectors (tw_div == rts, d | tw_mod), so element i of the
  // vector in block blk takes entry (blk mod tw_per) * d + i — one division per vector instead of two per element
  int32_t tw_per;   // tw_mod / d, or 0
  int32_t pad[2];
  // multiply-shift reciprocals (floor(2^40/v)+1) of rts, d, tw_div, tw_mod, tw_per: x/v = (x*M)>>40 for x < 2^20
  uint64_t m_rts, m_d, m_twdiv, m_twmod, m_twper;
};

struct StageProgram {
  std::vector<Stage> stages;   // host copy
  Stage* d_stages = nullptr;   // device copy
  int nstages = 0;
  bool big = false;            // holds a dense vector of length 18, 20 (merged 3^3 / 5^2) or 8 (3 (x) 5): the BIG kernels of classes 2 / 4 only
};

// ---- power-of-two fast path -------------------------------------------------
struct Pow2Tables {
  int L = 0;                       // n = 2^L
  // per component t, n Shoup pairs: entry (N/2 + i) = psi_N^(2i+1), N = 2..n (entry 0 unused)
  u64* d_tw_fwd = nullptr;         // [T][n][2]
  u64* d_tw_inv = nullptr;         // [T][n][2]  inverse twiddles; the level-1 entry is pre-scaled
  u64* d_scale = nullptr;          // [T][8]     Shoup pairs of mhatInv, the level-1 inverse twiddle times mhatInv, and both times 2^64 (plan.cpp)
  // the same three tables as 32-bit Shoup pairs (w, floor(w*2^32/q)) when every q_t < 2^31
  uint32_t *d_tw_fwd32 = nullptr, *d_tw_inv32 = nullptr, *d_scale32 = nullptr;
  int arith32 = 0;                 // 4: every q_t < 2^27; 2: < 2^30; 3: < 2^31; 0: no 32-bit tables
};

struct Plan {
  std::vector<PP> pps;
  int T = 0;
  i64 m = 0, n = 0;
  std::vector<u64> qs;
  bool has_crt = false;            // every q_t prime with m | q_t - 1 (or caller-supplied roots)
  bool device = false;             // device tables uploaded

  // host tables (the reference's marshalled arguments)
  std::vector<std::vector<u64>> omega_pp;   // [k][t]   omega_{pp_k} mod q_t
  std::vector<std::vector<i64>> ru, ruinv;  // [k][pp_k*T] interleaved, tensor.h:91
  std::vector<i64> mhatinv;                 // [T]
  std::vector<i64> gcrt, ginvcrt;           // [n*T] AoS
  std::vector<u64> oddrad_inv;              // [T], 0 when not invertible
  bool has_ginvcrt = false;                 // every odd p | m invertible mod every q_t
  std::vector<u64> host_consts;             // [T][consts_per_comp] generic-path constant pool

  // device tables
  ModCtx* d_mod = nullptr;                  // [T]
  u64* d_consts = nullptr;                  // generic-path constant pool, [T][consts_per_comp]
  u64* d_consts_mont = nullptr;             // the same pool times 2^64 mod q_t (every q_t odd): Montgomery class of mixed.hip
  uint32_t* d_consts32 = nullptr;           // 32-bit copy of the pool the 32-bit classes of the vector interpreter read (class 2: the Montgomery one)
  int consts_per_comp = 0;
  StageProgram prog_crt, prog_crtinv, prog_l, prog_linv, prog_gpow, prog_gdec, prog_ginvpow, prog_ginvdec;
  // m = 2^e * odd, 5 <= e <= 15: the 2-power tensor factor is innermost (tensor.h:46-73), i.e. it
  // acts on contiguous blocks of 2^(e-1) coefficients — a batch of B*n/2^(e-1) short transforms
  // for the m = 2^k kernels; these programs hold only the odd primes' stages (the inverse one
  // without mhat^-1, which the 2-power inverse kernel folds in).
  bool pow2_part = false;
  StageProgram prog_crt_odd, prog_crtinv_odd;
  // m = 2^e * odd, e >= 2, in ONE launch of the vector interpreter: ST_POW2F/ST_POW2I tiles for the
  // 2-power factor followed / preceded by the odd primes' stages.  This is what makes the poly-mul of the
  // reference's own parameter sets (64*27, 64*81, 64*9*25, 128*7*13 ...) a fused 3-pass kernel.
  bool fused2 = false;
  StageProgram prog_crt_fused, prog_crtinv_fused;
  // classes 2 / 4 of the vector interpreter: prime powers of small totient as ONE dense stage each (plan.cpp
  // merge_stages).  Totients up to 13 (3^2) run in every kernel: prog_crt_fused / prog_crtinv_fused are merged
  // in place, prog_crt_mg / prog_crtinv_mg are the merged copies of prog_crt / prog_crtinv (empty when nothing
  // merges).  Vector lengths 18 and 20 (3^3, 5^2), 8 (3 (x) 5 as one Kronecker stage) and 9 (the DFT_9 tail of 3^e,
  // e >= 4) need the BIG kernels (mixed_impl.h): the *_big programs, empty when they would equal the others; lone
  // transforms and the fused poly-mul take them, the fused key switch at <= 12 coefficients per thread.
  StageProgram prog_crt_mg, prog_crtinv_mg, prog_crt_mg_big, prog_crtinv_mg_big, prog_crt_fused_big, prog_crtinv_fused_big;
  i64* d_gcrt = nullptr;                    // [n*T]
  i64* d_ginvcrt = nullptr;                 // [n*T]
  Pow2Tables pow2;
  bool is_pow2 = false;
  int device_id = -1;                       // HIP device the tables were uploaded to
  int mixed_cls = 0;                        // arithmetic/storage class of the vector interpreter (mixed.hip): 0-3, 4 = 2 with lazy dense stages
  // floating-point side (floatpath.hip): the CRT stage lists over C — a second
  // constant pool, (re, im) interleaved, same offsets as host_consts — and the real maps of
  // tensorGaussianDec
  std::vector<double> host_cconsts, host_rconsts;
  double *d_cconsts = nullptr, *d_rconsts = nullptr;
  StageProgram prog_gauss;
  bool float_ok = false;                    // n <= 8192 and every prime <= 13 (one LDS-resident polynomial, register vectors)
  // Nothing in a plan is written after plan_upload(): per-call workspaces (the operand copy of the
  // unfused poly-mul, the HBM ping-pong ring of polynomials too large for LDS) are stream-ordered
  // allocations made by the call that needs them, so host threads and streams can share a plan.
  bool needs_scratch = false;               // 2 * n * 8 B exceeds the LDS ping-pong budget
};

// ring-extension plan for m | m'
struct ExtPlan {
  const Plan* lo = nullptr;    // index m
  const Plan* hi = nullptr;    // index m'
  ExtTables host;
  int32_t *d_twace_powdec = nullptr, *d_ext_crt = nullptr, *d_embed_pow = nullptr,
          *d_embed_dec = nullptr, *d_embed_crt = nullptr, *d_coeffs = nullptr;
  std::vector<i64> tweak;      // [n'*T] AoS, twaceCRT's tweak vector (Extension.hs:110-125)
  i64* d_tweak = nullptr;
};

// plan.cpp
int plan_build_host(Plan& P, const std::vector<PP>& pps, const std::vector<u64>& qs,
                    const u64* omega_pp /* [npp*T] or null */, const i64* mhatinv /* [T] or null */);
int plan_upload(Plan& P);
void plan_free_device(Plan& P);

// A/B switches of the launch paths (development and tests): read ONCE from the environment
// (LOLHIP_<NAME>) into atomics; tests flip them through lolhip_debug_set, never through setenv
// (getenv racing with setenv is undefined behaviour, and plans are used from concurrent threads).
enum Switch { SW_GENERIC_SCALAR, SW_NO_FUSED2, SW_NO_POW2_PART, SW_POLYMUL_UNFUSED, SW_KEYSWITCH_UNFUSED, SW_NO_T1, SW_NO_PIPE, SW_FORCE_PIPE, SW_NO_OWN_DIAG, SW_NO_MERGE, SW_NO_LAZY, SW_NO_KRON, SW_COUNT };
bool sw(Switch which);
inline bool pow2_no_t1() { return sw(SW_NO_T1); }

}  // namespace lolhip
