// lol_amd/csrc/plan.cpp — plan construction: host tables by Lol's rules, the generic
// stage programs, the power-of-two Shoup tables, and their upload to HBM.
#include "plan.h"

#include <hip/hip_runtime_api.h>

#include <cmath>
#include <complex>
#include <cstring>

#include "lolhip.h"

namespace lolhip {

namespace {

template <typename E>
struct PoolBuilderT {
  int T;
  std::vector<std::vector<E>> pool;   // [t][...]
  explicit PoolBuilderT(int T_) : T(T_), pool((size_t)T_) {}
  int size() const { return (int)pool[0].size(); }
  // append one table per component; tab[t] all of equal length; returns offset
  int add(const std::vector<std::vector<E>>& tab) {
    int off = size();
    for (int t = 0; t < T; ++t) pool[(size_t)t].insert(pool[(size_t)t].end(), tab[(size_t)t].begin(), tab[(size_t)t].end());
    return off;
  }
  template <typename Fn> int per_comp(Fn fn, size_t len) {
    std::vector<std::vector<E>> tab((size_t)T, std::vector<E>(len));
    for (int t = 0; t < T; ++t) fn(t, tab[(size_t)t]);
    return add(tab);
  }
};
typedef PoolBuilderT<u64> PoolBuilder;

// The element rings the CRT stage program is instantiated over.  Both give, for prime power k,
// component t and direction, the table ru[i] = omega_{pp_k}^{+-i} (CPP.hs:422-442).
struct ModRing {            // Z_q, per RNS component
  typedef u64 E;
  const Plan& P; const std::vector<u64>& qs; bool ok;     // ok = the plan has a CRT basis (otherwise the tables are zeros)
  int T() const { return (int)qs.size(); }
  E ru(int k, bool inv, int t, i64 i) const { return ok ? (u64)(inv ? P.ruinv : P.ru)[(size_t)k][(size_t)(i * T() + t)] : 0; }
  E one(int t) const { return 1 % qs[(size_t)t]; }
  E sub(int t, E a, E b) const { const u64 q = qs[(size_t)t]; return (a + q - b) % q; }
  E mhatinv(int t) const { return (u64)P.mhatinv[(size_t)t]; }
};
struct CplxRing {           // C with omega_m = exp(2 pi i / m) (what Lol's Complex instance of CRTrans uses)
  typedef std::complex<double> E;
  const std::vector<PP>& pps; i64 m;
  int T() const { return 1; }
  E ru(int k, bool inv, int, i64 i) const {
    const i64 pp = ipow(pps[(size_t)k].p, pps[(size_t)k].e);
    const double ang = 2.0 * M_PI * (double)(i % pp) / (double)pp;
    return E(std::cos(ang), inv ? -std::sin(ang) : std::sin(ang));
  }
  E one(int) const { return E(1.0, 0.0); }
  E sub(int, E a, E b) const { return a - b; }
  E mhatinv(int) const { return E(1.0 / (double)value_hat(m), 0.0); }
};

struct ProgBuilder {
  std::vector<Stage> st;
  void dense(int kind, int p, int d, i64 rts, int wp_off) {
    Stage s;
    std::memset(&s, 0, sizeof(s));
    s.kind = kind; s.p = p; s.d = d; s.rts = (int32_t)rts; s.wp_off = wp_off;
    s.tw_off = -1; s.tw_mod = 1; s.tw_div = 1; s.mat_off = -1;
    st.push_back(s);
  }
  // a diagonal between two stages is folded into the stage before it
  void diag(int tw_off, i64 tw_div, i64 tw_mod) {
    if (!st.empty() && st.back().tw_off < 0) {
      st.back().tw_off = tw_off; st.back().tw_div = (int32_t)tw_div; st.back().tw_mod = (int32_t)tw_mod;
      return;
    }
    Stage s;
    std::memset(&s, 0, sizeof(s));
    s.kind = ST_DIAG; s.p = 1; s.d = 1; s.rts = 1; s.wp_off = 0; s.mat_off = -1;
    s.tw_off = tw_off; s.tw_div = (int32_t)tw_div; s.tw_mod = (int32_t)tw_mod;
    st.push_back(s);
  }
};

// emits every operation into the full program and, for prime powers other than the first,
// into the odd-only program as well (diagonals fold per program, so the two are built
// side by side rather than sliced out of one another)
struct DualBuilder {
  ProgBuilder* full; ProgBuilder* odd;
  void dense(int kind, int p, int d, i64 rts, int wp_off, int mat_off) {
    full->dense(kind, p, d, rts, wp_off); full->st.back().mat_off = mat_off;
    if (odd) { odd->dense(kind, p, d, rts, wp_off); odd->st.back().mat_off = mat_off; }
  }
  void diag(int tw_off, i64 tw_div, i64 tw_mod) {
    full->diag(tw_off, tw_div, tw_mod);
    if (odd) odd->diag(tw_off, tw_div, tw_mod);
  }
};

// ru-table accessor for component t of prime power k
struct RuView {
  const std::vector<i64>& tab; int T; int t;
  u64 operator()(i64 i) const { return (u64)tab[(size_t)(i * T + t)]; }
};

}  // namespace

static bool valid_pps(const std::vector<PP>& pps) {
  int last = 1;
  for (auto& pe : pps) {
    if (pe.p <= last || pe.e < 1 || !is_prime((u64)pe.p)) return false;
    last = pe.p;
  }
  return true;
}

// The CRT / CRT^-1 stage programs of index m = prod pps over ring R, constants appended to `pool`
// (tensor.h:76-95 over crt.cpp:459-560).  The Stage lists depend on pps only; called once per
// ring with pools that share a prefix, the offsets agree, so ONE stage list serves Z_q and C.
template <typename Ring>
static void build_crt_programs(const std::vector<PP>& pps, const Ring& R, PoolBuilderT<typename Ring::E>& pool,
                               ProgBuilder* crt, ProgBuilder* crtinv, ProgBuilder* crt_odd, ProgBuilder* crtinv_odd) {
  typedef typename Ring::E E;
  const int K = (int)pps.size();
  i64 rts = 1;
  for (int k = 0; k < K; ++k) {
    const int p = pps[(size_t)k].p, e = pps[(size_t)k].e;
    const i64 mprime = ipow(p, e - 1), phi = (p - 1) * mprime;
    for (int inv = 0; inv < 2; ++inv) {
      DualBuilder pb{inv ? crtinv : crt, (k >= 1) ? (inv ? crtinv_odd : crt_odd) : nullptr};
      auto ru = [&](int t, i64 i) { return R.ru(k, inv != 0, t, i); };
      // omega_p^j, j < p  (ru[j * p^(e-1)], crt.cpp:526 rustride = mprime)
      const int wp_off = pool.per_comp([&](int t, std::vector<E>& o) {
        for (int j = 0; j < p; ++j) o[(size_t)j] = ru(t, j * mprime);
      }, (size_t)p);
      // dense coefficient matrices, so the kernel does no index arithmetic per term:
      //   DFT_p[i][c] = w^(c*i)               (crt.cpp:226-245)
      //   CRT_p[i][c] = w^(c*(i+1))           (crt.cpp:324-345)
      //   CRT_p^-1[i][c] = w^(i*(c+1)) - w^(p-c-1)   (crt.cpp:433-456, the shift folded in)
      auto wpow = [&](int t, i64 ex) { return ru(t, (ex % p) * mprime); };
      const int mat_dft = pool.per_comp([&](int t, std::vector<E>& o) {
        for (int i = 0; i < p; ++i) for (int c = 0; c < p; ++c) o[(size_t)(i * p + c)] = wpow(t, (i64)c * i);
      }, (size_t)p * p);
      const int mat_crt = (p == 2) ? -1 : pool.per_comp([&](int t, std::vector<E>& o) {
        for (int i = 0; i < p - 1; ++i) for (int c = 0; c < p - 1; ++c)
          o[(size_t)(i * (p - 1) + c)] = inv ? R.sub(t, wpow(t, (i64)i * (c + 1)), wpow(t, p - c - 1)) : wpow(t, (i64)c * (i + 1));
      }, (size_t)(p - 1) * (p - 1));
      // crtTwiddle diagonal over the phi(pp) digit (crt.cpp:35-81)
      int ctw_off = -1;
      if (mprime > 1)
        ctw_off = pool.per_comp([&](int t, std::vector<E>& o) {
          for (i64 i0 = 0; i0 < mprime; ++i0)
            for (int i1 = 0; i1 < p - 1; ++i1)
              o[(size_t)(i0 * (p - 1) + i1)] = ru(t, digit_rev(p, e - 1, i0) * (i1 + 1));
        }, (size_t)phi);
      // dftTwiddle diagonals, one per DFT stage of DFT_{p^(e-1)} (crt.cpp:84-126, 459-516)
      const int e1 = e - 1;
      const i64 rts1 = rts * (p - 1);
      auto dft_diag = [&](int levels, i64 dim, i64 root_step) -> int {
        if (dim / p <= 1) return -1;
        return pool.per_comp([&](int t, std::vector<E>& o) {
          for (i64 c = 0; c < dim; ++c) {
            i64 i0 = c / p, i1 = c % p;
            o[(size_t)c] = (i0 == 0 || i1 == 0) ? R.one(t) : ru(t, digit_rev(p, levels - 1, i0) * i1 * root_step);
          }
        }, (size_t)dim);
      };
      // DFT_{p^(e-1)} = e-1 radix-p stages.  Stage j (j = 0 .. e-2) applies DFT_p to vectors of stride
      // rts1 p^j and is followed by the diagonal of the sub-transform of size p^(e-1-j), whose entry (i0, i1)
      // is omega_{p^e}^(digit_rev_{e-1-j}(i0) i1 p^(j+1)).  The inverse transform is the same list backwards,
      // each diagonal (built from the inverse roots) BEFORE its butterfly stage.  (What crt.cpp:459-516 walks
      // with running lts/rts/stride variables, in closed form.)
      auto dft_stage = [&](int j) {
        const i64 stride = rts1 * ipow(p, j), dim = ipow(p, e1 - j);
        const int diag = dft_diag(e1 - j, dim, ipow(p, j + 1));
        if (inv && diag >= 0) pb.diag(diag, stride, dim);
        pb.dense(ST_DFTP, p, p, stride, wp_off, mat_dft);
        if (!inv && diag >= 0) pb.diag(diag, stride, dim);
      };
      if (!inv) {
        if (p != 2) pb.dense(ST_CRTP, p, p - 1, rts, wp_off, mat_crt);
        if (ctw_off >= 0) pb.diag(ctw_off, rts, phi);
        for (int j = 0; j < e1; ++j) dft_stage(j);
      } else {
        for (int j = e1 - 1; j >= 0; --j) dft_stage(j);
        if (ctw_off >= 0) pb.diag(ctw_off, rts, phi);
        if (p != 2) pb.dense(ST_CRTPINV, p, p - 1, rts, wp_off, mat_crt);
      }
    }
    rts *= phi;
  }
  // mhat^-1 (crt.cpp:573-579)
  const int mh_off = pool.per_comp([&](int t, std::vector<E>& o) { o[0] = R.mhatinv(t); }, 1);
  crtinv->diag(mh_off, 1, 1);
}


// ---- prime powers with a small totient as ONE dense stage (class 2 of the vector interpreter) ------------
// CRT_{p^e} runs as CRT_p, the crtTwiddle diagonal and e-1 radix-p DFT stages with their diagonals between
// (crt.cpp:459-538): e round trips through LDS whose fixed cost (index arithmetic, a barrier, one Montgomery
// reduction per coefficient and stage) exceeds the arithmetic for 3^2, 5^2, 3^3.  Their product is a
// phi(p^e) x phi(p^e) matrix over Z_q — the same linear map, so the same residues bit for bit — and
// phi <= 20 fits one register vector: 20 multiply-adds and ONE reduction per coefficient instead of
// (4 + 5) multiply-adds, two diagonal products and three reductions.  Built by pushing the unit vectors
// through the staged form on the host; the merged stage replaces the run in a COPY of the program
// (the floating-point path and the scalar interpreter keep the staged list).
// KRON: the second pass — two ADJACENT tensor factors that are one dense stage each (after the first pass: a prime or a
// merged prime power) and whose Kronecker product is a vector length the kernels hold: 3 (x) 5 as ONE 8-vector stage
// (m = 15015: five round trips become four), 3 (x) 7 as a 12-vector.  Same host construction, same residues.
template <bool KRON>
static bool merge_stages(std::vector<Stage>& st, PoolBuilder& pool, const std::vector<u64>& qs, int max_phi, bool big) {      // big: lengths of the BIG kernels (8, 9) allowed
  bool any = false;
  std::vector<Stage> out;
  for (size_t a = 0; a < st.size();) {
    auto dense = [](const Stage& s) { return s.kind == ST_DFTP || s.kind == ST_CRTP || s.kind == ST_CRTPINV; };
    size_t b = a;
    if constexpr (!KRON) {
      if (dense(st[a]) && st[a].p > 2) { b = a + 1; while (b < st.size() && dense(st[b]) && st[b].p == st[a].p) ++b; }
    } else {
      // st[a], st[a+1]: whole factors (no neighbour shares their prime), consecutive (stride of the second = stride * length
      // of the first), no diagonal other than a single constant
      auto whole = [&](size_t k) { return dense(st[k]) && st[k].p > 2 && (st[k].tw_off < 0 || st[k].tw_mod == 1) &&
                                          (k == 0 || !dense(st[k - 1]) || st[k - 1].p != st[k].p) &&
                                          (k + 1 >= st.size() || !dense(st[k + 1]) || st[k + 1].p != st[k].p); };
      if (a + 1 < st.size() && whole(a) && whole(a + 1) && st[a].p != st[a + 1].p && (i64)st[a + 1].rts == (i64)st[a].rts * st[a].d) {
        const int D = st[a].d * st[a + 1].d;
        if (D == 8 || D == 12) b = a + 2;
      }
    }
    if (b < a + 2) { out.push_back(st[a]); ++a; continue; }
    // the run [a, b) is (I (x) A (x) I_R) with R the smallest stride in it and A of the largest extent d * stride / R
    auto extent = [&](size_t lo, size_t hi, i64& R, i64& phi) {
      R = st[lo].rts;
      for (size_t k = lo; k < hi; ++k) R = std::min<i64>(R, st[k].rts);
      phi = 1;
      for (size_t k = lo; k < hi; ++k) phi = std::max<i64>(phi, (i64)st[k].d * (st[k].rts / R));
    };
    const size_t a0 = a, b0 = b;              // the whole run; [a, b) becomes the part that is merged
    i64 R, phi;
    extent(a, b, R, phi);
    if constexpr (!KRON) {
      // a power of 3 too long for one vector (3^4 = 81: 54): its two OUTERMOST radix-3 stages and the diagonal between
      // them are DFT_9 on 9-vectors of their own — one stage instead of two (64 * 81 is a reference benchmark index)
      if (big && phi > max_phi && st[a].p == 3 && b - a >= 3 && max_phi >= 9) {
        if (st[b - 1].kind == ST_DFTP && st[b - 2].kind == ST_DFTP && st[b - 1].rts > st[b - 2].rts) a = b - 2;           // forward: the last two
        else if (st[a].kind == ST_DFTP && st[a + 1].kind == ST_DFTP && st[a].rts > st[a + 1].rts) b = a + 2;             // inverse: the first two
        extent(a, b, R, phi);
      }
    }
    bool inv = false;
    for (size_t k = a; k < b; ++k) if (st[k].kind == ST_CRTPINV) inv = true;
    const int p = st[a].p;
    bool ok = phi <= max_phi;
    bool carry = false;                       // the last stage's diagonal reaches beyond the merged block: it stays the merged stage's diagonal
    for (size_t k = a; k < b && ok; ++k) {
      const Stage& s = st[k];
      if (s.rts % R || phi % ((i64)s.d * (s.rts / R))) ok = false;
      if (s.tw_off >= 0 && s.tw_mod > 1) {
        bool inside = s.tw_div % R == 0;
        if (inside) { const i64 kdiv = s.tw_div / R; inside = phi % kdiv == 0 && (phi / kdiv) % s.tw_mod == 0; }
        if (!inside) { if (k + 1 == b) carry = true; else ok = false; }
      }
    }
    if (!ok) {
      if (KRON) { out.push_back(st[a0]); a = a0 + 1; } else { for (size_t k = a0; k < b0; ++k) out.push_back(st[k]); a = b0; }
      continue;
    }
    for (size_t k = a0; k < a; ++k) out.push_back(st[k]);
    const int mat = pool.per_comp([&](int t, std::vector<u64>& o) {
      const u64 q = qs[(size_t)t];
      const std::vector<u64>& tab = pool.pool[(size_t)t];
      std::vector<u64> y((size_t)phi), v, w;
      for (i64 c = 0; c < phi; ++c) {
        std::fill(y.begin(), y.end(), 0); y[(size_t)c] = 1 % q;
        for (size_t k = a; k < b; ++k) {
          const Stage& s = st[k];
          const i64 d = s.d, stride = s.rts / R;
          v.assign((size_t)d, 0); w.assign((size_t)d, 0);
          for (i64 blk = 0; blk < phi / (d * stride); ++blk)
            for (i64 r = 0; r < stride; ++r) {
              const i64 x0 = blk * d * stride + r;
              for (i64 i = 0; i < d; ++i) v[(size_t)i] = y[(size_t)(x0 + i * stride)];
              for (i64 i = 0; i < d; ++i) {
                u64 acc = 0;
                for (i64 j = 0; j < d; ++j) acc = (acc + mulmod(tab[(size_t)(s.mat_off + i * d + j)] % q, v[(size_t)j], q)) % q;
                if (s.tw_off >= 0 && !(carry && k + 1 == b)) {
                  const i64 xl = x0 + i * stride;
                  const i64 idx = s.tw_mod > 1 ? (xl / (s.tw_div / R)) % s.tw_mod : 0;
                  acc = mulmod(acc, tab[(size_t)(s.tw_off + idx)] % q, q);
                }
                w[(size_t)i] = acc;
              }
              for (i64 i = 0; i < d; ++i) y[(size_t)(x0 + i * stride)] = w[(size_t)i];
            }
        }
        for (i64 i = 0; i < phi; ++i) o[(size_t)(i * phi + c)] = y[(size_t)i];
      }
    }, (size_t)(phi * phi));
    Stage m;
    std::memset(&m, 0, sizeof(m));
    m.kind = inv ? ST_CRTPINV : ST_CRTP; m.p = p; m.d = (int32_t)phi; m.rts = (int32_t)R; m.wp_off = st[a].wp_off;
    m.tw_off = -1; m.tw_mod = 1; m.tw_div = 1; m.mat_off = mat;
    if (carry) { m.tw_off = st[b - 1].tw_off; m.tw_mod = st[b - 1].tw_mod; m.tw_div = st[b - 1].tw_div; }
    out.push_back(m);
    for (size_t k = b; k < b0; ++k) out.push_back(st[k]);
    any = true;
    a = b0;
  }
  if (any) st.swap(out);
  return any;
}

int plan_build_host(Plan& P, const std::vector<PP>& pps, const std::vector<u64>& qs,
                    const u64* omega_pp_in, const i64* mhatinv_in) {
  if (!valid_pps(pps) || qs.empty() || qs.size() > 64) return LOLHIP_ERR_INVALID;
  for (u64 q : qs) if (q < 2 || q >= (1ull << 62)) return LOLHIP_ERR_MODULUS;
  P.pps = pps;
  P.T = (int)qs.size();
  P.qs = qs;
  P.m = value_pps(pps);
  P.n = totient_pps(pps);
  if (P.n > (1 << 24)) return LOLHIP_ERR_INVALID;
  const int T = P.T, K = (int)pps.size();

  // ---- roots of unity ------------------------------------------------------
  P.has_crt = true;
  P.omega_pp.assign((size_t)K, std::vector<u64>((size_t)T, 0));
  for (int t = 0; t < T; ++t) {
    if (omega_pp_in) {
      for (int k = 0; k < K; ++k) P.omega_pp[(size_t)k][(size_t)t] = omega_pp_in[k * T + t] % qs[(size_t)t];
    } else {
      u64 w = principal_root((u64)P.m, qs[(size_t)t]);
      if (w == 0) { P.has_crt = false; break; }
      for (int k = 0; k < K; ++k) {
        i64 pp = ipow(pps[(size_t)k].p, pps[(size_t)k].e);
        P.omega_pp[(size_t)k][(size_t)t] = powmod(w, (u64)(P.m / pp), qs[(size_t)t]);
      }
    }
  }
  // a caller-supplied root must really have order pp
  if (P.has_crt && omega_pp_in) {
    for (int t = 0; t < T && P.has_crt; ++t)
      for (int k = 0; k < K; ++k) {
        u64 q = qs[(size_t)t], w = P.omega_pp[(size_t)k][(size_t)t];
        i64 pp = ipow(pps[(size_t)k].p, pps[(size_t)k].e);
        if (powmod(w, (u64)pp, q) != 1 % q || powmod(w, (u64)(pp / pps[(size_t)k].p), q) == 1 % q) return LOLHIP_ERR_ROOT;
      }
  }

  P.ru.clear(); P.ruinv.clear(); P.mhatinv.assign((size_t)T, 0);
  P.gcrt.clear(); P.ginvcrt.clear();
  if (P.has_crt) {
    P.ru.resize((size_t)K); P.ruinv.resize((size_t)K);
    for (int k = 0; k < K; ++k) {
      i64 pp = ipow(pps[(size_t)k].p, pps[(size_t)k].e);
      P.ru[(size_t)k].assign((size_t)(pp * T), 0);
      P.ruinv[(size_t)k].assign((size_t)(pp * T), 0);
      for (int t = 0; t < T; ++t) {
        u64 q = qs[(size_t)t], w = P.omega_pp[(size_t)k][(size_t)t], wi = invmod(w, q);
        u64 x = 1 % q, xi = 1 % q;
        for (i64 i = 0; i < pp; ++i) {
          P.ru[(size_t)k][(size_t)(i * T + t)] = (i64)x;
          P.ruinv[(size_t)k][(size_t)(i * T + t)] = (i64)xi;
          x = mulmod(x, w, q); xi = mulmod(xi, wi, q);
        }
      }
    }
    for (int t = 0; t < T; ++t) {
      u64 q = qs[(size_t)t];
      u64 mh = mhatinv_in ? ((u64)(mhatinv_in[t] % (i64)q + (i64)q)) % q : invmod((u64)value_hat(P.m) % q, q);
      if (mh == 0 && q != 1) return LOLHIP_ERR_MODULUS;
      P.mhatinv[(size_t)t] = (i64)mh;
    }
    // g vectors (AoS)
    P.gcrt.assign((size_t)(P.n * T), 0); P.ginvcrt.assign((size_t)(P.n * T), 0);
    P.has_ginvcrt = true;
    for (int t = 0; t < T; ++t) {
      u64 q = qs[(size_t)t];
      std::vector<u64> wp((size_t)K);
      bool ginv_ok = true;
      for (int k = 0; k < K; ++k) {
        i64 pp = ipow(pps[(size_t)k].p, pps[(size_t)k].e);
        wp[(size_t)k] = powmod(P.omega_pp[(size_t)k][(size_t)t], (u64)(pp / pps[(size_t)k].p), q);
        if (pps[(size_t)k].p != 2 && invmod((u64)pps[(size_t)k].p % q, q) == 0) ginv_ok = false;
      }
      std::vector<u64> g = g_crt(pps, wp, q, false);
      for (i64 i = 0; i < P.n; ++i) P.gcrt[(size_t)(i * T + t)] = (i64)g[(size_t)i];
      if (ginv_ok) {
        std::vector<u64> gi = g_crt(pps, wp, q, true);
        for (i64 i = 0; i < P.n; ++i) P.ginvcrt[(size_t)(i * T + t)] = (i64)gi[(size_t)i];
      } else {
        P.has_ginvcrt = false;
      }
    }
  }
  P.oddrad_inv.assign((size_t)T, 0);
  for (int t = 0; t < T; ++t) P.oddrad_inv[(size_t)t] = invmod(odd_rad(pps) % qs[(size_t)t], qs[(size_t)t]);

  // ---- generic stage programs + constant pool ---------------------------------
  auto magic40 = [](i64 v) -> uint64_t { return (uint64_t)(((unsigned __int128)1 << 40) / (uint64_t)(v > 0 ? v : 1)) + 1; };
  // class 3 of the vector interpreter (as plan_upload decides it): the dense stages of odd primes take the
  // even/odd form (mixed_impl.h apply_eo) — eo_off[p][direction] is filled below, before any program is finished
  bool cls3 = true;
  { bool fits32 = true;
    for (u64 q : qs) { if (q >= ((u64)1 << 32)) fits32 = false; if (!(q & 1) || q >= ((u64)1 << 61)) cls3 = false; }
    if (fits32) cls3 = false; }
  int eo_off[16][2];
  for (auto& r : eo_off) r[0] = r[1] = 0;
  auto finish = [&](std::vector<Stage>& st, bool inverse = false) {
    for (size_t si = 0; si < st.size(); ++si) {
      Stage& s = st[si];
      const bool dense_odd = (s.kind == ST_DFTP || s.kind == ST_CRTP || s.kind == ST_CRTPINV) && s.p > 2 && s.p <= 13 && s.d >= 3 && (s.d == s.p || s.d == s.p - 1);
      s.pad[0] = (cls3 && dense_odd) ? eo_off[s.p][inverse ? 1 : 0] : 0;
      // class 4 (lazy dense stages): the inverse tile that follows a dense stage or a diagonal canonicalises what it loads
      s.pad[1] = (s.kind == ST_POW2I && si > 0 && st[si - 1].kind != ST_POW2I) ? 1 : 0;
      s.m_rts = magic40(s.rts); s.m_d = magic40(s.d); s.m_twdiv = magic40(s.tw_div); s.m_twmod = magic40(s.tw_mod);
      const bool dense = s.kind == ST_DFTP || s.kind == ST_CRTP || s.kind == ST_CRTPINV;
      s.tw_per = (!sw(SW_NO_OWN_DIAG) && dense && s.tw_off >= 0 && s.tw_mod > 1 && s.tw_div == s.rts && s.tw_mod % s.d == 0) ? s.tw_mod / s.d : 0;
      s.m_twper = magic40(s.tw_per > 0 ? s.tw_per : 1);
    }
  };
  PoolBuilder pool(T);
  {  // slot 0: the constant 1 (keeps offsets non-negative and gives an identity diagonal)
    std::vector<std::vector<u64>> one((size_t)T, std::vector<u64>(1));
    for (int t = 0; t < T; ++t) one[(size_t)t][0] = 1 % qs[(size_t)t];
    pool.add(one);
  }
  auto per_comp = [&](auto fn, size_t len) {
    std::vector<std::vector<u64>> tab((size_t)T, std::vector<u64>(len));
    for (int t = 0; t < T; ++t) fn(t, tab[(size_t)t]);
    return pool.add(tab);
  };

  // class 2 of the vector interpreter (as plan_upload decides it), and the longest dense vector its one
  // 64-bit accumulator and 32-bit Montgomery step (mixed_impl.h redc64) take: D q^2 < 2^64 and D q <= 2^34
  bool cls2 = true;
  int max_phi = 20;
  for (u64 q : qs) {
    if (q >= ((u64)1 << 32) || !(q & 1) || (unsigned __int128)13 * (q - 1) * (q - 1) >= ((unsigned __int128)1 << 64)) cls2 = false;
    else if ((unsigned __int128)20 * q * q >= ((unsigned __int128)1 << 64) || 20 * q > ((u64)1 << 34)) max_phi = 13;
  }
  ProgBuilder crt, crtinv, crt_odd, crtinv_odd;
  const bool split2 = K >= 2 && pps[0].p == 2;   // odd-only programs skip the first (2-power) factor
  {
    ModRing ring{P, qs, P.has_crt};
    build_crt_programs(pps, ring, pool, &crt, &crtinv, split2 ? &crt_odd : nullptr, split2 ? &crtinv_odd : nullptr);
  }
  const int orad_off = per_comp([&](int t, std::vector<u64>& o) { o[0] = P.oddrad_inv[(size_t)t]; }, 1);
  // even/odd tables of the odd primes (class 3): for r, c = 1 .. h = (p-1)/2 and w = omega_p (inverse: omega_p^-1)
  //   cos[r][c] = (w^(rc) + w^(-rc)) / 2,   sin[r][c] = (w^(rc) - w^(-rc)) / 2      (q odd), cos first, row-major
  if (cls3 && P.has_crt) {
    for (int k = 0; k < K; ++k) {
      const int p = pps[(size_t)k].p;
      if (p < 3 || p > 13) continue;
      const i64 mprime = ipow(p, pps[(size_t)k].e - 1);
      const int h = (p - 1) / 2;
      for (int inv = 0; inv < 2; ++inv)
        eo_off[p][inv] = per_comp([&](int t, std::vector<u64>& o) {
          const u64 q = qs[(size_t)t], half = (q + 1) >> 1;
          auto w = [&](i64 ex) { return (u64)(inv ? P.ruinv : P.ru)[(size_t)k][(size_t)((((ex % p) + p) % p) * mprime * T + t)]; };
          for (int r = 1; r <= h; ++r)
            for (int c = 1; c <= h; ++c) {
              const u64 a = w((i64)r * c), b = w(-(i64)r * c);
              o[(size_t)((r - 1) * h + (c - 1))] = mulmod((a + b) % q, half, q);
              o[(size_t)(h * h + (r - 1) * h + (c - 1))] = mulmod((a + q - b) % q, half, q);
            }
        }, (size_t)(2 * h * h));
    }
  }

  auto prime_prog = [&](int kind, bool scale) {
    ProgBuilder pb;
    i64 rts = 1;
    for (int k = 0; k < K; ++k) {
      const int p = pps[(size_t)k].p, e = pps[(size_t)k].e;
      if (p != 2) pb.dense(kind, p, p - 1, rts, 0);
      rts *= totient_pp(p, e);
    }
    if (scale) pb.diag(orad_off, 1, 1);
    finish(pb.st);
    return pb.st;
  };
  finish(crt.st, false); finish(crtinv.st, true);
  P.prog_crt.stages = crt.st;
  P.prog_crtinv.stages = crtinv.st;
  P.prog_l.stages = prime_prog(ST_L, false);
  P.prog_linv.stages = prime_prog(ST_LINV, false);
  P.prog_gpow.stages = prime_prog(ST_GPOW, false);
  P.prog_gdec.stages = prime_prog(ST_GDEC, false);
  P.prog_ginvpow.stages = prime_prog(ST_GINVPOW, true);
  P.prog_ginvdec.stages = prime_prog(ST_GINVDEC, true);

  // ---- m = 2^e * odd in one launch: 2-power tiles + the odd primes' stages ----------------------
  std::vector<Stage> fused_f, fused_i;
  P.fused2 = false;
  if (P.has_crt && split2 && pps[0].e >= 2) {
    const int L = pps[0].e - 1;
    const i64 h = (i64)1 << L;
    // entry 2^(s-1) + i = omega_{2^e}^((h / 2^s) (2i+1)), s = 1..L, i < 2^(s-1); inverse table: the inverses,
    // its level-1 entry times mhat^-1 (the X output of that level is scaled through mat_off)
    auto table = [&](bool inverse) {
      return per_comp([&](int t, std::vector<u64>& o) {
        const u64 q = qs[(size_t)t];
        o[0] = 1 % q;
        for (int s = 1; s <= L; ++s) {
          const i64 half = (i64)1 << (s - 1), step = h >> s;
          for (i64 i = 0; i < half; ++i) {
            const i64 ex = step * (2 * i + 1);
            u64 w = (u64)(inverse ? P.ruinv : P.ru)[0][(size_t)(ex * T + t)];
            if (inverse && s == 1) w = mulmod(w, (u64)P.mhatinv[(size_t)t], q);
            o[(size_t)(half + i)] = w;
          }
        }
      }, (size_t)h);
    };
    const int twf = table(false), twi = table(true);
    const int mh = per_comp([&](int t, std::vector<u64>& o) { o[0] = (u64)P.mhatinv[(size_t)t]; }, 1);
    auto tile = [&](int kind, int s_lo, int k, int tw_off, int mat_off) {
      Stage s;
      std::memset(&s, 0, sizeof(s));
      s.kind = kind; s.p = s_lo; s.d = k; s.rts = (int32_t)1 << (s_lo - 1); s.wp_off = 0;
      s.tw_off = tw_off; s.tw_mod = (int32_t)h; s.tw_div = 1; s.mat_off = mat_off;
      return s;
    };
    // level groups, bottom up: 4 levels per tile, the remainder on top.  (Five levels in the first tile of the
    // 32-bit class — 32 residues, twiddles as scalar operands — were built: the monolithic interpreter kernel then
    // spills 27 VGPRs at its 80-register budget; in separate instantiations at 128 registers it does not spill and the
    // lone crt of 64*9*25 is 14 % SLOWER — 120 work items of 32 coefficients idle half the workgroup and serialise
    // each stage; profiles/r03_ab_tile5_negative.txt.)
    std::vector<std::pair<int, int>> groups;
    for (int s = 1; s <= L; s += 4) groups.push_back({s, std::min(4, L - s + 1)});
    for (auto& g : groups) fused_f.push_back(tile(ST_POW2F, g.first, g.second, twf, -1));
    fused_f.insert(fused_f.end(), crt_odd.st.begin(), crt_odd.st.end());
    fused_i = crtinv_odd.st;
    for (auto it = groups.rbegin(); it != groups.rend(); ++it)
      fused_i.push_back(tile(ST_POW2I, it->first, it->second, twi, it->first == 1 ? mh : -1));
    finish(fused_f, false); finish(fused_i, true);
    P.fused2 = true;
  }
  // class 2: small prime powers as one dense stage each (merge_stages; plan.h says which program runs where)
  for (StageProgram* sp : {&P.prog_crt_mg, &P.prog_crtinv_mg, &P.prog_crt_mg_big, &P.prog_crtinv_mg_big, &P.prog_crt_fused_big, &P.prog_crtinv_fused_big})
    sp->stages.clear();
  if (cls2 && P.has_crt && !sw(SW_NO_MERGE)) {
    // "big": a dense vector length only the BIG kernels hold — 18, 20 (merged 3^3, 5^2), 8 (3 (x) 5), 9 (DFT_9 of 3^e, e >= 4)
    auto has_big = [](const std::vector<Stage>& st) { for (const Stage& s : st) if (s.kind != ST_POW2F && s.kind != ST_POW2I && (s.d > 13 || s.d == 8 || s.d == 9)) return true; return false; };
    // both directions of a pair get the same treatment (the fused poly-mul launches them together)
    auto merged_pair = [&](const std::vector<Stage>& f0, const std::vector<Stage>& i0, int phi, bool kron, std::vector<Stage>& f, std::vector<Stage>& i) {
      f = f0; i = i0;
      bool mf = merge_stages<false>(f, pool, qs, phi, kron), mi = merge_stages<false>(i, pool, qs, phi, kron);
      if (mf != mi) { f = f0; i = i0; return false; }
      if (kron && !sw(SW_NO_KRON)) {
        std::vector<Stage> f1 = f, i1 = i;
        const bool kf = merge_stages<true>(f1, pool, qs, 13, true), ki = merge_stages<true>(i1, pool, qs, 13, true);
        if (kf && ki) { f = f1; i = i1; mf = true; }
      }
      if (mf) { finish(f, false); finish(i, true); }
      return mf;
    };
    std::vector<Stage> f, i;
    if (!fused_f.empty()) {
      if (merged_pair(fused_f, fused_i, max_phi, true, f, i) && (has_big(f) || has_big(i))) { P.prog_crt_fused_big.stages = f; P.prog_crtinv_fused_big.stages = i; }
      if (merged_pair(fused_f, fused_i, 13, false, f, i)) { fused_f = f; fused_i = i; }
    }
    if (merged_pair(crt.st, crtinv.st, max_phi, true, f, i) && (has_big(f) || has_big(i))) { P.prog_crt_mg_big.stages = f; P.prog_crtinv_mg_big.stages = i; }
    if (merged_pair(crt.st, crtinv.st, 13, false, f, i)) { P.prog_crt_mg.stages = f; P.prog_crtinv_mg.stages = i; }
  }
  P.prog_crt_fused.stages = fused_f;
  P.prog_crtinv_fused.stages = fused_i;

  P.consts_per_comp = pool.size();
  P.host_consts.clear();
  for (int t = 0; t < T; ++t) P.host_consts.insert(P.host_consts.end(), pool.pool[(size_t)t].begin(), pool.pool[(size_t)t].end());

  // ---- floating-point side -----------------------------------------------
  // tensorCRTC / tensorCRTInvC (crt.cpp:583-598) are the SAME templates as the Z_q transforms
  // instantiated at Complex (ppcrt/ppcrtinv over tensorFuserCRT): the stage lists above are
  // reused as they are with a second constant pool over C whose offsets coincide.
  {
    PoolBuilderT<std::complex<double>> cpool(1);
    cpool.add({{std::complex<double>(1.0, 0.0)}});
    ProgBuilder d0, d1, d2, d3;
    CplxRing ring{pps, P.m};
    build_crt_programs(pps, ring, cpool, &d0, &d1, split2 ? &d2 : nullptr, split2 ? &d3 : nullptr);
    P.host_cconsts.clear();
    for (const auto& z : cpool.pool[0]) { P.host_cconsts.push_back(z.real()); P.host_cconsts.push_back(z.imag()); }
  }
  // tensorGaussianDec (random.cpp:19-64): per odd prime p the real (p-1) x (p-1) map
  //   out[row] = (sum_{col=1}^{p-1} 2 c(row*col mod p) y[col-1]) / sqrt 2,
  //   c(k) = Re omega_p^k for col <= p/2, Im omega_p^k for col > p/2, omega_p = exp(2 pi i / p)
  // on every (p-1)-vector of axis k (lts * p^(e-1) blocks, stride rts: ppD, random.cpp:52-58).
  {
    ProgBuilder pb;
    P.host_rconsts.assign(1, 1.0);
    i64 rts = 1;
    for (int k = 0; k < K; ++k) {
      const int p = pps[(size_t)k].p, e = pps[(size_t)k].e;
      if (p != 2) {
        const int off = (int)P.host_rconsts.size();
        for (int row = 0; row < p - 1; ++row)
          for (int col = 1; col <= p - 1; ++col) {
            const double ang = 2.0 * M_PI * (double)((row * col) % p) / (double)p;
            P.host_rconsts.push_back(2.0 * (col <= (p >> 1) ? std::cos(ang) : std::sin(ang)));
          }
        pb.dense(ST_GAUSS, p, p - 1, rts, 0);
        pb.st.back().mat_off = off;
      }
      rts *= totient_pp(p, e);
    }
    finish(pb.st);
    P.prog_gauss.stages = pb.st;
  }
  P.float_ok = P.n <= 8192;
  for (const auto& pe : pps) if (pe.p > 13) P.float_ok = false;

  P.is_pow2 = P.has_crt && K == 1 && pps[0].p == 2 && pps[0].e >= 5 && pps[0].e <= 15;
  P.pow2_part = P.has_crt && K >= 2 && pps[0].p == 2 && pps[0].e >= 5 && pps[0].e <= 15;
  P.pow2.L = (P.is_pow2 || P.pow2_part) ? pps[0].e - 1 : 0;
  if (P.pow2_part) {        // mhat^-1 is not in crtinv_odd: the 2-power inverse kernel folds it in
    finish(crt_odd.st, false); finish(crtinv_odd.st, true);
    P.prog_crt_odd.stages = crt_odd.st;
    P.prog_crtinv_odd.stages = crtinv_odd.st;
  }
  return LOLHIP_OK;
}

// ---- device upload -------------------------------------------------------------

#define HIPCK(x) do { if ((x) != hipSuccess) return LOLHIP_ERR_HIP; } while (0)

template <typename Tp>
static int upload(Tp** dptr, const std::vector<Tp>& h) {
  *dptr = nullptr;
  if (h.empty()) return LOLHIP_OK;
  HIPCK(hipMalloc((void**)dptr, h.size() * sizeof(Tp)));
  HIPCK(hipMemcpy(*dptr, h.data(), h.size() * sizeof(Tp), hipMemcpyHostToDevice));
  return LOLHIP_OK;
}

static int upload_prog(StageProgram& sp) {
  sp.nstages = (int)sp.stages.size();
  sp.big = false;
  for (const Stage& s : sp.stages) if (s.kind != ST_POW2F && s.kind != ST_POW2I && (s.d > 13 || s.d == 8 || s.d == 9)) sp.big = true;
  return upload(&sp.d_stages, sp.stages);
}

int plan_upload(Plan& P) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count == 0) return LOLHIP_ERR_NO_DEVICE;
  const int T = P.T;
  std::vector<ModCtx> mods;
  for (int t = 0; t < T; ++t) mods.push_back(make_modctx(P.qs[(size_t)t]));
  int rc;
  if ((rc = upload(&P.d_mod, mods))) return rc;
  if ((rc = upload(&P.d_consts, P.host_consts))) return rc;
  StageProgram* progs[] = {&P.prog_crt, &P.prog_crtinv, &P.prog_l, &P.prog_linv, &P.prog_gpow, &P.prog_gdec, &P.prog_ginvpow, &P.prog_ginvdec,
                           &P.prog_crt_odd, &P.prog_crtinv_odd, &P.prog_gauss, &P.prog_crt_fused, &P.prog_crtinv_fused,
                           &P.prog_crt_mg, &P.prog_crtinv_mg, &P.prog_crt_mg_big, &P.prog_crtinv_mg_big, &P.prog_crt_fused_big, &P.prog_crtinv_fused_big};
  for (auto* sp : progs) if ((rc = upload_prog(*sp))) return rc;
  if ((rc = upload(&P.d_cconsts, P.host_cconsts))) return rc;
  if ((rc = upload(&P.d_rconsts, P.host_rconsts))) return rc;
  if ((rc = upload(&P.d_gcrt, P.gcrt))) return rc;
  if ((rc = upload(&P.d_ginvcrt, P.ginvcrt))) return rc;

  if (P.is_pow2 || P.pow2_part) {
    const int L = P.pow2.L;
    const i64 n = (i64)1 << L;          // length of the 2-power factor (= P.n for m = 2^k)
    // per component 8 words: Shoup pairs of S = mhat^-1, of the level-1 inverse twiddle times S, and of the
    // same two times 2^64 mod q (the 64-bit fused poly-mul's pointwise product is a Montgomery reduction,
    // which leaves a factor 2^-64 that the last inverse level takes back for free)
    std::vector<u64> fwd((size_t)(T * n * 2), 0), inv((size_t)(T * n * 2), 0), sc((size_t)(T * 8), 0);
    for (int t = 0; t < T; ++t) {
      const u64 q = P.qs[(size_t)t];
      const u64 S = (u64)P.mhatinv[(size_t)t];
      {
        const u64 w1 = mulmod((u64)P.ruinv[0][(size_t)((n >> 1) * T + t)], S, q);      // psi_2^-1 * S: the level-1 entry below
        const u64 R = (u64)((((unsigned __int128)1) << 64) % q);
        const u64 vals[4] = {S, w1, mulmod(S, R, q), mulmod(w1, R, q)};
        for (int k = 0; k < 4; ++k) {
          ShoupW ss = make_shoup(vals[k], q);
          sc[(size_t)(t * 8 + 2 * k)] = ss.w; sc[(size_t)(t * 8 + 2 * k + 1)] = ss.wp;
        }
      }
      for (int s = 1; s <= L; ++s) {
        const i64 N = (i64)1 << s, half = N >> 1, step = n / N;
        for (i64 i = 0; i < half; ++i) {
          const i64 ex = step * (2 * i + 1);                  // < 2n = m
          u64 w = (u64)P.ru[0][(size_t)(ex * T + t)];
          u64 wi = (u64)P.ruinv[0][(size_t)(ex * T + t)];
          if (s == 1) wi = mulmod(wi, S, q);                   // level 1 of the inverse carries mhat^-1
          ShoupW a = make_shoup(w, q), b = make_shoup(wi, q);
          const size_t o = ((size_t)t * n + (size_t)(half + i)) * 2;
          fwd[o] = a.w; fwd[o + 1] = a.wp;
          inv[o] = b.w; inv[o + 1] = b.wp;
        }
      }
    }
    if ((rc = upload(&P.pow2.d_tw_fwd, fwd))) return rc;
    if ((rc = upload(&P.pow2.d_tw_inv, inv))) return rc;
    if ((rc = upload(&P.pow2.d_scale, sc))) return rc;
    P.pow2.arith32 = 4;
    for (u64 q : P.qs) { if (q >= (1ull << 27)) P.pow2.arith32 = 2; }
    for (u64 q : P.qs) { if (q >= (1ull << 30)) P.pow2.arith32 = 3; }
    for (u64 q : P.qs) { if (q >= (1ull << 31) || !(q & 1)) P.pow2.arith32 = 0; }     // the 32-bit classes' pointwise product is a Montgomery step: odd q
    if (P.pow2.arith32) {   // 32-bit Shoup pairs: wp = floor(w * 2^32 / q)
      std::vector<uint32_t> f32(fwd.size()), i32(inv.size()), s32(sc.size());
      for (int t = 0; t < T; ++t) {
        const u64 q = P.qs[(size_t)t];
        for (i64 i = 0; i < n; ++i) {
          const size_t o = ((size_t)t * n + (size_t)i) * 2;
          f32[o] = (uint32_t)fwd[o]; f32[o + 1] = (uint32_t)((fwd[o] << 32) / q);
          i32[o] = (uint32_t)inv[o]; i32[o + 1] = (uint32_t)((inv[o] << 32) / q);
        }
        for (int k = 0; k < 4; ++k) {      // (the 2^64-scaled pairs are not used by the 32-bit classes)
          s32[(size_t)t * 8 + 2 * k] = (uint32_t)sc[(size_t)t * 8 + 2 * k];
          s32[(size_t)t * 8 + 2 * k + 1] = (uint32_t)((sc[(size_t)t * 8 + 2 * k] << 32) / q);
        }
      }
      if ((rc = upload(&P.pow2.d_tw_fwd32, f32))) return rc;
      if ((rc = upload(&P.pow2.d_tw_inv32, i32))) return rc;
      if ((rc = upload(&P.pow2.d_scale32, s32))) return rc;
    }
  }
  // polynomials too large for the LDS ping-pong run the generic path out of an HBM scratch ring
  // (allocated per call, stream-ordered: run_prog in capi.cpp)
  P.needs_scratch = 2 * (size_t)P.n * sizeof(u64) > 152 * 1024;
  // class of the vector interpreter: 32-bit residues when every q < 2^32; a dot product of up to 13
  // terms in ONE 64-bit accumulator when 13 (q-1)^2 < 2^64
  // class of the vector interpreter (mixed.hip):
  //   2: every q odd with 13 (q-1)^2 < 2^64: 32-bit residues, one 64-bit accumulator per dot
  //      product, constants pre-scaled by 2^32, one 32-bit Montgomery reduction per output;
  //   1: every q < 2^32 otherwise (128-bit accumulators, exact division step);
  //   3: 64-bit residues, every q odd and below 2^61: constants pre-scaled by 2^64, one 64-bit
  //      Montgomery reduction per output;   0: anything else (exact two-step division).
  bool fits32 = true, acc64 = true, odd = true, below61 = true;
  for (u64 q : P.qs) {
    if (q >= ((u64)1 << 32)) fits32 = false;
    if ((unsigned __int128)13 * (q - 1) * (q - 1) >= ((unsigned __int128)1 << 64)) acc64 = false;
    if (!(q & 1)) odd = false;
    if (q >= ((u64)1 << 61)) below61 = false;
  }
  P.mixed_cls = fits32 ? ((acc64 && odd) ? 2 : 1) : ((odd && below61) ? 3 : 0);
  //   4: class 2 with every q below 2^27: lazy dense stages (values in [0,2q) between them: a 3-instruction Montgomery step)
  if (P.mixed_cls == 2 && !sw(SW_NO_LAZY)) {
    bool below27 = true;
    for (u64 q : P.qs) if (q >= ((u64)1 << 27)) below27 = false;
    if (below27) P.mixed_cls = 4;
  }
  if (P.mixed_cls == 2 || P.mixed_cls == 4 || P.mixed_cls == 3) {
    const int sh = P.mixed_cls == 3 ? 64 : 32;
    std::vector<u64> mont(P.host_consts.size());
    for (int t = 0; t < T; ++t) {
      const u64 q = P.qs[(size_t)t];
      const u64 r = (u64)((((unsigned __int128)1) << sh) % q);
      for (int i = 0; i < P.consts_per_comp; ++i) {
        const size_t o = (size_t)t * P.consts_per_comp + i;
        mont[o] = mulmod(P.host_consts[o] % q, r, q);
      }
    }
    if ((rc = upload(&P.d_consts_mont, mont))) return rc;
    if (P.mixed_cls == 2 || P.mixed_cls == 4) {
      std::vector<uint32_t> c32(mont.begin(), mont.end());           // every entry is a residue below q < 2^32
      if ((rc = upload(&P.d_consts32, c32))) return rc;
    }
  }
  if (P.mixed_cls == 1) {
    std::vector<uint32_t> c32(P.host_consts.size());
    for (int t = 0; t < T; ++t)
      for (int i = 0; i < P.consts_per_comp; ++i) {
        const size_t o = (size_t)t * P.consts_per_comp + i;
        c32[o] = (uint32_t)(P.host_consts[o] % P.qs[(size_t)t]);
      }
    if ((rc = upload(&P.d_consts32, c32))) return rc;
  }
  HIPCK(hipGetDevice(&P.device_id));
  P.device = true;
  return LOLHIP_OK;
}

void plan_free_device(Plan& P) {
  auto fr = [](void* p) { if (p) (void)hipFree(p); };
  fr(P.d_mod); fr(P.d_consts); fr(P.d_consts_mont); fr(P.d_consts32); P.d_consts32 = nullptr; fr(P.d_gcrt); fr(P.d_ginvcrt); fr(P.d_cconsts); fr(P.d_rconsts);
  P.d_cconsts = nullptr; P.d_rconsts = nullptr; P.d_consts_mont = nullptr;
  fr(P.pow2.d_tw_fwd); fr(P.pow2.d_tw_inv); fr(P.pow2.d_scale);
  fr(P.pow2.d_tw_fwd32); fr(P.pow2.d_tw_inv32); fr(P.pow2.d_scale32);
  StageProgram* progs[] = {&P.prog_crt, &P.prog_crtinv, &P.prog_l, &P.prog_linv, &P.prog_gpow, &P.prog_gdec, &P.prog_ginvpow, &P.prog_ginvdec,
                           &P.prog_crt_odd, &P.prog_crtinv_odd, &P.prog_gauss, &P.prog_crt_fused, &P.prog_crtinv_fused,
                           &P.prog_crt_mg, &P.prog_crtinv_mg, &P.prog_crt_mg_big, &P.prog_crtinv_mg_big, &P.prog_crt_fused_big, &P.prog_crtinv_fused_big};
  for (auto* sp : progs) { fr(sp->d_stages); sp->d_stages = nullptr; }
  P.d_mod = nullptr; P.d_consts = nullptr; P.d_gcrt = nullptr; P.d_ginvcrt = nullptr;
  P.pow2 = Pow2Tables();
  P.device = false;
}

}  // namespace lolhip
