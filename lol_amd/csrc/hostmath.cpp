// lol_amd/csrc/hostmath.cpp — see hostmath.h for the reference map.
#include "hostmath.h"

#include <algorithm>
#include <numeric>
#include <set>

namespace lolhip {

u64 powmod(u64 b, u64 e, u64 q) {
  u64 r = 1 % q;
  b %= q;
  while (e) {
    if (e & 1) r = mulmod(r, b, q);
    b = mulmod(b, b, q);
    e >>= 1;
  }
  return r;
}

u64 invmod(u64 b, u64 q) {
  // extended Euclid on signed 128-bit to stay exact for q < 2^63
  __int128 r0 = q, r1 = b % q, t0 = 0, t1 = 1;
  while (r1 != 0) {
    __int128 k = r0 / r1, tmp = r0 - k * r1;
    r0 = r1; r1 = tmp;
    tmp = t0 - k * t1; t0 = t1; t1 = tmp;
  }
  if (r0 != 1) return 0;
  if (t0 < 0) t0 += q;
  return (u64)t0;
}

bool is_prime(u64 n) {
  if (n < 2) return false;
  static const u64 bases[] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};  // exact below 3.3e24
  for (u64 p : bases) if (n % p == 0) return n == p;
  u64 d = n - 1; int s = 0;
  while ((d & 1) == 0) { d >>= 1; ++s; }
  for (u64 a : bases) {
    u64 x = powmod(a, d, n);
    if (x == 1 || x == n - 1) continue;
    bool comp = true;
    for (int i = 1; i < s; ++i) {
      x = mulmod(x, x, n);
      if (x == n - 1) { comp = false; break; }
    }
    if (comp) return false;
  }
  return true;
}

static u64 pollard_rho(u64 n) {
  if ((n & 1) == 0) return 2;
  for (u64 c = 1;; ++c) {
    u64 x = 2, y = 2, d = 1;
    while (d == 1) {
      x = (mulmod(x, x, n) + c) % n;
      y = (mulmod(y, y, n) + c) % n;
      y = (mulmod(y, y, n) + c) % n;
      d = std::gcd(x > y ? x - y : y - x, n);
    }
    if (d != n) return d;
  }
}

static void factor_rec(u64 n, std::set<u64>& out) {
  if (n == 1) return;
  if (is_prime(n)) { out.insert(n); return; }
  for (u64 p : {2ull, 3ull, 5ull, 7ull, 11ull, 13ull, 17ull, 19ull, 23ull, 29ull, 31ull, 37ull, 41ull, 43ull, 47ull})
    if (n % p == 0) {
      out.insert(p);
      while (n % p == 0) n /= p;
      factor_rec(n, out);
      return;
    }
  u64 d = pollard_rho(n);
  factor_rec(d, out);
  factor_rec(n / d, out);
}

std::vector<u64> prime_factors(u64 n) {
  std::set<u64> s;
  factor_rec(n, s);
  return std::vector<u64>(s.begin(), s.end());
}

std::vector<PP> factor_pps(u64 m) {
  std::vector<PP> out;
  for (u64 p : prime_factors(m)) {
    int e = 0;
    while (m % p == 0) { m /= p; ++e; }
    out.push_back(PP{(int)p, e});
  }
  return out;
}

i64 ipow(i64 b, int e) { i64 r = 1; while (e-- > 0) r *= b; return r; }
i64 value_pps(const std::vector<PP>& pps) { i64 m = 1; for (auto& pe : pps) m *= ipow(pe.p, pe.e); return m; }
i64 totient_pp(int p, int e) { return e == 0 ? 1 : (p - 1) * ipow(p, e - 1); }
i64 totient_pps(const std::vector<PP>& pps) { i64 n = 1; for (auto& pe : pps) n *= totient_pp(pe.p, pe.e); return n; }
i64 value_hat(i64 m) { return (m % 2 == 0) ? m / 2 : m; }
u64 odd_rad(const std::vector<PP>& pps) { u64 r = 1; for (auto& pe : pps) if (pe.p != 2) r *= (u64)pe.p; return r; }

i64 digit_rev(int p, int e, i64 j) {
  i64 acc = 0;
  for (int k = e - 1; k >= 0; --k) { acc += (j % p) * ipow(p, k); j /= p; }
  return acc;
}

u64 first_good_q(u64 m, u64 lower) {
  u64 q = lower + ((m - lower % m) % m) + 1;
  while (!is_prime(q)) q += m;
  return q;
}

u64 smallest_generator(u64 q) {
  if (q == 2) return 1;
  u64 order = q - 1;
  std::vector<u64> exps;
  for (u64 p : prime_factors(order)) exps.push_back(order / p);
  for (u64 x = 1;; ++x) {
    bool gen = true;
    for (u64 e : exps) if (powmod(x, e, q) == 1) { gen = false; break; }
    if (gen) return x;
  }
}

u64 principal_root(u64 m, u64 q) {
  if (!is_prime(q) || (q - 1) % m != 0) return 0;
  return powmod(smallest_generator(q), (q - 1) / m, q);
}

std::vector<u64> g_crt(const std::vector<PP>& pps, const std::vector<u64>& omega_p, u64 q, bool inverse) {
  // per-prime (p-1)-vectors, Tensor.hs:315-337
  std::vector<std::vector<u64>> per(pps.size());
  for (size_t k = 0; k < pps.size(); ++k) {
    int p = pps[k].p;
    if (p == 2) { per[k] = {1 % q}; continue; }
    u64 w = omega_p[k];
    per[k].resize(p - 1);
    u64 phatinv = invmod((u64)p % q, q);
    for (int i = 0; i < p - 1; ++i) {
      if (!inverse) {
        per[k][i] = (1 % q + q - powmod(w, (u64)(i + 1), q)) % q;
      } else {
        u64 acc = 0;
        for (int j = 1; j < p; ++j)
          acc = (acc + mulmod((u64)j % q, powmod(w, (u64)(i + 1) * (u64)(p - 1 - j), q), q)) % q;
        per[k][i] = mulmod(phatinv, acc, q);
      }
    }
  }
  // Kronecker expansion, smallest prime innermost (fKron/ppKron/indexK, Tensor.hs:249-288)
  i64 n = totient_pps(pps);
  std::vector<u64> out((size_t)n);
  for (i64 i = 0; i < n; ++i) {
    i64 ii = i;
    u64 acc = 1 % q;
    for (size_t k = 0; k < pps.size(); ++k) {
      i64 phi = totient_pp(pps[k].p, pps[k].e);
      i64 ik = ii % phi; ii /= phi;
      acc = mulmod(acc, per[k][(size_t)(ik % (pps[k].p - 1))], q);
    }
    out[(size_t)i] = acc;
  }
  return out;
}

bool merge_pps(const std::vector<PP>& pps, const std::vector<PP>& pps2, std::vector<MergedPP>& out) {
  out.clear();
  size_t a = 0;
  for (auto& pe2 : pps2) {
    if (a < pps.size() && pps[a].p == pe2.p) {
      if (pps[a].e > pe2.e) return false;
      out.push_back(MergedPP{pe2.p, pps[a].e, pe2.e});
      ++a;
    } else {
      if (a < pps.size() && pps[a].p < pe2.p) return false;
      out.push_back(MergedPP{pe2.p, 0, pe2.e});
    }
  }
  return a == pps.size();
}

std::pair<i64, i64> to_index_pair(const std::vector<std::pair<i64, i64>>& tots, i64 i2) {
  i64 i1 = 0, i0 = 0, r1 = 1, r0 = 1;
  for (auto& t : tots) {
    i64 d = i2 % t.second; i2 /= t.second;
    i1 += (d / t.first) * r1; r1 *= t.second / t.first;
    i0 += (d % t.first) * r0; r0 *= t.first;
  }
  return {i1, i0};
}

i64 from_index_pair(const std::vector<std::pair<i64, i64>>& tots, i64 i1, i64 i0) {
  i64 out = 0, r = 1;
  for (auto& t : tots) {
    i64 rel = t.second / t.first;
    i64 d0 = i0 % t.first; i0 /= t.first;
    i64 d1 = i1 % rel; i1 /= rel;
    out += (d0 + d1 * t.first) * r;
    r *= t.second;
  }
  return out;
}

bool build_ext_tables(const std::vector<PP>& pps, const std::vector<PP>& pps2, ExtTables& X) {
  std::vector<MergedPP> mp;
  if (!merge_pps(pps, pps2, mp)) return false;
  std::vector<std::pair<i64, i64>> tots;
  for (auto& t : mp) tots.push_back({totient_pp(t.p, t.e), totient_pp(t.p, t.e2)});
  X.phi = totient_pps(pps);
  X.phi2 = totient_pps(pps2);
  i64 rel = X.phi2 / X.phi;
  X.twace_powdec.resize((size_t)X.phi);
  for (i64 i = 0; i < X.phi; ++i) X.twace_powdec[(size_t)i] = (int32_t)from_index_pair(tots, 0, i);
  X.coeffs.resize((size_t)X.phi2);     // Tensor.hs:472-477
  for (i64 i1 = 0; i1 < rel; ++i1)
    for (i64 i0 = 0; i0 < X.phi; ++i0) X.coeffs[(size_t)(i1 * X.phi + i0)] = (int32_t)from_index_pair(tots, i1, i0);
  X.ext_crt.resize((size_t)X.phi2);
  for (i64 k = 0; k < X.phi2; ++k) X.ext_crt[(size_t)k] = (int32_t)from_index_pair(tots, k % rel, k / rel);
  X.embed_pow.resize((size_t)X.phi2);
  X.embed_crt.resize((size_t)X.phi2);
  X.embed_dec.resize((size_t)X.phi2);
  for (i64 i2 = 0; i2 < X.phi2; ++i2) {
    auto pr = to_index_pair(tots, i2);
    X.embed_pow[(size_t)i2] = pr.first == 0 ? (int32_t)pr.second : -1;
    X.embed_crt[(size_t)i2] = (int32_t)pr.second;
    // baseIndexDec, Tensor.hs:484-498
    i64 rem = i2, idx = 0, radix = 1;
    bool neg = false, none = false;
    for (auto& t : mp) {
      i64 phi2k = totient_pp(t.p, t.e2), phik = totient_pp(t.p, t.e);
      i64 d = rem % phi2k; rem /= phi2k;
      if (t.p > 2 && t.e == 0 && t.e2 > 0) {
        if (d == 0) { /* (0,False) */ }
        else if (d == 1) neg = !neg;
        else { none = true; break; }
      } else {
        if (d < phik) idx += d * radix;
        else { none = true; break; }
      }
      radix *= phik;
    }
    X.embed_dec[(size_t)i2] = none ? -1 : (int32_t)(idx | (neg ? EMBED_NEG_FLAG : 0));
  }
  return true;
}

}  // namespace lolhip
