// tests/native/host_sanitize.cpp — drives the HOST side of liblolhip (plan construction, number
// theory, index tables, the protobuf codec, argument checking of the C ABI) under
// AddressSanitizer + UndefinedBehaviorSanitizer on the CPU (sanitizers on the
// CPU build only; GPU ASan is not available on this pool).  Built and run by
// tests/test_sanitize.py with -fsanitize=address,undefined -fno-sanitize-recover.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "lolhip.h"

#define CHECK(c) do { if (!(c)) { std::printf("CHECK failed line %d: %s\n", __LINE__, #c); std::exit(1); } } while (0)

static std::vector<lolhip_pp> factor(long m) {
  std::vector<lolhip_pp> v;
  for (long p = 2; m > 1; ++p) { int e = 0; while (m % p == 0) { m /= p; ++e; } if (e) v.push_back(lolhip_pp{(int16_t)p, (int16_t)e}); }
  return v;
}

int main() {
  std::mt19937_64 rng(1);
  // ---- plans (host only), tables, extensions ----------------------------------------------
  const long ms[] = {1, 2, 8, 9, 12, 45, 64, 75, 1024, 1728, 15015, 14336, 89, 2 * 97};
  std::vector<lolhip_plan*> plans;
  for (long m : ms) {
    auto pps = factor(m);
    for (int T = 1; T <= 3; ++T) {
      std::vector<int64_t> qs;
      int64_t lower = (int64_t)1 << (10 + 17 * T);
      for (int t = 0; t < T; ++t) { lower = lolhip_good_q(m, lower); CHECK(lower > 1); qs.push_back(lower); }
      lolhip_plan* P = nullptr;
      CHECK(lolhip_plan_create(pps.data(), (int)pps.size(), qs.data(), T, 1, &P) == LOLHIP_OK && P);
      CHECK(lolhip_plan_m(P) == m && lolhip_plan_T(P) == T && lolhip_plan_has_crt(P) == 1);
      const int64_t n = lolhip_plan_n(P);
      for (int which = 0; which < 6; ++which)
        for (int k = -1; k <= (int)pps.size(); ++k) {
          const int64_t cnt = lolhip_plan_table(P, which, k, nullptr, 0);
          std::vector<int64_t> buf((size_t)cnt + 1);
          CHECK(lolhip_plan_table(P, which, k, buf.data(), cnt) == cnt);
          CHECK(lolhip_plan_table(P, which, k, buf.data(), cnt / 2) == cnt);      // short buffer: partial copy only
        }
      // compute entry points refuse to run without a device, whatever the arguments
      std::vector<int64_t> y((size_t)(n * T));
      CHECK(lolhip_crt_batch(P, nullptr, y.data(), 1) == LOLHIP_ERR_NO_DEVICE);
      CHECK(lolhip_polymul_batch(P, nullptr, y.data(), y.data(), y.data(), 1) == LOLHIP_ERR_NO_DEVICE);
      CHECK(lolhip_op_host(P, LOLHIP_OP_L, y.data(), nullptr, 1) == LOLHIP_ERR_NO_DEVICE);
      CHECK(lolhip_crtc_batch(P, nullptr, nullptr, 0) == LOLHIP_ERR_NO_DEVICE);
      for (int64_t base : {(int64_t)0, (int64_t)2, (int64_t)256, (int64_t)1 << 40}) {
        const int L = lolhip_decompose_len(P, base);
        CHECK(L >= T);
        std::vector<int64_t> g((size_t)L * T);
        CHECK(lolhip_gadget(P, base, g.data(), (int64_t)g.size()) == L);
        CHECK(lolhip_gadget(P, base, g.data(), (int64_t)g.size() - 1) == LOLHIP_ERR_INVALID);
      }
      CHECK(lolhip_decompose_len(P, 1) == LOLHIP_ERR_INVALID && lolhip_decompose_len(P, -7) == LOLHIP_ERR_INVALID);
      plans.push_back(P);
    }
  }
  {  // malformed plan requests
    lolhip_plan* P = nullptr;
    lolhip_pp bad1[] = {{4, 1}}, bad2[] = {{3, 1}, {2, 2}}, bad3[] = {{2, 0}};
    int64_t q = 97, q0 = 1, qbig = (int64_t)1 << 62;
    CHECK(lolhip_plan_create(bad1, 1, &q, 1, 1, &P) == LOLHIP_ERR_INVALID);
    CHECK(lolhip_plan_create(bad2, 2, &q, 1, 1, &P) == LOLHIP_ERR_INVALID);
    CHECK(lolhip_plan_create(bad3, 1, &q, 1, 1, &P) == LOLHIP_ERR_INVALID);
    lolhip_pp ok[] = {{2, 3}};
    CHECK(lolhip_plan_create(ok, 1, &q0, 1, 1, &P) == LOLHIP_ERR_MODULUS);
    CHECK(lolhip_plan_create(ok, 1, &qbig, 1, 1, &P) == LOLHIP_ERR_MODULUS);
    CHECK(lolhip_plan_create(ok, 1, &q, 0, 1, &P) == LOLHIP_ERR_INVALID);
    int64_t wrong_root = 5;       // not of order 8 mod 97
    CHECK(lolhip_plan_create_roots(ok, 1, &q, 1, &wrong_root, nullptr, 1, &P) == LOLHIP_ERR_ROOT);
    int64_t q17 = 17;             // 8 | 16: fine; 45 does not divide 16: a plan without CRT basis is still a plan
    auto p45 = factor(45);
    CHECK(lolhip_plan_create(p45.data(), (int)p45.size(), &q17, 1, 1, &P) == LOLHIP_OK);
    CHECK(lolhip_plan_has_crt(P) == 0 && lolhip_plan_table(P, 0, 0, nullptr, 0) == 0);
    lolhip_plan_destroy(P);
    lolhip_plan_destroy(nullptr);
  }
  {  // ring extensions m | m'
    const long pairs[][2] = {{1, 8}, {4, 12}, {3, 21}, {8, 8}, {45, 45 * 7}, {128, 128 * 7 * 13}};
    for (auto& pr : pairs) {
      auto a = factor(pr[0]), b = factor(pr[1]);
      int64_t q = lolhip_good_q(pr[1], 1000);
      lolhip_plan *lo = nullptr, *hi = nullptr;
      CHECK(lolhip_plan_create(a.data(), (int)a.size(), &q, 1, 1, &lo) == LOLHIP_OK);
      CHECK(lolhip_plan_create(b.data(), (int)b.size(), &q, 1, 1, &hi) == LOLHIP_OK);
      lolhip_ext* X = nullptr;
      CHECK(lolhip_ext_create(lo, hi, &X) == LOLHIP_OK && X);
      for (int which = 0; which < 7; ++which) {
        const int64_t cnt = lolhip_ext_table(X, which, nullptr, 0);
        std::vector<int32_t> t((size_t)cnt + 1);
        CHECK(lolhip_ext_table(X, which, t.data(), cnt) == cnt);
      }
      lolhip_ext* Y = nullptr;
      if (pr[0] != pr[1]) CHECK(lolhip_ext_create(hi, lo, &Y) == LOLHIP_ERR_INVALID);     // m' does not divide m
      lolhip_ext_destroy(X);
      lolhip_plan_destroy(lo); lolhip_plan_destroy(hi);
    }
  }
  for (lolhip_plan* P : plans) lolhip_plan_destroy(P);

  // ---- wire format: round trips, every truncation, random mutations ----------------------------
  const int64_t qs[] = {97, ((int64_t)1 << 40) + 15, 12289};
  const int T = 3, n = 24, L = 3, K = 2;
  std::vector<int64_t> xs((size_t)L * K * n * T);
  for (size_t i = 0; i < xs.size(); ++i) xs[i] = (int64_t)(rng() % (uint64_t)qs[i % T]) - (i % 5 == 0 ? qs[i % T] / 2 : 0);
  const int64_t len1 = lolhip_rqproduct_write(12, qs, T, xs.data(), n, nullptr, 0);
  CHECK(len1 > 0);
  std::vector<uint8_t> rq((size_t)len1);
  CHECK(lolhip_rqproduct_write(12, qs, T, xs.data(), n, rq.data(), len1) == len1);
  CHECK(lolhip_rqproduct_write(12, qs, T, xs.data(), n, rq.data(), len1 - 1) == LOLHIP_ERR_INVALID);
  const int64_t len2 = lolhip_kshint_write(12, qs, T, L, K, xs.data(), n, 1, 2, nullptr, 0);
  std::vector<uint8_t> ks((size_t)len2);
  CHECK(lolhip_kshint_write(12, qs, T, L, K, xs.data(), n, 1, 2, ks.data(), len2) == len2);
  auto read_all = [&](const uint8_t* b, int64_t l) {
    uint32_t m, e, r, s; int Tt, Ll, Kk, C; int64_t q[8]; uint64_t p; double v;
    std::vector<int64_t> out((size_t)L * K * n * T);
    std::vector<double> outd((size_t)n * T);
    int64_t fo, fl, ho[4], hl[4];
    (void)lolhip_rqproduct_read(b, l, &m, q, 8, &Tt, out.data(), (int64_t)out.size());
    (void)lolhip_rqproduct_read(b, l, &m, q, 1, &Tt, out.data(), 3);
    (void)lolhip_kshint_read(b, l, &m, q, 8, &Tt, &Ll, &Kk, out.data(), (int64_t)out.size());
    (void)lolhip_kshint_read(b, l, &m, q, 8, &Tt, &Ll, &Kk, nullptr, 0);
    (void)lolhip_r_read(b, l, &m, out.data(), (int64_t)out.size());
    (void)lolhip_secretkey_read(b, l, &m, &v, out.data(), 2);
    (void)lolhip_kqproduct_read(b, l, &m, q, 8, &Tt, outd.data(), (int64_t)outd.size());
    (void)lolhip_linearrq_read(b, l, &e, &r, &C, &m, q, 8, &Tt, out.data(), (int64_t)out.size());
    (void)lolhip_tunnelhint_read(b, l, &e, &r, &s, &p, &fo, &fl, ho, hl, 4);
    (void)lolhip_chain_read(b, l, ho, hl, 4);
    (void)lolhip_chain_read(b, l, nullptr, nullptr, 0);
  };
  {
    uint32_t m; int Tt, Ll, Kk; int64_t q[8];
    std::vector<int64_t> back(xs.size());
    CHECK(lolhip_kshint_read(ks.data(), len2, &m, q, 8, &Tt, &Ll, &Kk, back.data(), (int64_t)back.size()) == n);
    CHECK(m == 12 && Tt == T && Ll == L && Kk == K);
    for (size_t i = 0; i < xs.size(); ++i) { int64_t w = xs[i] % qs[i % T]; if (w < 0) w += qs[i % T]; CHECK(back[i] == w); }
  }
  // round 3 writers: LinearRq, TunnelHint, a chain of them, R, SecretKey — size query = bytes written, short buffers refused
  const int64_t len3 = lolhip_linearrq_write(4, 12, 12, qs, T, K, xs.data(), n, nullptr, 0);
  std::vector<uint8_t> lin((size_t)len3);
  CHECK(len3 > 0 && lolhip_linearrq_write(4, 12, 12, qs, T, K, xs.data(), n, lin.data(), len3) == len3);
  CHECK(lolhip_linearrq_write(4, 12, 12, qs, T, K, xs.data(), n, lin.data(), len3 - 1) == LOLHIP_ERR_INVALID);
  const uint8_t* hp[2] = {ks.data(), ks.data()};
  const int64_t hl2[2] = {len2, len2};
  const int64_t len4 = lolhip_tunnelhint_write(lin.data(), len3, hp, hl2, 2, 4, 12, 12, 257, nullptr, 0);
  std::vector<uint8_t> th((size_t)len4);
  CHECK(len4 > 0 && lolhip_tunnelhint_write(lin.data(), len3, hp, hl2, 2, 4, 12, 12, 257, th.data(), len4) == len4);
  {
    uint32_t e, r, s2; uint64_t p; int64_t fo, fl, ho[4], hl[4];
    CHECK(lolhip_tunnelhint_read(th.data(), len4, &e, &r, &s2, &p, &fo, &fl, ho, hl, 4) == 2);
    CHECK(e == 4 && r == 12 && s2 == 12 && p == 257 && fl == len3 && hl[0] == len2 && hl[1] == len2);
  }
  const uint8_t* cp[3] = {th.data(), th.data(), th.data()};
  const int64_t cl[3] = {len4, len4, len4};
  const int64_t len5 = lolhip_chain_write(cp, cl, 3, nullptr, 0);
  std::vector<uint8_t> chain((size_t)len5);
  CHECK(len5 > 0 && lolhip_chain_write(cp, cl, 3, chain.data(), len5) == len5);
  { int64_t off[4], ln[4]; CHECK(lolhip_chain_read(chain.data(), len5, off, ln, 4) == 3 && ln[2] == len4); }
  const int64_t rv[5] = {-3, 0, 7, ((int64_t)1 << 50), -((int64_t)1 << 50)};
  const int64_t len6 = lolhip_secretkey_write(12, 0.5, rv, 5, nullptr, 0);
  std::vector<uint8_t> skb((size_t)len6);
  CHECK(len6 > 0 && lolhip_secretkey_write(12, 0.5, rv, 5, skb.data(), len6) == len6);
  { uint32_t m; double v; int64_t back[5]; CHECK(lolhip_secretkey_read(skb.data(), len6, &m, &v, back, 5) == 5 && m == 12 && v == 0.5 && back[4] == rv[4]); }
  for (const std::vector<uint8_t>* src : {&rq, &ks, &chain, &skb}) {
    for (int64_t l = 0; l <= (int64_t)src->size(); ++l) {          // every truncation, exact-size heap copy: overreads trap
      std::vector<uint8_t> cut(src->begin(), src->begin() + l);
      read_all(cut.data(), l);
    }
    for (int it = 0; it < 2500; ++it) {                             // random byte mutations
      std::vector<uint8_t> mut(*src);
      const int flips = 1 + (int)(rng() % 3);
      for (int f = 0; f < flips; ++f) mut[rng() % mut.size()] = (uint8_t)rng();
      read_all(mut.data(), (int64_t)mut.size());
    }
  }
  std::printf("host sanitize ok\n");
  return 0;
}
