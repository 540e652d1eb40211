// lol_amd/csrc/wire.cpp — reader/writer for Lol's protobuf ring-element messages (host side).
//
// Messages (lol/Lol.proto, proto2):
//   message Rq        { required uint32 m = 1; required uint64 q = 2; repeated sint64 xs = 3; }
//   message RqProduct { repeated Rq rqlist = 1; }
//   message R         { required uint32 m = 1; repeated sint64 xs = 2; }
//   message Kq / KqProduct: as Rq / RqProduct with `repeated double xs`
//   message LinearRq  { required uint32 e = 1; required uint32 r = 2; repeated RqProduct coeffs = 3; }
// and of lol-apps/SHE.proto: SecretKey, RqPolynomial, KSHint (read and write), TunnelHint.
// Conventions (lol/Crypto/Lol/Types/IZipVector.hs:127-205, Lol.proto:18-20): one Rq per
// modulus of the RNS tuple, first component first; xs are the coefficients in the DECODING
// basis, written as centred lifts in [-q/2, q/2) (toProto: `LP.lift`), read back with `reduce`.
// So ingest for the GPU path is: read -> slab [n][T] -> lolhip_l_batch -> lolhip_crt_batch.
//
// The codec is written against the protobuf wire specification (varint, zigzag, length-
// delimited sub-messages); it accepts both the unpacked encoding that proto2 `repeated sint64`
// uses by default (what hprotoc emits) and the packed one.  Unknown fields are skipped.
#include <cstdint>
#include <cstring>
#include <vector>

#include "lolhip.h"

namespace {

struct Reader {
  const uint8_t* p; const uint8_t* end; bool ok = true;
  bool varint(uint64_t& v) {
    v = 0;
    for (int shift = 0; shift < 64; shift += 7) {
      if (p >= end) return ok = false;
      const uint8_t b = *p++;
      v |= (uint64_t)(b & 0x7F) << shift;
      if (!(b & 0x80)) return true;
    }
    return ok = false;
  }
  bool skip(uint32_t wt) {
    uint64_t v;
    switch (wt) {
      case 0: return varint(v);
      case 1: if (end - p < 8) return ok = false; p += 8; return true;
      case 2: if (!varint(v) || (uint64_t)(end - p) < v) return ok = false; p += v; return true;
      case 5: if (end - p < 4) return ok = false; p += 4; return true;
      default: return ok = false;
    }
  }
};

inline int64_t unzigzag(uint64_t z) { return (int64_t)(z >> 1) ^ -(int64_t)(z & 1); }
inline uint64_t zigzag(int64_t v) { return ((uint64_t)v << 1) ^ (uint64_t)(v >> 63); }

inline int varint_len(uint64_t v) { int n = 1; while (v >= 0x80) { v >>= 7; ++n; } return n; }
inline void put_varint(uint8_t*& o, uint64_t v) { while (v >= 0x80) { *o++ = (uint8_t)(v | 0x80); v >>= 7; } *o++ = (uint8_t)v; }

struct RqView { uint32_t m = 0; uint64_t q = 0; bool has_m = false, has_q = false; std::vector<int64_t> xs; };

bool parse_rq(const uint8_t* b, const uint8_t* e, RqView& r) {
  Reader rd{b, e};
  while (rd.p < rd.end) {
    uint64_t key;
    if (!rd.varint(key)) return false;
    const uint32_t field = (uint32_t)(key >> 3), wt = (uint32_t)(key & 7);
    uint64_t v;
    if (field == 1 && wt == 0) { if (!rd.varint(v)) return false; r.m = (uint32_t)v; r.has_m = true; }
    else if (field == 2 && wt == 0) { if (!rd.varint(v)) return false; r.q = v; r.has_q = true; }
    else if (field == 3 && wt == 0) { if (!rd.varint(v)) return false; r.xs.push_back(unzigzag(v)); }
    else if (field == 3 && wt == 2) {                      // packed
      if (!rd.varint(v) || (uint64_t)(rd.end - rd.p) < v) return false;
      Reader in{rd.p, rd.p + v};
      while (in.p < in.end) { uint64_t z; if (!in.varint(z)) return false; r.xs.push_back(unzigzag(z)); }
      rd.p += v;
    } else if (!rd.skip(wt)) return false;
  }
  return r.has_m && r.has_q;                                // both are `required`
}

}  // namespace

extern "C" {

int64_t lolhip_rqproduct_read(const uint8_t* buf, int64_t len, uint32_t* m_out, int64_t* qs, int cap_T, int* T_out,
                              int64_t* xs, int64_t cap_xs) {
  if (!buf || len < 0) return LOLHIP_ERR_INVALID;
  std::vector<RqView> list;
  Reader rd{buf, buf + len};
  while (rd.p < rd.end) {
    uint64_t key;
    if (!rd.varint(key)) return LOLHIP_ERR_INVALID;
    const uint32_t field = (uint32_t)(key >> 3), wt = (uint32_t)(key & 7);
    if (field == 1 && wt == 2) {
      uint64_t l;
      if (!rd.varint(l) || (uint64_t)(rd.end - rd.p) < l) return LOLHIP_ERR_INVALID;
      list.emplace_back();
      if (!parse_rq(rd.p, rd.p + l, list.back())) return LOLHIP_ERR_INVALID;
      rd.p += l;
    } else if (!rd.skip(wt)) return LOLHIP_ERR_INVALID;
  }
  const int T = (int)list.size();
  if (T == 0) return LOLHIP_ERR_INVALID;
  const int64_t n = (int64_t)list[0].xs.size();
  for (const RqView& r : list) {
    if (r.m != list[0].m || (int64_t)r.xs.size() != n) return LOLHIP_ERR_INVALID;   // one ring, one length
    if (r.q < 2 || r.q >= ((uint64_t)1 << 62)) return LOLHIP_ERR_MODULUS;
  }
  if (m_out) *m_out = list[0].m;
  if (T_out) *T_out = T;
  if (qs) {
    if (cap_T < T) return LOLHIP_ERR_INVALID;
    for (int t = 0; t < T; ++t) qs[t] = (int64_t)list[(size_t)t].q;
  }
  if (xs) {
    if (cap_xs < n * T) return LOLHIP_ERR_INVALID;
    for (int t = 0; t < T; ++t) {
      const int64_t q = (int64_t)list[(size_t)t].q;
      for (int64_t j = 0; j < n; ++j) {                     // `reduce`: canonical residue of any integer
        int64_t r = list[(size_t)t].xs[(size_t)j] % q;
        xs[j * T + t] = r < 0 ? r + q : r;
      }
    }
  }
  return n;
}

int64_t lolhip_rqproduct_write(uint32_t m, const int64_t* qs, int T, const int64_t* xs, int64_t n, uint8_t* out,
                               int64_t cap) {
  if (!qs || T < 1 || n < 0 || (n > 0 && !xs)) return LOLHIP_ERR_INVALID;
  for (int t = 0; t < T; ++t) if (qs[t] < 2) return LOLHIP_ERR_MODULUS;
  // centred lift of component t of coefficient j (ZqBasic.hs:92-94), accepting (-q, q) inputs
  auto lifted = [&](int64_t j, int t) {
    const int64_t q = qs[t];
    int64_t x = xs[j * T + t] % q;
    if (x < 0) x += q;
    return (x < q - x) ? x : x - q;                         // 2x < q  without overflow
  };
  std::vector<int64_t> body((size_t)T);
  int64_t total = 0;
  for (int t = 0; t < T; ++t) {
    int64_t b = 1 + varint_len(m) + 1 + varint_len((uint64_t)qs[t]);
    for (int64_t j = 0; j < n; ++j) b += 1 + varint_len(zigzag(lifted(j, t)));
    body[(size_t)t] = b;
    total += 1 + varint_len((uint64_t)b) + b;
  }
  if (!out) return total;
  if (cap < total) return LOLHIP_ERR_INVALID;
  uint8_t* o = out;
  for (int t = 0; t < T; ++t) {
    *o++ = 0x0A; put_varint(o, (uint64_t)body[(size_t)t]);  // rqlist = 1, length-delimited
    *o++ = 0x08; put_varint(o, m);                          // m = 1
    *o++ = 0x10; put_varint(o, (uint64_t)qs[t]);            // q = 2
    for (int64_t j = 0; j < n; ++j) { *o++ = 0x18; put_varint(o, zigzag(lifted(j, t))); }   // xs = 3, unpacked sint64
  }
  return (int64_t)(o - out);
}

// KSHint (lol-apps/SHE.proto): { repeated RqPolynomial hint = 1; required TypeRep gad = 2; }
// RqPolynomial: { repeated RqProduct coeffs = 1; }  constant coefficient first.
// Reads the L hint polynomials of K coefficients each into xs [L][K][n][T] (decoding basis,
// canonical residues) — after l and crt over the L*K polynomials, the slab lolhip_keyswitch_batch
// takes.  The gadget TypeRep (a GHC fingerprint, "not intended to be platform independent",
// Lol.proto) is skipped.  Returns n; pass xs = NULL to query L, K, T, m, qs.
int64_t lolhip_kshint_read(const uint8_t* buf, int64_t len, uint32_t* m_out, int64_t* qs, int cap_T, int* T_out,
                           int* L_out, int* K_out, int64_t* xs, int64_t cap_xs) {
  if (!buf || len < 0) return LOLHIP_ERR_INVALID;
  struct Span { const uint8_t* p; int64_t n; };
  std::vector<std::vector<Span>> polys;
  Reader rd{buf, buf + len};
  while (rd.p < rd.end) {
    uint64_t key;
    if (!rd.varint(key)) return LOLHIP_ERR_INVALID;
    const uint32_t field = (uint32_t)(key >> 3), wt = (uint32_t)(key & 7);
    if (field == 1 && wt == 2) {
      uint64_t l;
      if (!rd.varint(l) || (uint64_t)(rd.end - rd.p) < l) return LOLHIP_ERR_INVALID;
      polys.emplace_back();
      Reader in{rd.p, rd.p + l};
      while (in.p < in.end) {
        uint64_t k2;
        if (!in.varint(k2)) return LOLHIP_ERR_INVALID;
        if ((k2 >> 3) == 1 && (k2 & 7) == 2) {
          uint64_t l2;
          if (!in.varint(l2) || (uint64_t)(in.end - in.p) < l2) return LOLHIP_ERR_INVALID;
          polys.back().push_back(Span{in.p, (int64_t)l2});
          in.p += l2;
        } else if (!in.skip((uint32_t)(k2 & 7))) return LOLHIP_ERR_INVALID;
      }
      rd.p += l;
    } else if (!rd.skip(wt)) return LOLHIP_ERR_INVALID;
  }
  const int L = (int)polys.size();
  if (L == 0) return LOLHIP_ERR_INVALID;
  const int K = (int)polys[0].size();
  if (K == 0) return LOLHIP_ERR_INVALID;
  for (auto& pl : polys) if ((int)pl.size() != K) return LOLHIP_ERR_INVALID;
  uint32_t m0 = 0; int T0 = 0;
  std::vector<int64_t> q0(cap_T > 0 ? (size_t)cap_T : 16);
  const int64_t n = lolhip_rqproduct_read(polys[0][0].p, polys[0][0].n, &m0, q0.data(), (int)q0.size(), &T0, nullptr, 0);
  if (n < 0) return n;
  if (m_out) *m_out = m0;
  if (T_out) *T_out = T0;
  if (L_out) *L_out = L;
  if (K_out) *K_out = K;
  if (qs) { if (cap_T < T0) return LOLHIP_ERR_INVALID; for (int t = 0; t < T0; ++t) qs[t] = q0[(size_t)t]; }
  if (!xs) return n;
  if (cap_xs < (int64_t)L * K * n * T0) return LOLHIP_ERR_INVALID;
  std::vector<int64_t> q1(q0.size());
  for (int j = 0; j < L; ++j)
    for (int k = 0; k < K; ++k) {
      uint32_t m1 = 0; int T1 = 0;
      const int64_t n1 = lolhip_rqproduct_read(polys[(size_t)j][(size_t)k].p, polys[(size_t)j][(size_t)k].n, &m1, q1.data(),
                                               (int)q1.size(), &T1, xs + ((int64_t)j * K + k) * n * T0, n * T0);
      if (n1 < 0) return n1;
      if (n1 != n || m1 != m0 || T1 != T0) return LOLHIP_ERR_INVALID;     // every entry over the same ring and moduli
      for (int t = 0; t < T0; ++t) if (q1[(size_t)t] != q0[(size_t)t]) return LOLHIP_ERR_INVALID;
    }
  return n;
}

}  // extern "C"

// ---- the remaining messages of Lol.proto / SHE.proto ---------------------

namespace {

// one length-delimited sub-message: returns false on a malformed length
bool sub(Reader& rd, const uint8_t*& b, int64_t& l) {
  uint64_t v;
  if (!rd.varint(v) || (uint64_t)(rd.end - rd.p) < v) return false;
  b = rd.p; l = (int64_t)v; rd.p += v;
  return true;
}

// message R { required uint32 m = 1; repeated sint64 xs = 2; }
int64_t parse_r(const uint8_t* buf, int64_t len, uint32_t* m_out, int64_t* xs, int64_t cap) {
  Reader rd{buf, buf + len};
  bool has_m = false;
  int64_t n = 0;
  auto put = [&](int64_t v) { if (xs && n < cap) xs[n] = v; ++n; };
  while (rd.p < rd.end) {
    uint64_t key, v;
    if (!rd.varint(key)) return LOLHIP_ERR_INVALID;
    const uint32_t field = (uint32_t)(key >> 3), wt = (uint32_t)(key & 7);
    if (field == 1 && wt == 0) { if (!rd.varint(v)) return LOLHIP_ERR_INVALID; if (m_out) *m_out = (uint32_t)v; has_m = true; }
    else if (field == 2 && wt == 0) { if (!rd.varint(v)) return LOLHIP_ERR_INVALID; put(unzigzag(v)); }
    else if (field == 2 && wt == 2) {
      const uint8_t* b; int64_t l;
      if (!sub(rd, b, l)) return LOLHIP_ERR_INVALID;
      Reader in{b, b + l};
      while (in.p < in.end) { uint64_t z; if (!in.varint(z)) return LOLHIP_ERR_INVALID; put(unzigzag(z)); }
    } else if (!rd.skip(wt)) return LOLHIP_ERR_INVALID;
  }
  if (!has_m) return LOLHIP_ERR_INVALID;
  if (xs && n > cap) return LOLHIP_ERR_INVALID;
  return n;
}

inline double f64_at(const uint8_t* p) { double d; std::memcpy(&d, p, 8); return d; }

}  // namespace

extern "C" {

int64_t lolhip_r_read(const uint8_t* buf, int64_t len, uint32_t* m, int64_t* xs, int64_t cap_xs) {
  if (!buf || len < 0) return LOLHIP_ERR_INVALID;
  return parse_r(buf, len, m, xs, cap_xs);
}

// message SecretKey { required R sk = 1; required double v = 2; }  (SHE.proto:9)
int64_t lolhip_secretkey_read(const uint8_t* buf, int64_t len, uint32_t* m, double* v_out, int64_t* xs, int64_t cap_xs) {
  if (!buf || len < 0) return LOLHIP_ERR_INVALID;
  Reader rd{buf, buf + len};
  const uint8_t* rb = nullptr; int64_t rl = 0;
  bool has_v = false;
  while (rd.p < rd.end) {
    uint64_t key;
    if (!rd.varint(key)) return LOLHIP_ERR_INVALID;
    const uint32_t field = (uint32_t)(key >> 3), wt = (uint32_t)(key & 7);
    if (field == 1 && wt == 2) { if (!sub(rd, rb, rl)) return LOLHIP_ERR_INVALID; }
    else if (field == 2 && wt == 1) { if (rd.end - rd.p < 8) return LOLHIP_ERR_INVALID; if (v_out) *v_out = f64_at(rd.p); rd.p += 8; has_v = true; }
    else if (!rd.skip(wt)) return LOLHIP_ERR_INVALID;
  }
  if (!rb || !has_v) return LOLHIP_ERR_INVALID;
  return parse_r(rb, rl, m, xs, cap_xs);
}

// message KqProduct { repeated Kq kqlist = 1; }, Kq { m = 1; q = 2; repeated double xs = 3; }
// -> xs [n][T] doubles (decoding basis, as written), qs [T].  Returns n.
int64_t lolhip_kqproduct_read(const uint8_t* buf, int64_t len, uint32_t* m_out, int64_t* qs, int cap_T, int* T_out,
                              double* xs, int64_t cap_xs) {
  if (!buf || len < 0) return LOLHIP_ERR_INVALID;
  struct KqView { uint32_t m = 0; uint64_t q = 0; bool hm = false, hq = false; std::vector<double> xs; };
  std::vector<KqView> list;
  Reader rd{buf, buf + len};
  while (rd.p < rd.end) {
    uint64_t key;
    if (!rd.varint(key)) return LOLHIP_ERR_INVALID;
    const uint32_t field = (uint32_t)(key >> 3), wt = (uint32_t)(key & 7);
    if (field == 1 && wt == 2) {
      const uint8_t* b; int64_t l;
      if (!sub(rd, b, l)) return LOLHIP_ERR_INVALID;
      list.emplace_back();
      KqView& k = list.back();
      Reader in{b, b + l};
      while (in.p < in.end) {
        uint64_t k2, v;
        if (!in.varint(k2)) return LOLHIP_ERR_INVALID;
        const uint32_t f2 = (uint32_t)(k2 >> 3), w2 = (uint32_t)(k2 & 7);
        if (f2 == 1 && w2 == 0) { if (!in.varint(v)) return LOLHIP_ERR_INVALID; k.m = (uint32_t)v; k.hm = true; }
        else if (f2 == 2 && w2 == 0) { if (!in.varint(v)) return LOLHIP_ERR_INVALID; k.q = v; k.hq = true; }
        else if (f2 == 3 && w2 == 1) { if (in.end - in.p < 8) return LOLHIP_ERR_INVALID; k.xs.push_back(f64_at(in.p)); in.p += 8; }
        else if (f2 == 3 && w2 == 2) {
          const uint8_t* pb; int64_t pl;
          if (!sub(in, pb, pl) || pl % 8) return LOLHIP_ERR_INVALID;
          for (int64_t i = 0; i < pl; i += 8) k.xs.push_back(f64_at(pb + i));
        } else if (!in.skip(w2)) return LOLHIP_ERR_INVALID;
      }
      if (!k.hm || !k.hq) return LOLHIP_ERR_INVALID;
    } else if (!rd.skip(wt)) return LOLHIP_ERR_INVALID;
  }
  const int T = (int)list.size();
  if (T == 0) return LOLHIP_ERR_INVALID;
  const int64_t n = (int64_t)list[0].xs.size();
  for (const KqView& k : list) if (k.m != list[0].m || (int64_t)k.xs.size() != n) return LOLHIP_ERR_INVALID;
  if (m_out) *m_out = list[0].m;
  if (T_out) *T_out = T;
  if (qs) { if (cap_T < T) return LOLHIP_ERR_INVALID; for (int t = 0; t < T; ++t) qs[t] = (int64_t)list[(size_t)t].q; }
  if (xs) {
    if (cap_xs < n * T) return LOLHIP_ERR_INVALID;
    for (int t = 0; t < T; ++t) for (int64_t j = 0; j < n; ++j) xs[j * T + t] = list[(size_t)t].xs[(size_t)j];
  }
  return n;
}

// message LinearRq { required uint32 e = 1; required uint32 r = 2; repeated RqProduct coeffs = 3; }
// (Lol.proto:11): the values of an E-linear function on the decoding basis of R/E — after l and
// crt, the ys of lolhip_evallin_batch.  xs [C][n][T] (decoding basis, canonical residues); n, T,
// qs are those of the output ring's RqProducts.  Returns n; xs = NULL queries e, r, C, m, T, qs.
int64_t lolhip_linearrq_read(const uint8_t* buf, int64_t len, uint32_t* e_out, uint32_t* r_out, int* C_out,
                             uint32_t* m_out, int64_t* qs, int cap_T, int* T_out, int64_t* xs, int64_t cap_xs) {
  if (!buf || len < 0) return LOLHIP_ERR_INVALID;
  struct Span { const uint8_t* p; int64_t n; };
  std::vector<Span> cs;
  bool he = false, hr = false;
  Reader rd{buf, buf + len};
  while (rd.p < rd.end) {
    uint64_t key, v;
    if (!rd.varint(key)) return LOLHIP_ERR_INVALID;
    const uint32_t field = (uint32_t)(key >> 3), wt = (uint32_t)(key & 7);
    if (field == 1 && wt == 0) { if (!rd.varint(v)) return LOLHIP_ERR_INVALID; if (e_out) *e_out = (uint32_t)v; he = true; }
    else if (field == 2 && wt == 0) { if (!rd.varint(v)) return LOLHIP_ERR_INVALID; if (r_out) *r_out = (uint32_t)v; hr = true; }
    else if (field == 3 && wt == 2) { const uint8_t* b; int64_t l; if (!sub(rd, b, l)) return LOLHIP_ERR_INVALID; cs.push_back(Span{b, l}); }
    else if (!rd.skip(wt)) return LOLHIP_ERR_INVALID;
  }
  if (!he || !hr) return LOLHIP_ERR_INVALID;
  const int C = (int)cs.size();
  if (C_out) *C_out = C;
  if (C == 0) { if (T_out) *T_out = 0; return 0; }
  uint32_t m0 = 0; int T0 = 0;
  std::vector<int64_t> q0(cap_T > 0 ? (size_t)cap_T : 16), q1(q0.size());
  const int64_t n = lolhip_rqproduct_read(cs[0].p, cs[0].n, &m0, q0.data(), (int)q0.size(), &T0, nullptr, 0);
  if (n < 0) return n;
  if (m_out) *m_out = m0;
  if (T_out) *T_out = T0;
  if (qs) { if (cap_T < T0) return LOLHIP_ERR_INVALID; for (int t = 0; t < T0; ++t) qs[t] = q0[(size_t)t]; }
  if (!xs) return n;
  if (cap_xs < (int64_t)C * n * T0) return LOLHIP_ERR_INVALID;
  for (int i = 0; i < C; ++i) {
    uint32_t m1 = 0; int T1 = 0;
    const int64_t n1 = lolhip_rqproduct_read(cs[(size_t)i].p, cs[(size_t)i].n, &m1, q1.data(), (int)q1.size(), &T1,
                                             xs + (int64_t)i * n * T0, n * T0);
    if (n1 < 0) return n1;
    if (n1 != n || m1 != m0 || T1 != T0) return LOLHIP_ERR_INVALID;
    for (int t = 0; t < T0; ++t) if (q1[(size_t)t] != q0[(size_t)t]) return LOLHIP_ERR_INVALID;
  }
  return n;
}

// KSHint writer: xs [L][K][n][T] decoding-basis residues -> bytes (the inverse of lolhip_kshint_read;
// gad_a / gad_b are the two words of the gadget's GHC fingerprint, TypeRep, copied through).
int64_t lolhip_kshint_write(uint32_t m, const int64_t* qs, int T, int L, int K, const int64_t* xs, int64_t n,
                            uint64_t gad_a, uint64_t gad_b, uint8_t* out, int64_t cap) {
  if (!qs || T < 1 || L < 1 || K < 1 || n < 0 || (n > 0 && !xs)) return LOLHIP_ERR_INVALID;
  std::vector<int64_t> prod((size_t)L * K), poly((size_t)L);
  int64_t total = 0;
  for (int j = 0; j < L; ++j) {
    int64_t pl = 0;
    for (int k = 0; k < K; ++k) {
      const int64_t b = lolhip_rqproduct_write(m, qs, T, xs + ((int64_t)j * K + k) * n * T, n, nullptr, 0);
      if (b < 0) return b;
      prod[(size_t)j * K + k] = b;
      pl += 1 + varint_len((uint64_t)b) + b;
    }
    poly[(size_t)j] = pl;
    total += 1 + varint_len((uint64_t)pl) + pl;
  }
  const int64_t gl = 1 + varint_len(gad_a) + 1 + varint_len(gad_b);
  total += 1 + varint_len((uint64_t)gl) + gl;
  if (!out) return total;
  if (cap < total) return LOLHIP_ERR_INVALID;
  uint8_t* o = out;
  for (int j = 0; j < L; ++j) {
    *o++ = 0x0A; put_varint(o, (uint64_t)poly[(size_t)j]);             // hint = 1
    for (int k = 0; k < K; ++k) {
      *o++ = 0x0A; put_varint(o, (uint64_t)prod[(size_t)j * K + k]);   // coeffs = 1
      const int64_t w = lolhip_rqproduct_write(m, qs, T, xs + ((int64_t)j * K + k) * n * T, n, o, prod[(size_t)j * K + k]);
      if (w < 0) return w;
      o += w;
    }
  }
  *o++ = 0x12; put_varint(o, (uint64_t)gl);                            // gad = 2
  *o++ = 0x08; put_varint(o, gad_a);
  *o++ = 0x10; put_varint(o, gad_b);
  return (int64_t)(o - out);
}

// message TunnelHint { LinearRq func = 1; repeated KSHint hint = 2; e = 3; r = 4; s = 5; p = 6; }
// (SHE.proto:26): one ring switch.  Returns the number of KSHints and the byte ranges
// (offset, length into buf) of the embedded LinearRq and of up to cap_hints KSHints, to be handed
// to lolhip_linearrq_read / lolhip_kshint_read.
int64_t lolhip_tunnelhint_read(const uint8_t* buf, int64_t len, uint32_t* e, uint32_t* r, uint32_t* s, uint64_t* p,
                               int64_t* func_off, int64_t* func_len, int64_t* hint_off, int64_t* hint_len, int cap_hints) {
  if (!buf || len < 0) return LOLHIP_ERR_INVALID;
  Reader rd{buf, buf + len};
  bool hf = false, he = false, hr = false, hs = false, hp = false;
  int64_t nh = 0;
  while (rd.p < rd.end) {
    uint64_t key, v;
    if (!rd.varint(key)) return LOLHIP_ERR_INVALID;
    const uint32_t field = (uint32_t)(key >> 3), wt = (uint32_t)(key & 7);
    if ((field == 1 || field == 2) && wt == 2) {
      const uint8_t* b; int64_t l;
      if (!sub(rd, b, l)) return LOLHIP_ERR_INVALID;
      if (field == 1) { if (func_off) *func_off = b - buf; if (func_len) *func_len = l; hf = true; }
      else { if (nh < cap_hints) { if (hint_off) hint_off[nh] = b - buf; if (hint_len) hint_len[nh] = l; } ++nh; }
    } else if (field >= 3 && field <= 6 && wt == 0) {
      if (!rd.varint(v)) return LOLHIP_ERR_INVALID;
      if (field == 3) { if (e) *e = (uint32_t)v; he = true; }
      else if (field == 4) { if (r) *r = (uint32_t)v; hr = true; }
      else if (field == 5) { if (s) *s = (uint32_t)v; hs = true; }
      else { if (p) *p = v; hp = true; }
    } else if (!rd.skip(wt)) return LOLHIP_ERR_INVALID;
  }
  if (!(hf && he && hr && hs && hp)) return LOLHIP_ERR_INVALID;        // all `required`
  return nh;
}

// ---- round 3: the writers still missing, and the chain messages of lol-apps/HomomPRF.proto --------------------

// message R { required uint32 m = 1; repeated sint64 xs = 2; }: integer coefficients, decoding basis, as they stand.
int64_t lolhip_r_write(uint32_t m, const int64_t* xs, int64_t n, uint8_t* out, int64_t cap) {
  if (n < 0 || (n > 0 && !xs)) return LOLHIP_ERR_INVALID;
  int64_t total = 1 + varint_len(m);
  for (int64_t j = 0; j < n; ++j) total += 1 + varint_len(zigzag(xs[j]));
  if (!out) return total;
  if (cap < total) return LOLHIP_ERR_INVALID;
  uint8_t* o = out;
  *o++ = 0x08; put_varint(o, m);
  for (int64_t j = 0; j < n; ++j) { *o++ = 0x10; put_varint(o, zigzag(xs[j])); }          // xs = 2, unpacked sint64
  return (int64_t)(o - out);
}

// message SecretKey { required R sk = 1; required double v = 2; }  (SHE.proto:9)
int64_t lolhip_secretkey_write(uint32_t m, double v, const int64_t* xs, int64_t n, uint8_t* out, int64_t cap) {
  const int64_t rl = lolhip_r_write(m, xs, n, nullptr, 0);
  if (rl < 0) return rl;
  const int64_t total = 1 + varint_len((uint64_t)rl) + rl + 1 + 8;
  if (!out) return total;
  if (cap < total) return LOLHIP_ERR_INVALID;
  uint8_t* o = out;
  *o++ = 0x0A; put_varint(o, (uint64_t)rl);
  o += lolhip_r_write(m, xs, n, o, rl);
  *o++ = 0x11;                                                                              // v = 2, 64-bit
  std::memcpy(o, &v, 8); o += 8;
  return (int64_t)(o - out);
}

// message LinearRq { required uint32 e = 1; required uint32 r = 2; repeated RqProduct coeffs = 3; }  (Lol.proto:11)
// xs [C][n][T]: the function's values on the relative decoding basis, decoding-basis residues of the output ring m.
int64_t lolhip_linearrq_write(uint32_t e, uint32_t r, uint32_t m, const int64_t* qs, int T, int C, const int64_t* xs, int64_t n,
                              uint8_t* out, int64_t cap) {
  if (C < 0 || (C > 0 && n > 0 && !xs)) return LOLHIP_ERR_INVALID;
  std::vector<int64_t> pl((size_t)C);
  int64_t total = 1 + varint_len(e) + 1 + varint_len(r);
  for (int c = 0; c < C; ++c) {
    const int64_t b = lolhip_rqproduct_write(m, qs, T, xs + (int64_t)c * n * T, n, nullptr, 0);
    if (b < 0) return b;
    pl[(size_t)c] = b;
    total += 1 + varint_len((uint64_t)b) + b;
  }
  if (!out) return total;
  if (cap < total) return LOLHIP_ERR_INVALID;
  uint8_t* o = out;
  *o++ = 0x08; put_varint(o, e);
  *o++ = 0x10; put_varint(o, r);
  for (int c = 0; c < C; ++c) {
    *o++ = 0x1A; put_varint(o, (uint64_t)pl[(size_t)c]);                                    // coeffs = 3
    const int64_t w = lolhip_rqproduct_write(m, qs, T, xs + (int64_t)c * n * T, n, o, pl[(size_t)c]);
    if (w < 0) return w;
    o += w;
  }
  return (int64_t)(o - out);
}

// message TunnelHint { LinearRq func = 1; repeated KSHint hint = 2; e = 3; r = 4; s = 5; p = 6; }  (SHE.proto:26),
// assembled from an encoded LinearRq and nh encoded KSHints (lolhip_linearrq_write / lolhip_kshint_write).
int64_t lolhip_tunnelhint_write(const uint8_t* func, int64_t func_len, const uint8_t* const* hints, const int64_t* hint_len, int nh,
                                uint32_t e, uint32_t r, uint32_t s, uint64_t p, uint8_t* out, int64_t cap) {
  if (!func || func_len < 0 || nh < 0 || (nh > 0 && (!hints || !hint_len))) return LOLHIP_ERR_INVALID;
  int64_t total = 1 + varint_len((uint64_t)func_len) + func_len + 1 + varint_len(e) + 1 + varint_len(r) + 1 + varint_len(s) + 1 + varint_len(p);
  for (int i = 0; i < nh; ++i) {
    if (!hints[i] || hint_len[i] < 0) return LOLHIP_ERR_INVALID;
    total += 1 + varint_len((uint64_t)hint_len[i]) + hint_len[i];
  }
  if (!out) return total;
  if (cap < total) return LOLHIP_ERR_INVALID;
  uint8_t* o = out;
  *o++ = 0x0A; put_varint(o, (uint64_t)func_len); std::memcpy(o, func, (size_t)func_len); o += func_len;
  for (int i = 0; i < nh; ++i) { *o++ = 0x12; put_varint(o, (uint64_t)hint_len[i]); std::memcpy(o, hints[i], (size_t)hint_len[i]); o += hint_len[i]; }
  *o++ = 0x18; put_varint(o, e);
  *o++ = 0x20; put_varint(o, r);
  *o++ = 0x28; put_varint(o, s);
  *o++ = 0x30; put_varint(o, p);
  return (int64_t)(o - out);
}

// lol-apps/HomomPRF.proto:18-26 — LinearFuncChain { repeated LinearRq funcs = 1 }, TunnelHintChain { repeated TunnelHint
// hints = 1 }, RoundHintChain { repeated KSHint hints = 1 }: one `repeated` sub-message field each.  read: the number of
// elements and the byte ranges (offset, length into buf) of up to cap of them, for the element readers above.
int64_t lolhip_chain_read(const uint8_t* buf, int64_t len, int64_t* off, int64_t* elem_len, int cap) {
  if (!buf || len < 0) return LOLHIP_ERR_INVALID;
  Reader rd{buf, buf + len};
  int64_t cnt = 0;
  while (rd.p < rd.end) {
    uint64_t key;
    if (!rd.varint(key)) return LOLHIP_ERR_INVALID;
    const uint32_t field = (uint32_t)(key >> 3), wt = (uint32_t)(key & 7);
    if (field == 1 && wt == 2) {
      const uint8_t* b; int64_t l;
      if (!sub(rd, b, l)) return LOLHIP_ERR_INVALID;
      if (cnt < cap) { if (off) off[cnt] = b - buf; if (elem_len) elem_len[cnt] = l; }
      ++cnt;
    } else if (!rd.skip(wt)) return LOLHIP_ERR_INVALID;
  }
  return cnt;
}
int64_t lolhip_chain_write(const uint8_t* const* elems, const int64_t* elem_len, int count, uint8_t* out, int64_t cap) {
  if (count < 0 || (count > 0 && (!elems || !elem_len))) return LOLHIP_ERR_INVALID;
  int64_t total = 0;
  for (int i = 0; i < count; ++i) {
    if (!elems[i] || elem_len[i] < 0) return LOLHIP_ERR_INVALID;
    total += 1 + varint_len((uint64_t)elem_len[i]) + elem_len[i];
  }
  if (!out) return total;
  if (cap < total) return LOLHIP_ERR_INVALID;
  uint8_t* o = out;
  for (int i = 0; i < count; ++i) { *o++ = 0x0A; put_varint(o, (uint64_t)elem_len[i]); std::memcpy(o, elems[i], (size_t)elem_len[i]); o += elem_len[i]; }
  return (int64_t)(o - out);
}

}  // extern "C"
