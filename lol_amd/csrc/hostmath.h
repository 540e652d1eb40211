// lol_amd/csrc/hostmath.h — host-side number theory for plan construction.
//
// Product code (linked into liblolhip.so).  Mirrors what Lol's Haskell side
// computes before it calls into C:
//   goodQs / principalRootUnity / mhatInv   lol/Crypto/Lol/Types/Unsafe/ZqBasic.hs:71-73,144-171
//   ppsFact / totientFact / valueHat         lol/Crypto/Lol/FactoredDefs.hs:360-361,428-445
//   ru / ruInv                               lol-cpp/Crypto/Lol/Cyclotomic/Tensor/CPP.hs:422-442
//   gCRTK / gInvCRTK                         lol/Crypto/Lol/Cyclotomic/Tensor.hs:290-337
//   index tables for twace/embed             lol/Crypto/Lol/Cyclotomic/Tensor.hs:390-509
#pragma once
#include <cstdint>
#include <utility>
#include <vector>

namespace lolhip {

typedef uint64_t u64;
typedef int64_t i64;
typedef unsigned __int128 u128;

struct PP { int p; int e; };

inline u64 mulmod(u64 a, u64 b, u64 q) { return (u64)(((u128)a * b) % q); }
u64 powmod(u64 b, u64 e, u64 q);
// b^-1 mod q, 0 when gcd(b,q) != 1
u64 invmod(u64 b, u64 q);
bool is_prime(u64 n);
// sorted distinct prime factors
std::vector<u64> prime_factors(u64 n);
std::vector<PP> factor_pps(u64 m);

i64 ipow(i64 b, int e);
i64 value_pps(const std::vector<PP>& pps);
i64 totient_pp(int p, int e);           // totient_pp(p,0) == 1
i64 totient_pps(const std::vector<PP>& pps);
i64 value_hat(i64 m);                   // m/2 if even else m
u64 odd_rad(const std::vector<PP>& pps);
i64 digit_rev(int p, int e, i64 j);

// first prime > lower congruent to 1 mod m (head of goodQs m lower)
u64 first_good_q(u64 m, u64 lower);
// smallest generator of Z_q^* for prime q
u64 smallest_generator(u64 q);
// omega_m = g0^((q-1)/m); returns 0 when q is not prime or m does not divide q-1
u64 principal_root(u64 m, u64 q);

// g vectors in the CRT basis, one modulus, length totient(m)
// omega_p[k] = primitive p_k-th root of unity to use (p_k = pps[k].p)
std::vector<u64> g_crt(const std::vector<PP>& pps, const std::vector<u64>& omega_p, u64 q, bool inverse);

// ---- ring-extension index tables (m | m') --------------------------------
struct MergedPP { int p, e, e2; };
// false if pps does not divide pps2
bool merge_pps(const std::vector<PP>& pps, const std::vector<PP>& pps2, std::vector<MergedPP>& out);
std::pair<i64, i64> to_index_pair(const std::vector<std::pair<i64, i64>>& tots, i64 i2);
i64 from_index_pair(const std::vector<std::pair<i64, i64>>& tots, i64 i1, i64 i0);

struct ExtTables {
  i64 phi = 0, phi2 = 0;
  std::vector<int32_t> twace_powdec;   // [phi]   extIndicesPowDec: out[i] = in[idx[i]]
  std::vector<int32_t> ext_crt;        // [phi2]  extIndicesCRT
  std::vector<int32_t> embed_pow;      // [phi2]  source index in O_m or -1 (zero)
  std::vector<int32_t> embed_dec;      // [phi2]  source index, -1 zero; bit 30 set => negate
  std::vector<int32_t> embed_crt;      // [phi2]  baseIndicesCRT
  std::vector<int32_t> coeffs;         // [phi2/phi][phi] extIndicesCoeffs: out[i1][i0] = in[idx[i1*phi + i0]]
};
static const int32_t EMBED_NEG_FLAG = 1 << 30;
bool build_ext_tables(const std::vector<PP>& pps, const std::vector<PP>& pps2, ExtTables& out);

}  // namespace lolhip
