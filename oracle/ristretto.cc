// ORACLE (test infrastructure only). See ristretto.h for provenance.
#include "ristretto.h"

#include <cstdio>

namespace orc {

const RistConsts& rist_consts() {
  // RFC 9496 §4.1 constants, little-endian hex (derivation re-checked in tests/test_oracle_group.py).
  static const RistConsts K = {
      fp_from_hex_le("a3785913ca4deb75abd841414d0a700098e879777940c78c73fe6f2bee6c0352"),  // D = -121665/121666
      fp_from_hex_le("59f1b226949bd6eb56b183829a14e00030d1f3eef2808e19e7fcdf56dcd90624"),  // 2D
      fp_from_hex_le("b0a00e4a271beec478e42fad0618432fa7d7fb3d99004d2b0bdfc14f8024832b"),  // SQRT_M1
      fp_from_hex_le("1b2e7b49a0f6977ebd54781b0c8e9daffdd1f531c9fc3c0fac48832bbf316937"),  // SQRT_AD_MINUS_ONE
      fp_from_hex_le("ea405d80aafdc899be72415a17162f9d40d801fe917bc216a2fcafcf05896c78"),  // INVSQRT_A_MINUS_D
      fp_from_hex_le("76c15f94c1097ce20f355ecd38a1812ce4df70beddab9499d7e0b3b2a8729002"),  // ONE_MINUS_D_SQ
      fp_from_hex_le("204ded44aa5aad3199191eb02c4a9ed2eb4e9b522fd3dc4c41226cf67ab36859"),  // D_MINUS_ONE_SQ
  };
  return K;
}

// curve25519_dalek::constants::RISTRETTO_BASEPOINT_COMPRESSED (group.rs:23-24)
static const uint8_t BASEPOINT_COMPRESSED[32] = {0xe2, 0xf2, 0xae, 0x0a, 0x6a, 0xbc, 0x4e, 0x71, 0xa8, 0x84, 0xa9,
                                                 0x61, 0xc5, 0x00, 0x51, 0x5f, 0x58, 0xe3, 0x0b, 0x6a, 0xa5, 0x82,
                                                 0xdd, 0x8d, 0xb6, 0xa6, 0x59, 0x45, 0xe0, 0x8d, 0x2d, 0x76};
void pt_basepoint_compressed(uint8_t out[32]) { memcpy(out, BASEPOINT_COMPRESSED, 32); }
const Pt& pt_basepoint() {
  static Pt B;
  static bool init = false;
  if (!init) {
    bool ok = pt_decompress(BASEPOINT_COMPRESSED, &B);
    if (!ok) {
      fprintf(stderr, "oracle: basepoint failed to decode\n");
      abort();
    }
    init = true;
  }
  return B;
}

// canonical (non-Montgomery) little-endian bytes of a scalar — scalar/mod.rs:32-36 decompress_scalar
static inline void scalar_bytes(const Fq& s, uint8_t out[32]) { fq_to_bytes(s, out); }

Pt pt_mul(const Fq& s, const Pt& p) {
  uint8_t b[32];
  scalar_bytes(s, b);
  // fixed 4-bit windows, MSB first
  Pt tab[16];
  tab[0] = pt_identity();
  for (int i = 1; i < 16; i++) tab[i] = pt_add(tab[i - 1], p);
  Pt acc = pt_identity();
  for (int i = 63; i >= 0; i--) {
    for (int k = 0; k < 4; k++) acc = pt_dbl(acc);
    int nib = (b[i / 2] >> ((i & 1) * 4)) & 15;
    if (nib) acc = pt_add(acc, tab[nib]);
  }
  return acc;
}

// Straus (shared doublings, 4-bit windows) for small n; Pippenger bucket method otherwise.
// dalek uses Straus below 190 points and Pippenger above (upstream behaviour, irrelevant to the result).
static Pt msm_straus(const Fq* scalars, const Pt* points, size_t n) {
  std::vector<Pt> tab(n * 16);
  std::vector<uint8_t> sb(n * 32);
  for (size_t k = 0; k < n; k++) {
    scalar_bytes(scalars[k], &sb[k * 32]);
    tab[k * 16] = pt_identity();
    for (int i = 1; i < 16; i++) tab[k * 16 + i] = pt_add(tab[k * 16 + i - 1], points[k]);
  }
  Pt acc = pt_identity();
  for (int i = 63; i >= 0; i--) {
    for (int k = 0; k < 4; k++) acc = pt_dbl(acc);
    for (size_t k = 0; k < n; k++) {
      int nib = (sb[k * 32 + i / 2] >> ((i & 1) * 4)) & 15;
      if (nib) acc = pt_add(acc, tab[k * 16 + nib]);
    }
  }
  return acc;
}

static Pt msm_pippenger(const Fq* scalars, const Pt* points, size_t n) {
  int c = n < 500 ? 6 : (n < 800 ? 7 : 8);
  if (n >= 1 << 15) c = 10;
  std::vector<uint8_t> sb(n * 32 + 8, 0);
  for (size_t k = 0; k < n; k++) scalar_bytes(scalars[k], &sb[k * 32]);
  int nwin = (253 + c - 1) / c;
  size_t nb = ((size_t)1 << c) - 1;
  std::vector<Pt> buckets(nb);
  std::vector<uint8_t> used(nb);
  Pt total = pt_identity();
  for (int w = nwin - 1; w >= 0; w--) {
    for (int k = 0; k < c; k++) total = pt_dbl(total);
    std::fill(used.begin(), used.end(), 0);
    int bit = w * c;
    for (size_t k = 0; k < n; k++) {
      const uint8_t* s = &sb[k * 32];
      uint64_t word = 0;
      int byte = bit / 8;
      for (int t = 0; t < 8 && byte + t < 32; t++) word |= (uint64_t)s[byte + t] << (8 * t);
      size_t dig = (size_t)((word >> (bit % 8)) & nb);
      if (!dig) continue;
      if (used[dig - 1]) buckets[dig - 1] = pt_add(buckets[dig - 1], points[k]);
      else { buckets[dig - 1] = points[k]; used[dig - 1] = 1; }
    }
    // sum_{d} d * bucket[d] by running sums
    Pt run = pt_identity(), sum = pt_identity();
    bool any = false;
    for (size_t d = nb; d-- > 0;) {
      if (used[d]) { run = pt_add(run, buckets[d]); any = true; }
      if (any) sum = pt_add(sum, run);
    }
    total = pt_add(total, sum);
  }
  return total;
}

Pt pt_msm(const Fq* scalars, const Pt* points, size_t n) {
  if (n == 0) return pt_identity();
  if (n < 190) return msm_straus(scalars, points, n);
  return msm_pippenger(scalars, points, n);
}

}  // namespace orc
