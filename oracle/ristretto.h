// ORACLE (test infrastructure only — never linked into the product path).
// CPU restatement of the group libspartan uses: ristretto255 over edwards25519.
// The reference takes this from the third-party crate curve25519-dalek ^4.1.1 (Cargo.toml:14-18), which is
// NOT present under /root/reference; call sites: src/group.rs:6-7,18,28-45,108-115, src/commitments.rs:25.
// Restated from the published algorithm (RFC 9496 "The ristretto255 and decaf448 Groups", §4) and pinned
// against RFC 9496 appendix vectors + libsodium 1.0.18 (tests/test_oracle_group.py, tests/golden/).
// Field: F_p, p = 2^255-19, 5 x 51-bit limbs. Points: extended twisted Edwards (X:Y:Z:T), a = -1.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "fq.h"

namespace orc {

struct Fp {
  uint64_t v[5];
};
static const uint64_t FP_M51 = ((uint64_t)1 << 51) - 1;

static inline Fp fp_zero() { return Fp{{0, 0, 0, 0, 0}}; }
static inline Fp fp_one() { return Fp{{1, 0, 0, 0, 0}}; }
static inline Fp fp_add(const Fp& a, const Fp& b) {
  Fp r;
  for (int i = 0; i < 5; i++) r.v[i] = a.v[i] + b.v[i];
  return r;
}
// a - b with a bias of 2p (limbs stay positive; inputs must have limbs < 2^52)
static inline Fp fp_sub(const Fp& a, const Fp& b) {
  Fp r;
  r.v[0] = a.v[0] + 0xfffffffffffdaULL - b.v[0];
  for (int i = 1; i < 5; i++) r.v[i] = a.v[i] + 0xffffffffffffeULL - b.v[i];
  return r;
}
static inline Fp fp_carry(const Fp& a) {
  Fp r = a;
  uint64_t c;
  for (int rep = 0; rep < 2; rep++) {
    c = r.v[0] >> 51; r.v[0] &= FP_M51; r.v[1] += c;
    c = r.v[1] >> 51; r.v[1] &= FP_M51; r.v[2] += c;
    c = r.v[2] >> 51; r.v[2] &= FP_M51; r.v[3] += c;
    c = r.v[3] >> 51; r.v[3] &= FP_M51; r.v[4] += c;
    c = r.v[4] >> 51; r.v[4] &= FP_M51; r.v[0] += c * 19;
  }
  return r;
}
static inline Fp fp_mul(const Fp& a, const Fp& b) {
  u128 h[5];
  uint64_t b19[5];
  for (int i = 0; i < 5; i++) b19[i] = b.v[i] * 19;
  for (int k = 0; k < 5; k++) {
    u128 s = 0;
    for (int i = 0; i < 5; i++) {
      int j = k - i;
      s += (j >= 0) ? (u128)a.v[i] * b.v[j] : (u128)a.v[i] * b19[j + 5];
    }
    h[k] = s;
  }
  Fp o;
  uint64_t c = 0;
  for (int i = 0; i < 5; i++) {
    h[i] += c;
    o.v[i] = (uint64_t)h[i] & FP_M51;
    c = (uint64_t)(h[i] >> 51);
  }
  o.v[0] += c * 19;
  c = o.v[0] >> 51; o.v[0] &= FP_M51; o.v[1] += c;
  return o;
}
static inline Fp fp_sqr(const Fp& a) { return fp_mul(a, a); }
static inline Fp fp_neg(const Fp& a) { return fp_carry(fp_sub(fp_zero(), a)); }
static inline Fp fp_sub_c(const Fp& a, const Fp& b) { return fp_carry(fp_sub(a, b)); }
static inline Fp fp_add_c(const Fp& a, const Fp& b) { return fp_carry(fp_add(a, b)); }

static inline void fp_to_bytes(const Fp& a, uint8_t out[32]) {
  Fp t = fp_carry(a);
  // t < 2^255 + small ; compute t + 19 and look at bit 255 to decide whether t >= p
  uint64_t q = (t.v[0] + 19) >> 51;
  q = (t.v[1] + q) >> 51; q = (t.v[2] + q) >> 51; q = (t.v[3] + q) >> 51; q = (t.v[4] + q) >> 51;
  t.v[0] += 19 * q;
  uint64_t c;
  c = t.v[0] >> 51; t.v[0] &= FP_M51; t.v[1] += c;
  c = t.v[1] >> 51; t.v[1] &= FP_M51; t.v[2] += c;
  c = t.v[2] >> 51; t.v[2] &= FP_M51; t.v[3] += c;
  c = t.v[3] >> 51; t.v[3] &= FP_M51; t.v[4] += c;
  t.v[4] &= FP_M51;
  uint64_t w[4];
  w[0] = t.v[0] | (t.v[1] << 51);
  w[1] = (t.v[1] >> 13) | (t.v[2] << 38);
  w[2] = (t.v[2] >> 26) | (t.v[3] << 25);
  w[3] = (t.v[3] >> 39) | (t.v[4] << 12);
  memcpy(out, w, 32);
}
// little-endian bytes, top bit ignored (value mod 2^255), not necessarily canonical
static inline Fp fp_from_bytes(const uint8_t b[32]) {
  uint64_t w[4];
  memcpy(w, b, 32);
  Fp r;
  r.v[0] = w[0] & FP_M51;
  r.v[1] = ((w[0] >> 51) | (w[1] << 13)) & FP_M51;
  r.v[2] = ((w[1] >> 38) | (w[2] << 26)) & FP_M51;
  r.v[3] = ((w[2] >> 25) | (w[3] << 39)) & FP_M51;
  r.v[4] = (w[3] >> 12) & FP_M51;
  return r;
}
static inline bool fp_eq(const Fp& a, const Fp& b) {
  uint8_t x[32], y[32];
  fp_to_bytes(a, x); fp_to_bytes(b, y);
  return memcmp(x, y, 32) == 0;
}
static inline bool fp_is_zero(const Fp& a) { return fp_eq(a, fp_zero()); }
static inline bool fp_is_negative(const Fp& a) {  // RFC 9496 §4.1: low bit of the canonical encoding
  uint8_t x[32];
  fp_to_bytes(a, x);
  return x[0] & 1;
}
static inline Fp fp_abs(const Fp& a) { return fp_is_negative(a) ? fp_neg(a) : a; }
static inline Fp fp_pow2k(Fp a, int k) {
  for (int i = 0; i < k; i++) a = fp_sqr(a);
  return a;
}
// a^(2^250-1) ladder shared by invert and pow_p58
static inline void fp_pow_ladder(const Fp& z, Fp* z2_250_0, Fp* z11) {
  Fp z2 = fp_sqr(z);
  Fp z9 = fp_mul(fp_pow2k(z2, 2), z);
  *z11 = fp_mul(z9, z2);
  Fp z2_5_0 = fp_mul(fp_sqr(*z11), z9);
  Fp z2_10_0 = fp_mul(fp_pow2k(z2_5_0, 5), z2_5_0);
  Fp z2_20_0 = fp_mul(fp_pow2k(z2_10_0, 10), z2_10_0);
  Fp z2_40_0 = fp_mul(fp_pow2k(z2_20_0, 20), z2_20_0);
  Fp z2_50_0 = fp_mul(fp_pow2k(z2_40_0, 10), z2_10_0);
  Fp z2_100_0 = fp_mul(fp_pow2k(z2_50_0, 50), z2_50_0);
  Fp z2_200_0 = fp_mul(fp_pow2k(z2_100_0, 100), z2_100_0);
  *z2_250_0 = fp_mul(fp_pow2k(z2_200_0, 50), z2_50_0);
}
static inline Fp fp_invert(const Fp& z) {  // z^(p-2) = z^(2^255-21)
  Fp t, z11;
  fp_pow_ladder(z, &t, &z11);
  return fp_mul(fp_pow2k(t, 5), z11);
}
static inline Fp fp_pow_p58(const Fp& z) {  // z^((p-5)/8) = z^(2^252-3)
  Fp t, z11;
  fp_pow_ladder(z, &t, &z11);
  return fp_mul(fp_pow2k(t, 2), z);
}

// --- curve / ristretto constants (RFC 9496 §4.1), little-endian bytes ---
struct RistConsts {
  Fp D, D2, SQRT_M1, SQRT_AD_MINUS_ONE, INVSQRT_A_MINUS_D, ONE_MINUS_D_SQ, D_MINUS_ONE_SQ;
};
static inline Fp fp_from_hex_le(const char* hex) {
  uint8_t b[32];
  for (int i = 0; i < 32; i++) {
    unsigned x;
    sscanf(hex + 2 * i, "%2x", &x);
    b[i] = (uint8_t)x;
  }
  return fp_from_bytes(b);
}
const RistConsts& rist_consts();  // defined in ristretto.cc (values checked algebraically in tests)

// SQRT_RATIO_M1 (RFC 9496 §4.2)
static inline bool fp_sqrt_ratio_m1(const Fp& u, const Fp& v, Fp* out) {
  const RistConsts& K = rist_consts();
  Fp v3 = fp_mul(fp_sqr(v), v);
  Fp v7 = fp_mul(fp_sqr(v3), v);
  Fp r = fp_mul(fp_mul(u, v3), fp_pow_p58(fp_mul(u, v7)));
  Fp check = fp_mul(v, fp_sqr(r));
  Fp neg_u = fp_neg(u);
  bool correct_sign = fp_eq(check, u);
  bool flipped = fp_eq(check, neg_u);
  bool flipped_i = fp_eq(check, fp_mul(neg_u, K.SQRT_M1));
  if (flipped || flipped_i) r = fp_mul(r, K.SQRT_M1);
  *out = fp_abs(r);
  return correct_sign || flipped;
}

struct Pt {  // extended coordinates, x = X/Z, y = Y/Z, T = XY/Z
  Fp X, Y, Z, T;
};
static inline Pt pt_identity() { return Pt{fp_zero(), fp_one(), fp_one(), fp_zero()}; }

// unified addition (add-2008-hwcd-3 with a = -1, k = 2d)
static inline Pt pt_add(const Pt& p, const Pt& q) {
  const RistConsts& K = rist_consts();
  Fp A = fp_mul(fp_sub_c(p.Y, p.X), fp_sub_c(q.Y, q.X));
  Fp B = fp_mul(fp_add_c(p.Y, p.X), fp_add_c(q.Y, q.X));
  Fp C = fp_mul(fp_mul(p.T, K.D2), q.T);
  Fp Dd = fp_mul(fp_add_c(p.Z, p.Z), q.Z);
  Fp E = fp_sub_c(B, A), F = fp_sub_c(Dd, C), G = fp_add_c(Dd, C), H = fp_add_c(B, A);
  return Pt{fp_mul(E, F), fp_mul(G, H), fp_mul(F, G), fp_mul(E, H)};
}
static inline Pt pt_neg(const Pt& p) { return Pt{fp_neg(p.X), p.Y, p.Z, fp_neg(p.T)}; }
static inline Pt pt_sub(const Pt& p, const Pt& q) { return pt_add(p, pt_neg(q)); }
// dbl-2008-hwcd, a = -1
static inline Pt pt_dbl(const Pt& p) {
  Fp A = fp_sqr(p.X), B = fp_sqr(p.Y);
  Fp C = fp_add_c(fp_sqr(p.Z), fp_sqr(p.Z));
  Fp Dd = fp_neg(A);
  Fp E = fp_sub_c(fp_sub_c(fp_sqr(fp_add_c(p.X, p.Y)), A), B);
  Fp G = fp_add_c(Dd, B), F = fp_sub_c(G, C), H = fp_sub_c(Dd, B);
  return Pt{fp_mul(E, F), fp_mul(G, H), fp_mul(F, G), fp_mul(E, H)};
}

// RFC 9496 §4.3.2 Encode
static inline void pt_compress(const Pt& p, uint8_t out[32]) {
  const RistConsts& K = rist_consts();
  Fp u1 = fp_mul(fp_add_c(p.Z, p.Y), fp_sub_c(p.Z, p.Y));
  Fp u2 = fp_mul(p.X, p.Y);
  Fp invsqrt;
  fp_sqrt_ratio_m1(fp_one(), fp_mul(u1, fp_sqr(u2)), &invsqrt);
  Fp den1 = fp_mul(invsqrt, u1), den2 = fp_mul(invsqrt, u2);
  Fp z_inv = fp_mul(fp_mul(den1, den2), p.T);
  Fp ix0 = fp_mul(p.X, K.SQRT_M1), iy0 = fp_mul(p.Y, K.SQRT_M1);
  Fp ench = fp_mul(den1, K.INVSQRT_A_MINUS_D);
  bool rotate = fp_is_negative(fp_mul(p.T, z_inv));
  Fp x = rotate ? iy0 : p.X;
  Fp y = rotate ? ix0 : p.Y;
  Fp den_inv = rotate ? ench : den2;
  if (fp_is_negative(fp_mul(x, z_inv))) y = fp_neg(y);
  Fp s = fp_abs(fp_mul(den_inv, fp_sub_c(p.Z, y)));
  fp_to_bytes(s, out);
}
// RFC 9496 §4.3.1 Decode; returns false on invalid encodings
static inline bool pt_decompress(const uint8_t in[32], Pt* out) {
  const RistConsts& K = rist_consts();
  Fp s = fp_from_bytes(in);
  uint8_t chk[32];
  fp_to_bytes(s, chk);
  if (memcmp(chk, in, 32) != 0) return false;  // non-canonical (also rejects the top bit)
  if (in[0] & 1) return false;                 // negative
  Fp ss = fp_sqr(s);
  Fp u1 = fp_sub_c(fp_one(), ss), u2 = fp_add_c(fp_one(), ss);
  Fp u2s = fp_sqr(u2);
  Fp v = fp_sub_c(fp_neg(fp_mul(K.D, fp_sqr(u1))), u2s);
  Fp invsqrt;
  bool was_square = fp_sqrt_ratio_m1(fp_one(), fp_mul(v, u2s), &invsqrt);
  Fp den_x = fp_mul(invsqrt, u2);
  Fp den_y = fp_mul(fp_mul(invsqrt, den_x), v);
  Fp x = fp_abs(fp_mul(fp_add_c(s, s), den_x));
  Fp y = fp_mul(u1, den_y);
  Fp t = fp_mul(x, y);
  if (!was_square || fp_is_negative(t) || fp_is_zero(y)) return false;
  *out = Pt{x, y, fp_one(), t};
  return true;
}
// RFC 9496 §4.3.4 MAP (Elligator)
static inline Pt pt_elligator(const Fp& t) {
  const RistConsts& K = rist_consts();
  Fp one = fp_one();
  Fp r = fp_mul(K.SQRT_M1, fp_sqr(t));
  Fp u = fp_mul(fp_add_c(r, one), K.ONE_MINUS_D_SQ);
  Fp v = fp_mul(fp_sub_c(fp_neg(one), fp_mul(r, K.D)), fp_add_c(r, K.D));
  Fp s;
  bool was_square = fp_sqrt_ratio_m1(u, v, &s);
  Fp s_prime = fp_neg(fp_abs(fp_mul(s, t)));
  if (!was_square) s = s_prime;
  Fp c = was_square ? fp_neg(one) : r;
  Fp N = fp_sub_c(fp_mul(fp_mul(c, fp_sub_c(r, one)), K.D_MINUS_ONE_SQ), v);
  Fp w0 = fp_mul(fp_add_c(s, s), v);
  Fp w1 = fp_mul(N, K.SQRT_AD_MINUS_ONE);
  Fp w2 = fp_sub_c(one, fp_sqr(s));
  Fp w3 = fp_add_c(one, fp_sqr(s));
  return Pt{fp_mul(w0, w3), fp_mul(w2, w1), fp_mul(w1, w3), fp_mul(w0, w2)};
}
// dalek RistrettoPoint::from_uniform_bytes == RFC 9496 §4.3.4 one-way map (commitments.rs:25)
static inline Pt pt_from_uniform_bytes(const uint8_t b[64]) {
  Fp t1 = fp_from_bytes(b), t2 = fp_from_bytes(b + 32);  // top bit of each half is masked
  return pt_add(pt_elligator(t1), pt_elligator(t2));
}
// ristretto equality (RFC 9496 §4.3.3): x1*y2 == y1*x2 or y1*y2 == x1*x2
static inline bool pt_eq(const Pt& a, const Pt& b) {
  return fp_eq(fp_mul(a.X, b.Y), fp_mul(a.Y, b.X)) || fp_eq(fp_mul(a.Y, b.Y), fp_mul(a.X, b.X));
}

// Scalar * point and multi-scalar multiplication. The reference funnels every MSM through
// GroupElement::vartime_multiscalar_mul (group.rs:98-117) after converting each Scalar out of Montgomery
// form (scalar/mod.rs:32-36). Any correct MSM yields the same group element.
Pt pt_mul(const Fq& s, const Pt& p);                               // group.rs:26-46
Pt pt_msm(const Fq* scalars, const Pt* points, size_t n);          // group.rs:98-117
const Pt& pt_basepoint();
void pt_basepoint_compressed(uint8_t out[32]);                     // group.rs:23-24

}  // namespace orc
