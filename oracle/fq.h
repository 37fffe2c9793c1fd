// ORACLE (test infrastructure only — never linked into the product path).
// CPU restatement of libspartan's scalar field F_q, q = 2^252 + 27742317777372353535851937790883648493.
// Follows /root/reference/src/scalar/ristretto255.rs: 4x64-bit little-endian limbs, Montgomery form
// with R = 2^256, every value fully reduced to [0,q)  (ristretto255.rs:199,248-328,642-760).
#pragma once
#include <cstdint>
#include <cstring>

namespace orc {

typedef unsigned __int128 u128;

struct Fq {
  uint64_t l[4];
};

// ristretto255.rs:248-253
static const Fq FQ_MODULUS = {{0x5812631a5cf5d3edULL, 0x14def9dea2f79cd6ULL, 0x0000000000000000ULL, 0x1000000000000000ULL}};
// ristretto255.rs:304  INV = -(q^{-1} mod 2^64) mod 2^64
static const uint64_t FQ_INV = 0xd2b51da312547e1bULL;
// ristretto255.rs:307-328
static const Fq FQ_R = {{0xd6ec31748d98951dULL, 0xc6ef5bf4737dcf70ULL, 0xfffffffffffffffeULL, 0x0fffffffffffffffULL}};
static const Fq FQ_R2 = {{0xa40611e3449c0f01ULL, 0xd00e1ba768859347ULL, 0xceec73d217f5be65ULL, 0x0399411b7c309a3dULL}};
static const Fq FQ_R3 = {{0x2a9e49687b83a2dbULL, 0x278324e6aef7f3ecULL, 0x8065dc6c04ec5b65ULL, 0x0e530b773599cec7ULL}};

static inline Fq fq_zero() { return Fq{{0, 0, 0, 0}}; }
static inline Fq fq_one() { return FQ_R; }  // ristretto255.rs:370-372: one() is R
static inline bool fq_eq(const Fq& a, const Fq& b) { return memcmp(a.l, b.l, 32) == 0; }
static inline bool fq_is_zero(const Fq& a) { return (a.l[0] | a.l[1] | a.l[2] | a.l[3]) == 0; }

// ristretto255.rs:718-733 : a - b, add q back on borrow
static inline Fq fq_sub(const Fq& a, const Fq& b) {
  Fq d;
  u128 t;
  uint64_t borrow = 0;
  for (int i = 0; i < 4; i++) {
    t = (u128)a.l[i] - b.l[i] - borrow;
    d.l[i] = (uint64_t)t;
    borrow = (uint64_t)(t >> 64) & 1;
  }
  uint64_t mask = 0 - borrow;
  u128 c = 0;
  for (int i = 0; i < 4; i++) {
    c += (u128)d.l[i] + (FQ_MODULUS.l[i] & mask);
    d.l[i] = (uint64_t)c;
    c >>= 64;
  }
  return d;
}

// ristretto255.rs:736-745 : a + b then try to subtract q
static inline Fq fq_add(const Fq& a, const Fq& b) {
  Fq s;
  u128 c = 0;
  for (int i = 0; i < 4; i++) {
    c += (u128)a.l[i] + b.l[i];
    s.l[i] = (uint64_t)c;
    c >>= 64;
  }
  // a,b < q < 2^253 so no carry out of limb 3
  return fq_sub(s, FQ_MODULUS);
}

// ristretto255.rs:749-763
static inline Fq fq_neg(const Fq& a) {
  if (fq_is_zero(a)) return a;
  Fq z = fq_zero();
  return fq_sub(z, a);
}

// ristretto255.rs:642-686 : Montgomery reduction (HAC 14.32) of a 512-bit value r[0..8)
static inline Fq fq_mont_reduce(const uint64_t rin[8]) {
  uint64_t r[9];
  for (int i = 0; i < 8; i++) r[i] = rin[i];
  r[8] = 0;
  uint64_t carry2 = 0;
  for (int i = 0; i < 4; i++) {
    uint64_t k = r[i] * FQ_INV;
    u128 c = 0;
    for (int j = 0; j < 4; j++) {
      c += (u128)k * FQ_MODULUS.l[j] + r[i + j];
      r[i + j] = (uint64_t)c;
      c >>= 64;
    }
    c += (u128)r[i + 4] + carry2;
    r[i + 4] = (uint64_t)c;
    carry2 = (uint64_t)(c >> 64);
  }
  Fq t = {{r[4], r[5], r[6], r[7]}};
  // "Result may be within MODULUS of the correct value" (ristretto255.rs:684-685).
  // carry2 can only be set when the input exceeded R*q, which the callers never produce.
  return fq_sub(t, FQ_MODULUS);
}

// ristretto255.rs:690-714 : schoolbook 4x4 then Montgomery reduce
static inline Fq fq_mul(const Fq& a, const Fq& b) {
  uint64_t r[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 4; i++) {
    u128 c = 0;
    for (int j = 0; j < 4; j++) {
      c += (u128)a.l[i] * b.l[j] + r[i + j];
      r[i + j] = (uint64_t)c;
      c >>= 64;
    }
    r[i + 4] = (uint64_t)c;
  }
  return fq_mont_reduce(r);
}
static inline Fq fq_sqr(const Fq& a) { return fq_mul(a, a); }  // ristretto255.rs:476-504 (same value)

// ristretto255.rs:222-226  From<u64>
static inline Fq fq_from_u64(uint64_t v) {
  Fq t = {{v, 0, 0, 0}};
  return fq_mul(t, FQ_R2);
}

// ristretto255.rs:448-466 : 512-bit little-endian integer -> Fq  (d0*R2 + d1*R3)
static inline Fq fq_from_u512(const uint64_t limbs[8]) {
  Fq d0 = {{limbs[0], limbs[1], limbs[2], limbs[3]}};
  Fq d1 = {{limbs[4], limbs[5], limbs[6], limbs[7]}};
  return fq_add(fq_mul(d0, FQ_R2), fq_mul(d1, FQ_R3));
}
// ristretto255.rs:435-446
static inline Fq fq_from_bytes_wide(const uint8_t b[64]) {
  uint64_t limbs[8];
  memcpy(limbs, b, 64);  // little-endian host
  return fq_from_u512(limbs);
}
// ristretto255.rs:419-431 : canonical little-endian bytes
static inline void fq_to_bytes(const Fq& a, uint8_t out[32]) {
  uint64_t r[8] = {a.l[0], a.l[1], a.l[2], a.l[3], 0, 0, 0, 0};
  Fq t = fq_mont_reduce(r);
  memcpy(out, t.l, 32);
}
// ristretto255.rs:390-416 : returns false when the encoding is not canonical (>= q)
static inline bool fq_from_bytes(const uint8_t b[32], Fq* out) {
  Fq t;
  memcpy(t.l, b, 32);
  uint64_t borrow = 0;
  for (int i = 0; i < 4; i++) {
    u128 d = (u128)t.l[i] - FQ_MODULUS.l[i] - borrow;
    borrow = (uint64_t)(d >> 64) & 1;
  }
  *out = fq_mul(t, FQ_R2);
  return borrow == 1;
}

// ristretto255.rs:523-536 pow_vartime (square-and-multiply, MSB first)
static inline Fq fq_pow(const Fq& a, const uint64_t e[4]) {
  Fq res = fq_one();
  for (int w = 3; w >= 0; w--)
    for (int i = 63; i >= 0; i--) {
      res = fq_sqr(res);
      if ((e[w] >> i) & 1) res = fq_mul(res, a);
    }
  return res;
}
// ristretto255.rs:541-595 computes a^(q-2) with an addition chain; the value is a^(q-2) (test :1130-1172)
static inline Fq fq_invert(const Fq& a) {
  uint64_t e[4] = {FQ_MODULUS.l[0] - 2, FQ_MODULUS.l[1], FQ_MODULUS.l[2], FQ_MODULUS.l[3]};
  return fq_pow(a, e);
}
// ristretto255.rs:597-640
static inline Fq fq_batch_invert(Fq* v, size_t n) {
  Fq* scratch = new Fq[n];
  Fq acc = fq_one();
  for (size_t i = 0; i < n; i++) {
    scratch[i] = acc;
    acc = fq_mul(acc, v[i]);
  }
  acc = fq_invert(acc);
  Fq ret = acc;
  for (size_t i = n; i-- > 0;) {
    Fq tmp = fq_mul(acc, v[i]);
    v[i] = fq_mul(acc, scratch[i]);
    acc = tmp;
  }
  delete[] scratch;
  return ret;
}

// operator sugar so the protocol restatement reads like the reference
static inline Fq operator+(const Fq& a, const Fq& b) { return fq_add(a, b); }
static inline Fq operator-(const Fq& a, const Fq& b) { return fq_sub(a, b); }
static inline Fq operator*(const Fq& a, const Fq& b) { return fq_mul(a, b); }
static inline Fq operator-(const Fq& a) { return fq_neg(a); }
static inline bool operator==(const Fq& a, const Fq& b) { return fq_eq(a, b); }
static inline bool operator!=(const Fq& a, const Fq& b) { return !fq_eq(a, b); }
static inline Fq& operator+=(Fq& a, const Fq& b) { a = fq_add(a, b); return a; }
static inline Fq& operator-=(Fq& a, const Fq& b) { a = fq_sub(a, b); return a; }
static inline Fq& operator*=(Fq& a, const Fq& b) { a = fq_mul(a, b); return a; }

}  // namespace orc
