// spartan_amd: radix-2^25.5 arithmetic mod p = 2^255-19 for SERIAL chains executed by a lone wavefront.
//
// A single wave issues roughly one instruction every 5 cycles whatever its type, so the time of the LDS tree
// additions and of the 254-squaring inverse-square-root ladder inside every ristretto encode (RFC 9496 §4.3.2;
// one per commitment, ~140 of them strictly sequential per proof) is their instruction count. With ten signed
// limbs of 26/25 bits (the classic "ref10" layout) a partial product plus accumulation is ONE v_mad_i64_i32,
// there are ten independent accumulators (no dependent stalls), additions and subtractions are ten plain 32-bit
// ops with no carries, and reduction is a shift/mask carry chain: ~130 VALU instructions per squaring, against
// ~360 scalar instructions when the compiler scalarises the 4x64 form of field.hpp (it does, for wave-uniform
// data) or ~250 vector ones. Values are identical to field.hpp's; tests/test_host_arith.py checks every function
// against Python integers, including the limb-bound edge cases.
//
// Bounds (as in ref10): mul/sqr outputs have |even limb| <= 1.01*2^25, |odd limb| <= 1.01*2^24; ONE level of
// add/sub of such values (<= 1.1*2^26 / 2^25) is a valid mul/sqr input. Every formula below respects that.
#pragma once
#include "field.hpp"

namespace sp {

struct Fe10 {
  int32_t v[10];  // limb i has weight 2^ceil(25.5 i)
};

// keep the limbs in vector registers: inline-asm outputs are treated as divergent, which stops the compiler from
// moving a wave-uniform chain to the scalar ALU (no 64-bit multiply-add there).
SP_HD void fe10_pin(Fe10& a) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
  for (int i = 0; i < 10; i++) asm volatile("" : "+v"(a.v[i]));
#else
  (void)a;
#endif
}

SP_HD Fe10 fe10_from_fp(const Fp& x) {
  Fp c = fp_canon(x);  // < 2^255: limbs come out in [0,2^26) / [0,2^25)
  const int off[10] = {0, 26, 51, 77, 102, 128, 153, 179, 204, 230};
  Fe10 r;
#pragma unroll
  for (int i = 0; i < 10; i++) {
    int w = off[i] >> 6, s = off[i] & 63, bits = (i & 1) ? 25 : 26;
    uint64_t lo = c.v[w] >> s;
    if (s + bits > 64 && w < 3) lo |= c.v[w + 1] << (64 - s);
    r.v[i] = (int32_t)(lo & (((uint64_t)1 << bits) - 1));
  }
  return r;
}
// canonical value (ref10 fe_tobytes): works for any limbs within the bounds above
SP_HD Fp fe10_to_fp(const Fe10& a) {
  int32_t h[10];
#pragma unroll
  for (int i = 0; i < 10; i++) h[i] = a.v[i];
  int32_t q = (19 * h[9] + (1 << 24)) >> 25;
#pragma unroll
  for (int i = 0; i < 10; i++) q = (h[i] + q) >> ((i & 1) ? 25 : 26);
  h[0] += 19 * q;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    int sh = (i & 1) ? 25 : 26;
    int32_t c = h[i] >> sh;
    h[i + 1] += c;
    h[i] -= c << sh;
  }
  h[9] -= (h[9] >> 25) << 25;
  const int off[10] = {0, 26, 51, 77, 102, 128, 153, 179, 204, 230};
  Fp r = fp_zero();
#pragma unroll
  for (int i = 0; i < 10; i++) {
    int w = off[i] >> 6, s = off[i] & 63;
    uint64_t x = (uint64_t)(uint32_t)h[i];
    r.v[w] |= x << s;
    if (s > 38 && w < 3) r.v[w + 1] |= x >> (64 - s);  // a 26-bit limb straddles the word when s > 38
  }
  return r;
}

SP_HD Fe10 fe10_add(const Fe10& a, const Fe10& b) {
  Fe10 r;
#pragma unroll
  for (int i = 0; i < 10; i++) r.v[i] = a.v[i] + b.v[i];
  return r;
}
SP_HD Fe10 fe10_sub(const Fe10& a, const Fe10& b) {
  Fe10 r;
#pragma unroll
  for (int i = 0; i < 10; i++) r.v[i] = a.v[i] - b.v[i];
  return r;
}
SP_HD Fe10 fe10_neg(const Fe10& a) {
  Fe10 r;
#pragma unroll
  for (int i = 0; i < 10; i++) r.v[i] = -a.v[i];
  return r;
}
SP_HD Fe10 fe10_select(const Fe10& a, const Fe10& b, bool take_b) {
  Fe10 r;
#pragma unroll
  for (int i = 0; i < 10; i++) r.v[i] = take_b ? b.v[i] : a.v[i];
  return r;
}
SP_HD Fe10 fe10_one() { return Fe10{{1, 0, 0, 0, 0, 0, 0, 0, 0, 0}}; }

SP_HD Fe10 fe10_carry(int64_t h[10]) {  // ref10 fe_mul / fe_sq tail: two interleaved rounding carry chains
  int64_t c;
  c = (h[0] + (1 << 25)) >> 26; h[1] += c; h[0] -= c << 26;
  c = (h[4] + (1 << 25)) >> 26; h[5] += c; h[4] -= c << 26;
  c = (h[1] + (1 << 24)) >> 25; h[2] += c; h[1] -= c << 25;
  c = (h[5] + (1 << 24)) >> 25; h[6] += c; h[5] -= c << 25;
  c = (h[2] + (1 << 25)) >> 26; h[3] += c; h[2] -= c << 26;
  c = (h[6] + (1 << 25)) >> 26; h[7] += c; h[6] -= c << 26;
  c = (h[3] + (1 << 24)) >> 25; h[4] += c; h[3] -= c << 25;
  c = (h[7] + (1 << 24)) >> 25; h[8] += c; h[7] -= c << 25;
  c = (h[4] + (1 << 25)) >> 26; h[5] += c; h[4] -= c << 26;
  c = (h[8] + (1 << 25)) >> 26; h[9] += c; h[8] -= c << 26;
  c = (h[9] + (1 << 24)) >> 25; h[0] += c * 19; h[9] -= c << 25;
  c = (h[0] + (1 << 25)) >> 26; h[1] += c; h[0] -= c << 26;
  Fe10 r;
#pragma unroll
  for (int i = 0; i < 10; i++) r.v[i] = (int32_t)h[i];
  return r;
}
// conversion with limbs centred (|limb| <= 2^25 / 2^24) so the result is a valid operand of add/sub followed by mul
SP_HD Fe10 fe10_load(const Fp& x) {
  Fe10 t = fe10_from_fp(x);
  int64_t h[10];
#pragma unroll
  for (int i = 0; i < 10; i++) h[i] = t.v[i];
  return fe10_carry(h);
}
SP_HD Fe10 fe10_mul(const Fe10& f, const Fe10& g) {
  int32_t g19[10], f2[10];
#pragma unroll
  for (int i = 0; i < 10; i++) {
    g19[i] = 19 * g.v[i];
    f2[i] = 2 * f.v[i];
  }
  int64_t h[10];
#pragma unroll
  for (int k = 0; k < 10; k++) h[k] = 0;
#pragma unroll
  for (int i = 0; i < 10; i++)
#pragma unroll
    for (int j = 0; j < 10; j++) {
      int32_t a = ((i & 1) && (j & 1)) ? f2[i] : f.v[i];   // 2^ceil(25.5i) 2^ceil(25.5j) = 2 * 2^ceil(25.5(i+j)) when both odd
      int32_t b = (i + j >= 10) ? g19[j] : g.v[j];          // 2^255 = 19
      h[(i + j) % 10] += (int64_t)a * b;
    }
  return fe10_carry(h);
}
SP_HD Fe10 fe10_sqr(const Fe10& f) {
  int32_t f19[10], f2[10];
#pragma unroll
  for (int i = 0; i < 10; i++) {
    f19[i] = 19 * f.v[i];
    f2[i] = 2 * f.v[i];
  }
  int64_t h[10];
#pragma unroll
  for (int k = 0; k < 10; k++) h[k] = 0;
#pragma unroll
  for (int i = 0; i < 10; i++) {
    {  // diagonal term f_i^2 (x2 when i is odd)
      int32_t a = (i & 1) ? f2[i] : f.v[i];
      int32_t b = (2 * i >= 10) ? f19[i] : f.v[i];
      h[(2 * i) % 10] += (int64_t)a * b;
    }
#pragma unroll
    for (int j = i + 1; j < 10; j++) {  // cross terms counted twice (x4 when both odd): 2*f_i fits, the extra 2 goes on the wide side
      int32_t a = f2[i];
      int32_t b = (i + j >= 10) ? f19[j] : f.v[j];
      int64_t prod = (int64_t)a * b;
      h[(i + j) % 10] += ((i & 1) && (j & 1)) ? 2 * prod : prod;
    }
  }
  return fe10_carry(h);
}
SP_HD Fe10 fe10_pow2k(Fe10 a, int k) {
  for (int i = 0; i < k; i++) a = fe10_sqr(a);
  return a;
}
// z^(2^250-1) and z^11
SP_HD void fe10_pow_ladder(const Fe10& z, Fe10* z2_250_0, Fe10* z11) {
  Fe10 z2 = fe10_sqr(z);
  Fe10 z9 = fe10_mul(fe10_pow2k(z2, 2), z);
  *z11 = fe10_mul(z9, z2);
  Fe10 z2_5_0 = fe10_mul(fe10_sqr(*z11), z9);
  Fe10 z2_10_0 = fe10_mul(fe10_pow2k(z2_5_0, 5), z2_5_0);
  Fe10 z2_20_0 = fe10_mul(fe10_pow2k(z2_10_0, 10), z2_10_0);
  Fe10 z2_40_0 = fe10_mul(fe10_pow2k(z2_20_0, 20), z2_20_0);
  Fe10 z2_50_0 = fe10_mul(fe10_pow2k(z2_40_0, 10), z2_10_0);
  Fe10 z2_100_0 = fe10_mul(fe10_pow2k(z2_50_0, 50), z2_50_0);
  Fe10 z2_200_0 = fe10_mul(fe10_pow2k(z2_100_0, 100), z2_100_0);
  *z2_250_0 = fe10_mul(fe10_pow2k(z2_200_0, 50), z2_50_0);
}
SP_HD Fe10 fe10_pow_p58(const Fe10& x) {  // x^((p-5)/8)
  Fe10 t, z11;
  fe10_pow_ladder(x, &t, &z11);
  return fe10_mul(fe10_pow2k(t, 2), x);
}
SP_HD Fp fp_pow_p58_serial(const Fp& z) {
  Fe10 x = fe10_load(z);
  fe10_pin(x);
  return fe10_to_fp(fe10_pow_p58(x));
}
SP_HD Fp fp_invert_serial(const Fp& z) {  // z^(p-2)
  Fe10 x = fe10_load(z);
  fe10_pin(x);
  Fe10 t, z11;
  fe10_pow_ladder(x, &t, &z11);
  return fe10_to_fp(fe10_mul(fe10_pow2k(t, 5), z11));
}

}  // namespace sp
